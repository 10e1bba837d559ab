#!/bin/bash
# compaction trigger sweep of the default evaluator kernel, same box
for t in ${@:-60 42 90 60}; do echo -n "trigger $t: "; SKR_FUSED_TRIGGER=$t bash tools/r3_eval.sh f16x2:0; done
