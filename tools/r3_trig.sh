#!/bin/bash
# compaction trigger sweep of the default evaluator kernel, same box: $1 = top_k, rest = triggers
k=$1; shift
for t in "$@"; do echo -n "top_k $k trigger $t: "; SKR_FUSED_TRIGGER=$t bash tools/r3_eval.sh f16x2:0:$k; done
