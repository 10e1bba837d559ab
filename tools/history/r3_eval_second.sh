#!/bin/bash
mkdir -p gpurun_out
bash tools/r3_eval_pmc.sh bf16x3g:0 bf16x3g:7 bf16x3g:1 bf16x3s:7 2>&1 | tee gpurun_out/r3_eval_pmc.txt
cd $GRAFT_REPO_ROOT
for t in 26 42 60 90; do echo "trigger $t"; SKR_FUSED_TRIGGER=$t bash tools/r3_eval.sh bf16x3g:0; done 2>&1 | tee gpurun_out/r3_eval_trig.txt
