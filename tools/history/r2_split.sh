#!/bin/bash
set -e
for sp in 1 0; do
  SKR_FUSED_SPLIT_END=$sp timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_models.py tests/test_gpu_config0.py -x -q -k "fused or bprmf" > gpurun_out/r2_split_tests_$sp.log 2>&1 || { tail -30 gpurun_out/r2_split_tests_$sp.log; exit 1; }
  tail -1 gpurun_out/r2_split_tests_$sp.log
done
bash tools/r2_sweep.sh X=0 SKR_FUSED_SPLIT_END=1 X=1 SKR_FUSED_SPLIT_END=1
