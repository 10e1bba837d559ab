#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_edges.py tests/test_abi.py -x -q -m gpu > gpurun_out/r3_f16_tests.log 2>&1; echo "tests exit $?"; tail -15 gpurun_out/r3_f16_tests.log
for m in fp32 bf16x3 f16x2; do SKR_FUSED_MODE=$m timeout -k 10 300 python3 tools/fused_accuracy.py 2>&1 | tail -5; done | tee gpurun_out/r3_f16_accuracy.txt
