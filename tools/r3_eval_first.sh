#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_edges.py -x -q -m gpu -k "fused or eval" > gpurun_out/r3_eval_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r3_eval_tests.log
bash tools/r3_eval.sh "$@" 2>&1 | tee gpurun_out/r3_eval_ab.txt
