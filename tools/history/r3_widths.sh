#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_config0.py tests/test_gpu_eval.py tests/test_gpu_train.py tests/test_gpu_models.py tests/test_gpu_edges.py -q > gpurun_out/r3_widths.log 2>&1; echo "tests exit $?"; tail -12 gpurun_out/r3_widths.log
