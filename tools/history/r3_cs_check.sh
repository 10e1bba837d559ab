#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_config0.py tests/test_gpu_gru.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/r3_cs_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r3_cs_tests.log
bash tools/r3_compute_stream.sh 1 0
