#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "SKR_SPMM_HOT=0" "SKR_SPMM_HOT=1" "SKR_SPMM_HOT_DENSITY=2" "SKR_SPMM_HOT_DENSITY=3" "SKR_SPMM_HOT=0" "SKR_SPMM_HOT=1"; do
env $v timeout -k 10 200 python3 tools/hot_rows_lab.py 2>/dev/null | cut -c1-120
done
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -q -k "spmm" 2>&1 | tail -2
