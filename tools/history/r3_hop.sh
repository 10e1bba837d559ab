#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_config0.py tests/test_gpu_dist.py -q -x -k "spmm or lightgcn or layergcn or graph or propagation or sharded" > gpurun_out/r3_hop_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/r3_hop_tests.log
for v in 1 0; do
SKR_FIRST_HOP_SCATTER=$v timeout -k 10 300 python3 bench.py --workload lightgcn --steps 10 --warmup 2 --no-cpu-baseline --large-batches "" > gpurun_out/r3_hop$v.json 2> gpurun_out/r3_hop$v.err; echo "lightgcn scatter=$v exit $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_hop$v.json').read().strip().splitlines()[-1])
print('scatter=$v lightgcn ms/step', d['ms_per_step'])"
SKR_FIRST_HOP_SCATTER=$v timeout -k 10 300 python3 tools/layergcn_fullsize.py 10 2>&1 | tail -1
done
