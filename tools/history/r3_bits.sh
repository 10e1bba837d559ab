#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_config0.py -q -x -k "spmm or lightgcn or layergcn or graph or propagation" > gpurun_out/r3_bits_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r3_bits_tests.log
for v in 1 0 1; do
SKR_SPMM_MASK_BITS=$v timeout -k 10 300 python3 bench.py --workload lightgcn --steps 10 --warmup 2 --no-cpu-baseline --large-batches "" > gpurun_out/r3_bits$v.json 2> gpurun_out/r3_bits$v.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_bits$v.json').read().strip().splitlines()[-1])
print('mask bits=$v lightgcn ms/step', d['ms_per_step'])"
done
