#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_config0.py tests/test_gpu_dist.py -q -x -k "spmm or lightgcn or layergcn or graph or propagation or sharded" > gpurun_out/r3_hot_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r3_hot_tests.log
for hot in 1 0; do
SKR_SPMM_HOT=$hot timeout -k 10 300 python3 bench.py --workload lightgcn --steps 10 --warmup 2 --no-cpu-baseline --large-batches "" > gpurun_out/r3_hot$hot.json 2> gpurun_out/r3_hot$hot.err; echo "lightgcn hot=$hot exit $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_hot$hot.json').read().strip().splitlines()[-1])
print('hot=$hot lightgcn ms/step', d['ms_per_step'], 'layer', d['roofline']['avg_launch_ms'], d['roofline']['user_side_ms'], d['roofline']['item_side_ms'], d['roofline']['plan']['item_side'])"
done
