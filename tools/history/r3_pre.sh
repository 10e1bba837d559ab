#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_config0.py -q -x -k "fused or bprmf or blocked" > gpurun_out/r3_pre_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r3_pre_tests.log
for pre in 1 0; do
SKR_FUSED_PRE=$pre timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-lightgcn --no-gru --no-cpu-baseline --no-eval --large-batches "" > gpurun_out/r3_pre$pre.json 2> gpurun_out/r3_pre$pre.err; echo "bench pre=$pre exit $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_pre$pre.json').read().strip().splitlines()[-1])
print('pre=$pre value', d['value'], d['repeats']['seconds'], 'epochs', d['full_epoch']['epochs_seconds'])
r=d['roofline_step']; print(' step us', r['avg_launch_us'], 'end', r['end_launch_us_per_block'], 'alone', r['alone']['avg_launch_us'], 'cold ms', d['roofline']['avg_launch_ms'])"
done
