#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): the planning stream and the optimiser's side stream of
# the BPRMF epoch were seen on ONE queue (rocprofv3 Queue_Id), i.e. the next block's plan kernels in front of this block's
# cold pass.  Same box: the epoch leg with 4 (default) / 8 / 16 hardware queues.
for q in ${@:-default 8 16 default 8}; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-lightgcn --no-gru --no-eval --large-batches "" 2> gpurun_out/hw_queues.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hw queues $q value', round(d['value']/1e6, 2), 'epochs', [round(x, 4) for x in d.get('full_epoch', {}).get('epochs_seconds', [])], 'step_us', round(d['roofline_step']['avg_launch_us'], 2), 'end_us', round(d['roofline_step']['end_launch_us_per_block'], 1), 'cold_ms', round(d['roofline']['avg_launch_ms'], 4))" || exit 1
done
