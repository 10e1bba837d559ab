#!/bin/bash
# length of the Adam block (SKR_ADAM_BLOCK) with six cold-pass workgroups per CU for whole blocks, same box; $@ = lengths
for k in ${@:-32 48 64 32}; do
  SKR_ADAM_BLOCK=$k timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-lightgcn --no-gru --no-eval --large-batches "" 2> gpurun_out/block_len.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('block $k value', round(d['value']/1e6, 2), 'epochs', [round(x, 4) for x in d.get('full_epoch', {}).get('epochs_seconds', [])], 'step_us', round(d['roofline_step']['avg_launch_us'], 2), 'end_us', round(d['roofline_step']['end_launch_us_per_block'], 1), 'cold_ms', round(d['roofline']['avg_launch_ms'], 4))" || exit 1
done
