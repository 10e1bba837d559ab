#!/bin/bash
# cold-pass workgroups per CU: short blocks (SKR_COLD_BPC) x whole 32-step blocks (SKR_COLD_BPC_FULL), same box; $@ = "BPC:FULL" pairs
for cfg in ${@:-5:0 5:6 5:0 5:6 6:0}; do
  v=${cfg%%:*}; g=${cfg##*:}
  SKR_COLD_BPC=$v SKR_COLD_BPC_FULL=$g timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-lightgcn --no-gru --no-eval --large-batches "" 2> gpurun_out/bpc_full.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bpc $v full $g value', round(d['value']/1e6, 2), [round(20*1024/x/1e6, 1) for x in d['repeats']['seconds']], 'epochs', [round(x, 4) for x in d.get('full_epoch', {}).get('epochs_seconds', [])], 'step_us', round(d['roofline_step']['avg_launch_us'], 2), 'cold_ms', round(d['roofline']['avg_launch_ms'], 4))" || exit 1
done
