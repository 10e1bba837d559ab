#!/bin/bash
# workgroups per CU of the persistent cold pass (SKR_COLD_BPC) x issue priority of the step kernel (SKR_FUSED_DBG=16 switches it OFF since round 3), same box:
# $@ = "BPC:DBG" pairs
for cfg in ${@:-5:0 4:16 5:0 6:0}; do
  v=${cfg%%:*}; g=${cfg##*:}
  SKR_COLD_BPC=$v SKR_FUSED_DBG=$g timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-lightgcn --no-gru --no-eval --large-batches "" 2> gpurun_out/cold_bpc.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bpc $v prio $g value', round(d['value']/1e6, 2), 'epochs', [round(x, 4) for x in d.get('full_epoch', {}).get('epochs_seconds', [])], 'step_us', round(d['roofline_step']['avg_launch_us'], 2), 'cold_ms', round(d['roofline']['avg_launch_ms'], 4))" || exit 1
done
