#!/bin/bash
# whole bench line with the compute stream = the null stream (0) or a stream of its own (1), same box
for p in ${@:-0 1}; do
  SKR_COMPUTE_STREAM=$p timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r3_priofull_$p.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
lg = d.get('lightgcn', {}); g = d.get('gru4rec', {})
print('stream $p value', round(d['value']/1e6, 2), 'epochs', [round(x, 4) for x in d.get('full_epoch', {}).get('epochs_seconds', [])], 'eval', round(d['eval']['users_per_sec']/1e6, 2), 'lightgcn', round(lg.get('ms_per_step', 0), 3), {k: round(v['ms_per_step'], 2) for k, v in lg.get('large_batch', {}).items()}, 'layer', round(lg.get('roofline', {}).get('avg_launch_ms', 0), 3), 'gru', round(g.get('ms_per_step', 0), 4), 'large', {k: round(v['value']/1e6, 1) for k, v in d.get('large_batch', {}).items()})" || exit 1
done
