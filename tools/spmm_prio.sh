#!/bin/bash
# LightGCN leg with the short-row kernel (rows) or the long-row tasks (tasks) at issue priority 3, same box
for v in ${@:-none rows tasks none}; do
  SKR_SPMM_PRIO=$v timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-eval --no-epoch --no-gru --large-batches "" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
lg = d['lightgcn']
print('prio $v lightgcn ms/step', round(lg['ms_per_step'], 3), 'layer ms', round(lg['roofline']['avg_launch_ms'], 4))" || exit 1
done
