#!/bin/bash
# the plan's shape and launch parameters on the LightGCN step (same box); $@ = "LONG_FROM:CBLK" pairs (0 = default); HOTD, SPAN, RWGS, TWGS in the environment:
# rows cut into tasks from SKR_SPMM_LONG_FROM entries, SKR_SPMM_CBLK columns per block, SKR_SPMM_HOT_DENSITY, SKR_SPMM_TASK_SPAN, SKR_SPMM_ROWS_WGS, SKR_SPMM_TASK_WGS
for cfg in ${@:-512:16384 256:16384 384:16384 768:16384 1024:16384 512:32768 256:32768 512:16384}; do
  lf=${cfg%%:*}; cb=${cfg##*:}
  SKR_SPMM_LONG_FROM=$lf SKR_SPMM_CBLK=$cb SKR_SPMM_HOT_DENSITY=${HOTD:-4} SKR_SPMM_TASK_SPAN=${SPAN:-1} env ${RWGS:+SKR_SPMM_ROWS_WGS=$RWGS} ${TWGS:+SKR_SPMM_TASK_WGS=$TWGS} timeout -k 10 300 python3 bench.py --workload lightgcn --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-layergcn --large-batches "" 2> gpurun_out/spmm_params.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('rows_wgs ${RWGS:-default} task_wgs ${TWGS:-default} span ${SPAN:-1} hotd ${HOTD:-4} long_from $lf cblk $cb ms/step', round(d['ms_per_step'], 3), 'layer', round(r['avg_launch_ms'], 3), 'user', round(r['user_side_ms'], 3), 'item', round(r['item_side_ms'], 3), 'tasks', r['plan']['item_side']['tasks'], 'long', r['plan']['item_side']['long_rows'], 'hot', r['plan']['item_side']['hot_rows'])" || exit 1
done
