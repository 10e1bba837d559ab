#!/bin/bash
for k in 10 50 100; do for mode in bf16x3 bf16x3s; do
  SKR_FUSED_MODE=$mode timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 1 --top-k $k --no-cpu-baseline --no-lightgcn --no-gru --no-epoch 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$k $mode', 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'users/s=%.0f'%d['eval']['users_per_sec'])" || exit 1
done; done
