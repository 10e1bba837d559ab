#!/bin/bash
# fused evaluator on 262 144 users: $@ = "MODE:ABLATE" pairs
for cfg in "$@"; do
  mode=${cfg%%:*}; abl=${cfg##*:}
  SKR_FUSED_MODE=$mode SKR_FUSED_ABLATE=$abl timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-lightgcn --no-gru --no-epoch 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode ablate=$abl', 'TF=%.1f'%d['roofline_eval']['achieved'], 'frac=%.3f'%d['roofline_eval']['frac'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'users/s=%.0f'%d['eval']['users_per_sec'], 'HR=%.5f'%d['eval']['HR@10'])" || exit 1
done
