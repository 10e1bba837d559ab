#!/bin/bash
# fused evaluator on 262 144 users, same box, same call: $@ = "MODE:ABLATE[:TOPK]" triples
for cfg in "$@"; do
  IFS=: read mode abl k <<< "$cfg"; k=${k:-10}
  SKR_FUSED_MODE=$mode SKR_FUSED_ABLATE=$abl timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --no-lightgcn --no-gru --no-epoch --large-batches "" --top-k $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode ablate=$abl top_k=$k', 'TF=%.1f'%d['roofline_eval']['achieved'], 'frac=%.3f'%d['roofline_eval']['frac'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'users/s=%.0f'%d['eval']['users_per_sec'], 'HR=%.5f'%d['eval']['HR@$k'])" || exit 1
done
