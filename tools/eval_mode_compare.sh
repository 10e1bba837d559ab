# usage: bash tools/eval_mode_compare.sh -- fused evaluation, fp32 MFMA vs the bf16x3 split, at several top_k
for k in 10 50 100; do for m in fp32 bf16x3; do
  SKR_FUSED_MODE=$m python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users 262144 --top-k $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['eval']; print('$m top_k=$k', 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'], 'users/s=%.0f'%e['users_per_sec'], {k:v for k,v in e.items() if '@' in k})"
done; done
