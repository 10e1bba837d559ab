for tr in 20 26 42 58 90 150; do
  SKR_FUSED_TRIGGER=$tr python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users 262144 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trigger=$tr', 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'])"
done
for k in 20 50 100; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users 262144 --top-k $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('top_k=$k', 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'])"
done
