for k in 50 100; do for tr in 0 128 160 192 224; do
  SKR_FUSED_TRIGGER=$tr python bench.py --steps 5 --warmup 1 --no-cpu-baseline --eval-users 262144 --top-k $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('top_k=$k trigger=$tr', 'TF=%.1f'%d['roofline_eval']['achieved'], 'ms=%.2f'%d['roofline_eval']['avg_launch_ms'])"
done; done
