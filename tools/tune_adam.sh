for cfg in 8,1,0 8,2,0 8,4,0 4,2,0 4,4,0 16,1,0 16,2,0 8,1,1 8,2,1 8,4,1 4,4,1 2,4,1 32,1,0; do
  SKR_ADAM_CFG=$cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-eval --no-epoch 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['achieved'],1))"
done
