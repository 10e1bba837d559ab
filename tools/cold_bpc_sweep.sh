# usage: bash tools/cold_bpc_sweep.sh -- training leg by blocks per CU of the cold pass (its share of the chip) and block length k
for k in 24; do for bpc in 2 3 4 8; do
  SKR_ADAM_BLOCK=$k SKR_COLD_BPC=$bpc python bench.py --no-cpu-baseline --no-eval --no-epoch --steps 960 --warmup 48 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('k=$k bpc=$bpc value=%.0f ms/step=%.4f cold ms=%.3f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
