# usage: bash tools/adam_block_sweep.sh -- blocked dense Adam: overlap of the cold pass on/off x its blocks per CU
for ov in 0 1; do for bpc in 1 2 4; do
  SKR_ADAM_OVERLAP=$ov SKR_COLD_BPC=$bpc python bench.py --no-cpu-baseline --no-eval --no-epoch 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench overlap=$ov bpc=$bpc value=%.0f ms/step=%.4f cold ms=%.3f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
for ov in 0 1; do for bpc in 2 4; do
  SKR_ADAM_OVERLAP=$ov SKR_COLD_BPC=$bpc python tools/e2e_scale.py --users 1000000 --items 100000 --interactions 50000000 --epochs 1 2>&1 | grep "\[e2e\] epoch" | cut -c1-70 | sed "s/^/e2e overlap=$ov bpc=$bpc /"
done; done
