# usage: bash tools/adam_k_sweep.sh -- bench.py training leg by block length k of the blocked Adam, fresh and aged optimiser
for k in 16 24 32; do for t in 0 20000; do
  SKR_ADAM_BLOCK=$k python bench.py --no-cpu-baseline --no-eval --no-epoch --steps 480 --warmup 48 --start-step $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('k=$k start=$t value=%.0f ms/step=%.4f cold ms=%.3f'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
