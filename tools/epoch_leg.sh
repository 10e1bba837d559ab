# usage: bash tools/epoch_leg.sh -- bench.py's whole-epoch leg by block length of the blocked Adam
for k in 8 16; do
  SKR_ADAM_BLOCK=$k python bench.py --no-cpu-baseline --no-eval 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['full_epoch']
print('k=$k value=%.0f | epoch: %.0f int/s %.3f s (first epoch %.0f int/s %.3f s) steps=%d'%(d['value'], e['interactions_per_sec'], e['seconds'], e['first_epoch_interactions_per_sec'], e['first_epoch_seconds'], e['steps']))"
done
