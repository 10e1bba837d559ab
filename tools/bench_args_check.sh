# usage: bash tools/bench_args_check.sh -- bench.py must survive any K / W the driver may choose (small workload, quick)
S="--users 60000 --items 5000 --interactions 1500000 --eval-users 4096 --no-cpu-baseline"
for kw in "5 1" "50 0" "24 24" "1 0" "97 3"; do set -- $kw
  python bench.py $S --steps $1 --warmup $2 2>gpurun_out/args_check.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=$1 W=$2 ok value=%.0f steps=%d warmup=%d epoch=%s'%(d['value'], d['steps'], d['warmup'], 'full_epoch' in d))" || { echo "K=$1 W=$2 FAILED"; tail -5 gpurun_out/args_check.err; }
done
bash tools/bench_ranks.sh 2 --steps 10 --warmup 2 --users 60000 --items 5000 --interactions 1500000 --eval-users 4096 | cut -c1-120
bash tools/bench_ranks.sh 3 --steps 7 --warmup 0 --users 60000 --items 5000 --interactions 1500000 --eval-users 4096 | cut -c1-120
