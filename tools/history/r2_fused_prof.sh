#!/bin/bash
# per-kernel time of the N = 1 BPRMF loop (200 timed steps), fused one-launch step; $1 = tag, env passes through
R=$GRAFT_REPO_ROOT
TAG=${1:-fused}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o fused -- python3 $R/bench.py --gpus 1 --steps 200 --warmup 5 --no-lightgcn --no-gru --no-epoch --no-cpu-baseline --no-eval > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
cd $R
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
python3 -c "
import json
d=json.loads(open('gpurun_out/prof_$TAG.json').read().strip().splitlines()[-1]); print('$TAG value %.3g' % d['value'], d['ms_per_step'])"
