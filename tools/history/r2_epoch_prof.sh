#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_epoch -o ep -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-lightgcn --no-gru --no-cpu-baseline --no-eval > $R/gpurun_out/prof_epoch.json 2> $R/gpurun_out/prof_epoch.err
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_epoch/ep_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
rm -f gpurun_out/prof_epoch/ep_kernel_trace.csv
