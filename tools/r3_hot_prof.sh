#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hot -o h -- python3 $R/bench.py --workload lightgcn --steps 10 --warmup 2 --no-cpu-baseline --large-batches "" > $R/gpurun_out/r3_hotp.json 2> $R/gpurun_out/r3_hotp.err; echo "exit $?"
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_hot/h_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"])
PY
rm -f gpurun_out/prof_hot/h_kernel_trace.csv
