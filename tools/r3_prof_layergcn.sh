#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_layer -o lay -- python3 $R/tools/layergcn_fullsize.py 10 > $R/gpurun_out/r3_layergcn.txt 2>&1; echo "exit $?"
cd $R
tail -1 gpurun_out/r3_layergcn.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_layer/lay_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"])
PY
rm -f gpurun_out/prof_layer/lay_kernel_trace.csv
