#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gru -o gru -- python3 $R/bench.py --gpus 1 --workload gru4rec --steps 100 --warmup 5 --no-cpu-baseline --no-eval --no-epoch > $R/gpurun_out/prof_gru.json 2> $R/gpurun_out/prof_gru.err
cd $R
python3 - <<'PY'
import csv, json
rows=list(csv.DictReader(open("gpurun_out/prof_gru/gru_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
d=json.loads(open("gpurun_out/prof_gru.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
rm -f gpurun_out/prof_gru/gru_kernel_trace.csv
