#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py's training leg only (no evaluation, no CPU baseline); extra bench args after the tag
TAG=${1:-r01_train}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o stats -- python3 $R/bench.py --steps 208 --warmup 16 --no-cpu-baseline --no-eval --no-epoch "$@" > $R/gpurun_out/prof_$TAG.stats.log 2>&1
echo "profile exit $?"
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_$TAG/**/stats_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-60s calls=%6s avg_us=%9.2f total_ms=%9.2f pct=%s"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
