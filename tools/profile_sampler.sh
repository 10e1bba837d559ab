R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sampler -o s -- python3 $R/tools/microbench.py > /dev/null 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_sampler/s_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("mt_generate","exact_assign","mt_commit","sample_fast","max_row_len","spmm")):
        print(n.split("(")[1][:40] if n.startswith("(") else n[:50], r["Calls"], "avg_ms=%.3f"%(float(r["AverageNs"])/1e6), "max_ms=%.3f"%(float(r["MaxNs"])/1e6))
PY
