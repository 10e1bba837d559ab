R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_sampler -o p -- python3 $R/tools/microbench.py > /dev/null 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(dict)
for r in csv.DictReader(open("$R/gpurun_out/prof_sampler/p_counter_collection.csv")):
    if "exact_assign" in r["Kernel_Name"] or "mt_generate" in r["Kernel_Name"]:
        d=agg[(r["Kernel_Name"][:50], r["Dispatch_Id"])]
        d[r["Counter_Name"]]=float(r["Counter_Value"]); d["dur_ms"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,v in list(agg.items())[:4]: print(k, v)
PY
