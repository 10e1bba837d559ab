#!/bin/bash
# PMC pass on the fused evaluation kernel (own run, --kernel-trace only).
TAG=${1:-r01}; EU=${2:-65536}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d $R/gpurun_out/prof_$TAG -o evalpmc -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-epoch --no-lightgcn --no-gru --large-batches= --eval-users $EU > $R/gpurun_out/prof_$TAG.evalpmc.log 2>&1
echo "exit $?"
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$TAG/evalpmc_counter_collection.csv")))
agg=collections.defaultdict(dict)
for r in rows:
    if "fused_topk" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
        agg[r["Dispatch_Id"]]["dur_us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        agg[r["Dispatch_Id"]]["grid"]=r["Grid_Size"]
for k,v in agg.items(): print(k, v)
PY
