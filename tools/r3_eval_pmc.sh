#!/bin/bash
# PMC passes on the fused evaluation kernel, one per "MODE:ABLATE" argument (own runs, --kernel-trace only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  mode=${cfg%%:*}; abl=${cfg##*:}
  export SKR_FUSED_MODE=$mode SKR_FUSED_ABLATE=$abl
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d $R/gpurun_out/prof_evalpmc_${mode}_$abl -o evalpmc -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --no-epoch --no-lightgcn --no-gru --large-batches= --eval-users 262144 > $R/gpurun_out/prof_evalpmc_${mode}_$abl.log 2>&1 || { echo "rocprof failed for $cfg"; exit 1; }
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/prof_evalpmc_${mode}_$abl/evalpmc_counter_collection.csv")))
agg=collections.defaultdict(dict)
for r in rows:
    if ("fused_topk_kernel_v7" if "$mode" == "f16x2" else "fused_topk_kernel_v3" if "$mode" == "fp32" else "fused_topk_kernel_v") in r["Kernel_Name"] and not ("$mode" != "fp32" and "kernel_v3" in r["Kernel_Name"]):
        d=agg[r["Dispatch_Id"]]
        d[r["Counter_Name"]]=float(r["Counter_Value"]); d["dur_us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; d["grid"]=int(r["Grid_Size"])
big=max(agg.values(), key=lambda d:(d["grid"], d["dur_us"]))
clk=big["GRBM_GUI_ACTIVE"]/8/(big["dur_us"]*1e3)
print("$cfg", "dur_ms=%.2f"%(big["dur_us"]/1e3), "clock_GHz=%.3f"%clk, "mfma_busy=%.3f"%(big["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(big["GRBM_GUI_ACTIVE"]/8)) , "valu_insts=%.3g"%big["SQ_INSTS_VALU"], "wave_cycles=%.4g"%big["SQ_WAVE_CYCLES"], "busy_raw=%.4g"%big["SQ_VALU_MFMA_BUSY_CYCLES"])
PY
done
