#!/bin/bash
# a slice of the kernel trace from the middle of the THIRD whole epoch (4 000 dispatches): what the streams do at block boundaries
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_epoch_slice
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o ep -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-lightgcn --no-gru --no-cpu-baseline --no-eval --large-batches= > $OUT.json 2> $OUT.err
cd $R && python3 - <<'PY'
import csv
src = "gpurun_out/prof_epoch_slice/ep_kernel_trace.csv"
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(rows) if "bpr_fused_step_kernel" in r["Kernel_Name"]]
mid = steps[int(len(steps) * 0.85)]
keep = rows[mid - 2000: mid + 2000]
with open("gpurun_out/epoch_slice.csv", "w") as f:
    f.write("start_ns,end_ns,queue,stream,kernel\n")
    for r in keep:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        f.write(f"{r['Start_Timestamp']},{r['End_Timestamp']},{r['Queue_Id']},{r['Stream_Id']},{n}\n")
print("kept", len(keep), "of", len(rows))
PY
rm -rf $OUT
