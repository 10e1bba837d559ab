"""Condense the rocprofv3 output of tools/profile_epoch.sh (gpurun_out/prof_epoch_<tag>/) into profiles/<tag>_epoch_kernel_stats.csv
and the `...@epoch3` entries of profiles/<tag>_pmc_summary.json: per kernel, the launches of the THIRD whole epoch -- the state
bench.py's `roofline` / `roofline_step` are measured in -- are the last third of the whole-epoch launches."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
COMMAND = "python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-lightgcn --no-gru --no-cpu-baseline --no-eval --large-batches="
src = f"gpurun_out/prof_epoch_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


rows = list(csv.DictReader(open(f"{src}/stats_kernel_stats.csv")))
with open(f"profiles/{tag}_epoch_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- {COMMAND} (MI355X): pre-steps, 5 x 20 timed steps, three whole epochs, 40 blocks alone\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows[:24]:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")

# per-dispatch durations of the third epoch from the kernel trace (dispatch order == time order per kernel)
trace = collections.defaultdict(list)
for r in csv.DictReader(open(f"{src}/stats_kernel_trace.csv")):
    k = short(r["Kernel_Name"])
    if k.startswith(("adam_cold_rows_kernel", "bpr_fused_step_kernel", "bpr_fused_end_kernel", "bpr_fused_pre_kernel")):
        trace[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
pmc = {}
for name in ("fetch", "write"):
    agg = collections.defaultdict(list)
    p = f"{src}/{name}_counter_collection.csv"
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            agg[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    pmc[name] = agg

path = f"profiles/{tag}_pmc_summary.json"
summary = json.load(open(path)) if os.path.exists(path) else {"tag": tag, "command": None, "kernels": {}}
summary["epoch_command"] = COMMAND
for k, lst in trace.items():
    lst.sort()
    # whole-epoch launches: 3 epochs of the same length dominate the list; the third epoch = the launches before the final
    # `alone` slice (40 blocks).  Cold pass: one per block; step: 32 per block.
    per_block = 32 if k.startswith("bpr_fused_step") else 1
    n_alone = 40 * per_block
    body = lst[:-n_alone] if len(lst) > 3 * n_alone else lst
    third = body[-(len(body) // 3 + 1) // 2 * 1:] if False else body[-(len(body) // 3):]
    third = third[len(third) // 6:]                  # skip the epoch's first sixth (blocks 257.. in bench.py's own events)
    ent = {"launches": len(third), "avg_ns": sum(d for _, d in third) / max(len(third), 1), "state": "third whole epoch of the run"}
    for name, key in (("fetch", "FETCH_SIZE_KB"), ("write", "WRITE_SIZE_KB")):
        vals = sorted(pmc[name].get(k, []))
        if vals:
            vb = vals[:-n_alone] if len(vals) > 3 * n_alone else vals
            v3 = vb[-(len(vb) // 3):]
            v3 = v3[len(v3) // 6:]
            ent[key] = sum(v for _, v in v3) / len(v3)
    if "FETCH_SIZE_KB" in ent and "WRITE_SIZE_KB" in ent:
        # FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies a 128-byte request at 64 B): calibrated on the cold
        # pass against a known byte count in round 1 (x 1.92, profiles/r01_pmc_calibration_cold.json); the step / end / pre launches
        # read in the same shape (one dword per lane, whole 256-byte rows), so the same factor applies to them
        mult = 2.0
        ent["hbm_read_bytes"] = ent["FETCH_SIZE_KB"] * 1024 * mult
        ent["hbm_write_bytes"] = ent["WRITE_SIZE_KB"] * 1024
        ent["hbm_bytes_per_launch"] = ent["hbm_read_bytes"] + ent["hbm_write_bytes"]
        ent["note"] = f"FETCH_SIZE x {mult:g}"
    summary["kernels"][k + "@epoch3"] = ent
json.dump(summary, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in summary["kernels"].items() if k.endswith("@epoch3")}, indent=1))
