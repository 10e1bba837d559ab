"""Condense the rocprofv3 CSVs that tools/profile_bench.sh / profile_eval.sh left under
gpurun_out/prof_<tag>/ into the small, tracked files under profiles/."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
COMMAND = "python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-epoch"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


# 1. --kernel-trace --stats summary (top kernels)
rows = list(csv.DictReader(open(f"{src}/stats_kernel_stats.csv")))
with open(f"profiles/{tag}_bench_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- {COMMAND} (MI355X)\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows[:40]:
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
stats = {short(r["Name"]): r for r in rows}

# 2. PMC passes: FETCH_SIZE / WRITE_SIZE per launch (KB), gfx950 correction: FETCH_SIZE x2 for wide
#    coalesced streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-B stores.
pmc = {}
for name in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{src}/{name}_counter_collection.csv")):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    pmc[name] = agg
summary = {"tag": tag, "command": COMMAND, "kernels": {}}
# L2 hit rate and vector-cache -> L2 read requests of the propagation kernels (own --pmc passes)
extra = {}
for fname, counters in (("l2", ("TCC_HIT_sum", "TCC_MISS_sum")), ("tcp", ("TCP_TCC_READ_REQ_sum",))):
    pth = f"{src}/{fname}_counter_collection.csv"
    if not os.path.exists(pth):
        continue
    for r in csv.DictReader(open(pth)):
        extra.setdefault(short(r["Kernel_Name"]), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("adam_cold_rows_kernel<4>", "adam_cold_kernel<2>", "adam_hot_kernel", "bpr_fused_step_kernel", "bpr_fused_end_kernel", "bpr_fused_pre_kernel", "bpr_step_kernel<1>", "clear_marked_rows_kernel",
          "fused_plan_kernel<0>", "fused_plan_kernel<1>", "fused_plan_kernel<2>", "fused_plan_kernel<3>", "adam_kernel<true, 4, true>", "fused_topk_kernel_v5<true>", "fused_topk_kernel_v4<true>", "fused_topk_kernel_v3<true>", "split_items_kernel",
          "bpr_step_kernel", "exact_assign_kernel<false>", "mt_generate_kernel", "slab_detect_kernel", "slab_resolve_kernel", "slab_scatter_kernel",
          "shuffle_gather_kernel<false>", "adam_kernel<false, 4, true>", "adam_kernel<true, 4, true>", "gru_fwd_kernel<4>", "session_logits_kernel",
          "spmm_rows_kernel<false>", "spmm_rows_kernel<true>", "spmm_tasks_kernel<false>", "spmm_tasks_kernel<true>", "spmm_reduce_kernel"):
    if k not in stats:
        continue
    fe, wr = pmc["fetch"].get(k, []), pmc["write"].get(k, [])
    ent = {"calls": int(stats[k]["Calls"]), "avg_ns": float(stats[k]["AverageNs"])}
    if fe and wr:
        # for multi-size kernels (the eval kernel has a short warm-up launch) take the largest launch
        fkb, wkb = (max(fe), max(wr)) if "fused" in k else (sum(fe) / len(fe), sum(wr) / len(wr))
        ent.update(FETCH_SIZE_KB=fkb, WRITE_SIZE_KB=wkb, hbm_read_bytes=fkb * 1024 * 2, hbm_write_bytes=wkb * 1024,
                   hbm_bytes_per_launch=fkb * 1024 * 2 + wkb * 1024,
                   note="FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read stream)")
    for cname, vals in extra.get(k, {}).items():
        ent[cname + "_per_launch"] = sum(vals) / len(vals)
    if "TCC_HIT_sum_per_launch" in ent:
        h, m_ = ent["TCC_HIT_sum_per_launch"], ent["TCC_MISS_sum_per_launch"]
        ent["l2_hit_rate"] = h / max(h + m_, 1.0)
    summary["kernels"][k] = ent

# 3. eval kernel PMC (MFMA utilisation, clock): the largest launch of each fused kernel (v7 = f16x2, the default; v6 = bf16x3;
#    v3 = FP32 MFMA, run by bench.py for comparison)
p = f"{src}/evalpmc_counter_collection.csv"
if os.path.exists(p):
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        if "fused_topk" in r["Kernel_Name"]:
            d = agg[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            d["grid_threads"] = int(r["Grid_Size"])
            kn = r["Kernel_Name"]
            d["kernel"] = ("v7_f16x2" if "topk_kernel_v7" in kn else "v6_bf16x3" if "topk_kernel_v6" in kn else "v5_bf16x3" if "topk_kernel_v5" in kn else
                           "v4_bf16x3" if "topk_kernel_v4" in kn else "v3_fp32")
    for tagk in ("v7_f16x2", "v6_bf16x3", "v5_bf16x3", "v4_bf16x3", "v3_fp32"):
        cand = [d for d in agg.values() if d["kernel"] == tagk]
        if not cand:
            continue
        big = max(cand, key=lambda d: (d["grid_threads"], d["dur_us"]))
        if big["dur_us"] < 100.0:     # the guard's fall-back launch over an empty list: not a measurement
            continue
        clk = big["GRBM_GUI_ACTIVE"] / 8 / (big["dur_us"] * 1e-6)
        simd_cycles = big["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        big["effective_clock_GHz"] = clk / 1e9
        big["mfma_busy_fraction_at_effective_clock"] = simd_cycles / (big["GRBM_GUI_ACTIVE"] / 8)
        summary[f"eval_pmc_largest_launch_{tagk}"] = big
# entries of the whole-epoch state (tools/profile_epoch.sh -> tools/summarize_epoch_profile.py) live in the same file: keep them
_old = f"profiles/{tag}_pmc_summary.json"
if os.path.exists(_old):
    _prev = json.load(open(_old))
    summary["kernels"].update({k: v for k, v in _prev.get("kernels", {}).items() if "@" in k})
    if "epoch_command" in _prev:
        summary["epoch_command"] = _prev["epoch_command"]
json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
