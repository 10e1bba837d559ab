"""bench.py -- the hot path of BASELINE.json on N MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

HEADLINE (`value`): BASELINE configs[1] -- BPRMF d=64, synthetic MovieLens-shaped 1M users / 100K items / ~50M
interactions, exact-stream negative sampling + fused BPR step + dense Adam (train), fused MFMA top-K (eval).
A "step" is one mini-batch through the whole training path: its share of the epoch's negative sampling (the sampler call
that produces exactly the negatives these K steps consume sits INSIDE the timed region), the device shuffle + batch
assembly (skr_shuffle_gather, one launch), skr_bpr_step, the exchange of the item gradient (N > 1), and the step's dense
Adam update of every parameter of the flat [U|V|b] buffer (the reference's dense-Adam semantics) in its temporally
blocked, bit-identical form: one cold pass per 32 steps over the rows no batch of the block touches + one hot launch per
step (SKR_ADAM_BLOCK=1: one skr_adam_step per step).  Users are sharded u % N; the item table and bias are replicated.

N > 1, `--scaling strong` (default; what BASELINE's north_star scores): the SAME job on more GPUs -- global batch fixed at
--batch (1024), every rank walks the same global batches and keeps its users' triples (skrec.parallel.ShardedBPRMF, the
engine the drop-in API runs under torchrun).  `--scaling weak`: per-rank batch fixed at --batch, global batch N * batch.

SECONDARY LEG in the same JSON line (`lightgcn`): BASELINE configs[2] (N = 1) / configs[3] (N > 1) -- LightGCN, 3 layers,
the reference's full-graph propagation forward AND backward on every mini-batch, through skrec.parallel.ShardedLightGCN.
`--workload lightgcn` makes that leg the headline instead.
Inputs are resident in HBM before the timed regions.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md ("~2.5 PF dense")
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak (spec)
D = 64


def synth_dataset(n_users, n_items, n_inter, seed, dev):
    """MovieLens-shaped implicit feedback, generated on the device: Zipf(0.9) item popularity over a
    random item permutation, log-normal user activity clipped to [20, I/2], items without replacement
    per user, one held-out item per user (leave-one-out => Recall@K == HR@K)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    act = torch.exp(torch.randn(n_users, generator=g, device=dev) * 0.6)
    act = act / act.sum() * (n_inter + n_users) * 1.04     # ~4 % is lost to de-duplication below
    act = act.clamp(20, n_items // 2).round().long()
    pop = 1.0 / torch.arange(1, n_items + 1, device=dev, dtype=torch.float32) ** 0.9
    pop = pop[torch.randperm(n_items, generator=g, device=dev)]
    total = int(act.sum())
    owner = torch.repeat_interleave(torch.arange(n_users, device=dev), act)
    items = torch.empty(total, dtype=torch.long, device=dev)
    chunk = 1 << 24
    for s in range(0, total, chunk):
        items[s:s + chunk] = torch.multinomial(pop, min(chunk, total - s), replacement=True, generator=g)
    key = torch.unique(owner * n_items + items)            # sorted by (user, item), duplicates dropped
    del owner, items
    u = key // n_items
    it = (key % n_items).int()
    counts = torch.bincount(u, minlength=n_users)
    rowptr = torch.zeros(n_users + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    # hold one interaction per user out (position hashed from the user id)
    pick = rowptr[:-1] + (torch.arange(n_users, device=dev) * 2654435761 % counts.clamp(min=1))
    test_item = it[pick].clone()
    keep = torch.ones(len(it), dtype=torch.bool, device=dev)
    keep[pick] = False
    u, it = u[keep].int(), it[keep]
    counts = counts - 1
    rowptr = torch.zeros(n_users + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    return dict(rowptr=rowptr, users=u.contiguous(), items=it.contiguous(), test_item=test_item.contiguous())


def same_on_every_rank(ds, rank, world, dev, dist):
    """rank 0's data set on every rank.  The generator above is seeded, but its device kernels (multinomial's prefix sums,
    unique) do not promise bit-identical results from process to process; under --scaling strong every rank must walk the
    SAME global batches, so the arrays are broadcast once, before anything is timed."""
    if world == 1:
        return ds
    out = {}
    for k in ("rowptr", "users", "items", "test_item"):
        n = torch.tensor([ds[k].numel() if rank == 0 else 0], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=0)
        t = ds[k].contiguous() if rank == 0 else torch.empty(int(n), dtype=ds[k].dtype, device=dev)
        dist.broadcast(t, src=0)
        out[k] = t
    return out


def shard(ds, rank, world, dev):
    """users u % world == rank, re-indexed 0..U_local-1 (their global id is local*world + rank)"""
    if world == 1:
        return ds, torch.arange(len(ds["rowptr"]) - 1, device=dev, dtype=torch.int32)
    n_users = len(ds["rowptr"]) - 1
    mine = torch.arange(rank, n_users, world, device=dev)
    lens = (ds["rowptr"][1:] - ds["rowptr"][:-1])[mine]
    rowptr = torch.zeros(len(mine) + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(lens, 0)
    sel = (ds["users"].long() % world) == rank
    return dict(rowptr=rowptr, users=(ds["users"][sel] // world).int().contiguous(), items=ds["items"][sel].contiguous(),
                test_item=ds["test_item"][mine].contiguous()), mine.int()


def lightgcn_leg(args, world, rank, dev, dist, full, K, W, cpu_baseline):
    """BASELINE configs[2] (N = 1) / configs[3] (N > 1).  A step is the reference's step (LightGCN.py:180-199): full-graph
    3-layer propagation forward AND backward for one global batch, fused BPR on the propagated tables, dense Adam over
    [U_local; I]; users sharded u % N, one all-reduce of the [I, 64] block per layer and direction, started beside the
    user-side product of the same layer (skrec.parallel.ShardedLightGCN).  Returns the leg's dict (every rank)."""
    from skrec import _hip
    from skrec.parallel import DistContext, ShardedLightGCN
    from skrec.utils.py.random import DeviceSampler
    ctx = DistContext(rank, world)
    nU, nI, b = args.users, args.items, args.batch
    strong = args.scaling == "strong"
    gb = b if strong else b * world
    n_inter_total = int(full["rowptr"][-1])
    mine = torch.from_numpy(ctx.owned_users(nU)).to(dev)
    g0 = torch.Generator().manual_seed(2021)
    bound = (6.0 / (nU + D)) ** 0.5
    user0 = ((torch.rand(nU, D, generator=g0) * 2 - 1) * bound)[mine.cpu()]
    item0 = (torch.rand(nI, D, generator=torch.Generator().manual_seed(7)) * 2 - 1) * (6.0 / (nI + D)) ** 0.5
    eng = ShardedLightGCN.from_device_edges(ctx, full["users"], full["items"], nU, nI, user0, item0, 3, 1e-3, 1e-3, b)
    # the epoch slice these steps consume: a user prefix, sampled with the exact stream on every rank
    need = (W + K + 8) * gb
    if world == 1 and args.large_batches:
        need = max(need, 4 * max(int(x) for x in args.large_batches.split(",") if x))
    if world == 1 and not getattr(args, "no_layergcn", False):
        need = max(need, (3 + max(3, min(K, 10))) * 2048)          # the LayerGCN steps at the end of the leg
    end_user = min(int(torch.searchsorted(full["rowptr"], torch.tensor(need, device=dev))) + 1, nU)
    nnz = int(full["rowptr"][end_user])
    assert nnz >= need, "dataset too small for the LightGCN leg"
    rp = full["rowptr"][:end_user + 1].contiguous()
    sampler = DeviceSampler(2020)
    neg = torch.empty(nnz, dtype=torch.int32, device=dev)
    cols_src = [full["users"][:nnz].contiguous(), full["items"][:nnz].contiguous(), neg]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    pair_events = []
    run_no = [0]

    def run(n_steps, bracket):
        sampler.sample_epoch_exact(nI, end_user, rp, cols_src[1], nnz, 1, neg)
        run_no[0] += 1
        uu, ii, jj = _hip.shuffle_gather(cols_src, None, seed=1000 + run_no[0], n_out=n_steps * gb)   # same key on every rank
        for s_ in range(n_steps):
            sl = slice(s_ * gb, (s_ + 1) * gb)
            if bracket:   # one whole (unmasked) product per side, bracketed by HIP events on the stream it is launched on
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                eng.a_ui.spmm(eng.ego[eng.n_local:], eng._xu[0])
                e[1].record()
                eng.a_iu.spmm(eng.ego[:eng.n_local], eng._xi[0])
                e[2].record()
                pair_events.append(e)
            eng.train_step(uu[sl], ii[sl], jj[sl])
    eng.a_ui.spmm(eng.ego[eng.n_local:], eng._xu[0])      # builds the two plans outside every timed region
    eng.a_iu.spmm(eng.ego[:eng.n_local], eng._xi[0])
    if W:
        run(W, False)
    barrier()
    t0 = time.perf_counter()
    run(K, False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    run(8, True)    # untimed extra steps only to bracket the two products with events
    barrier()
    ui_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in pair_events]))
    iu_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in pair_events]))

    # algorithmic bytes of one product: nnz * 8 + (rows + 1) * 8 + X read once + Y written once
    def spmm_bytes(csr, n_x):
        return csr.nnz * 8 + (csr.shape[0] + 1) * 8 + n_x * 256 + csr.shape[0] * 256
    alg = spmm_bytes(eng.a_ui, nI) + spmm_bytes(eng.a_iu, eng.n_local)
    ach = alg / ((ui_ms + iu_ms) * 1e-3) / 1e9
    # counter bytes of ONE layer (both products), recorded: the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command average
    # every launch of a kernel over the whole run; a layer is two spmm_rows_kernel launches plus the long rows' task / reduce
    # launches in the proportion that run made them (the LDS-streamed densest rows' kernel is not in the counter passes)
    _, pmc_source = pmc_lookup()
    e_rows, e_tasks, e_red = (pmc_lookup.entry(n) for n in ("spmm_rows_kernel<false>", "spmm_tasks_kernel<false>", "spmm_reduce_kernel"))
    layer_traffic = traffic_note = None
    if world == 1 and e_rows and e_tasks and e_red and e_rows.get("calls"):
        pairs = e_rows["calls"] / 2.0
        layer_traffic = (2 * e_rows["hbm_bytes_per_launch"] + e_tasks["calls"] / pairs * e_tasks["hbm_bytes_per_launch"]
                         + e_red["calls"] / pairs * e_red["hbm_bytes_per_launch"])
        traffic_note = (f"{pmc_source}: 2 x spmm_rows_kernel + {e_tasks['calls'] / pairs:.1f} x spmm_tasks_kernel + {e_red['calls'] / pairs:.1f} x "
                        "spmm_reduce_kernel launches per layer, bytes that enter or leave the L2s (fills from Infinity Cache or HBM): the "
                        "row gathers that miss an XCD's L2")
    leg = {
        "value": K * gb / dt, "unit": "train interactions/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "scaling": "strong" if strong else "weak", "dtype": "f32", "global_batch": gb,
        "config": {"workload": f"BASELINE configs[{2 if world == 1 else 3}]: LightGCN 3-layer d=64, synthetic {nU}-user/{nI}-item/"
                               f"{args.interactions}-interaction graph, full-graph propagation fwd+bwd per mini-batch (reference "
                               f"semantics), exact-stream sampler, dense Adam",
                   "users": nU, "items": nI, "train_interactions": n_inter_total, "global_batch": gb,
                   "sharding": f"users u%{world}; exchange = dense RCCL all-reduce of the [I, 64] item block per layer and "
                               f"direction ({2 * 3 + 1} x {nI * 256 / 1e6:.1f} MB per step), started beside the user-side product",
                   "not_computed": "rows of the LAST forward layer no batch reads and, in the FIRST backward hop, the products "
                                   "with rows of dL/dE-bar that are zero (everything outside the batch): same results "
                                   "(tests/test_gpu_fullsize.py); SKR_LIGHTGCN_DENSE=1 computes them"},
        "roofline": {"kernel": "skr_spmm_plan_run: spmm_rows_kernel (short rows, 16 B per lane) + spmm_tasks_kernel (long rows, column-"
                               "blocked tasks) + spmm_reduce_kernel; ONE layer = user-side product + item-side product",
                     "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": layer_traffic,
                     "traffic_source": traffic_note,
                     "traffic_over_algorithmic": (layer_traffic / alg) if layer_traffic else None,
                     "avg_launch_ms": ui_ms + iu_ms, "user_side_ms": ui_ms, "item_side_ms": iu_ms, "launches_averaged": len(pair_events),
                     "algorithmic_bytes_per_launch": alg, "local_nnz": eng.a_ui.nnz,
                     "row_gather_TBps": 2 * eng.a_ui.nnz * 256 / ((ui_ms + iu_ms) * 1e-3) / 1e12,
                     "plan": {"user_side": eng.a_ui.plan_info(), "item_side": eng.a_iu.plan_info()},
                     "note": "algorithmic bytes count every row of X once; the kernels move nnz * 256 B of row gathers out of L2 / "
                             "Infinity Cache (row_gather_TBps) -- DESIGN.md 4.3"},
    }
    if cpu_baseline and world == 1:
        from oracle import cpu_baseline as CB
        from skrec.recommender.LightGCN import build_adjacency_device
        adj, _ = build_adjacency_device(full["users"], full["items"], nU, nI, "pre", dev)
        t_step, cores, sample = CB.time_lightgcn_layer(adj.rowptr.cpu().numpy(), adj.col.cpu().numpy(), adj.val.cpu().numpy(), nU + nI, D)
        del adj
        leg["cpu_baseline"] = {"value": b / t_step, "unit": "train interactions/s", "cores": cores, "kind": "port", "sample": sample}
        leg["speedup_vs_cpu_baseline"] = leg["value"] / leg["cpu_baseline"]["value"]
        leg["speedup_note"] = ("against a PORT: the reference's torch-CPU op sequence restated and timed on this box's host cores, one "
                               "layer extrapolated to the step (the reference's Python cannot travel to the GPU box); a reported "
                               "baseline, not a target")
    # separately labelled large-batch variants (batch_size is a config knob, LightGCN.py:37): the same engine,
    # the same full-graph propagation per step, more interactions per step
    for bl in ([int(x) for x in args.large_batches.split(",") if x] if (world == 1 and args.large_batches) else []):
        if (1 + 3) * bl > nnz:
            continue
        eng.batch_size_cfg, gb_l = bl, bl

        def run_l(n_steps, off):
            for s_ in range(n_steps):
                sl = slice(off + s_ * gb_l, off + (s_ + 1) * gb_l)
                eng.train_step(uu_l[sl], ii_l[sl], jj_l[sl])
        sampler.sample_epoch_exact(nI, end_user, rp, cols_src[1], nnz, 1, neg)
        uu_l, ii_l, jj_l = _hip.shuffle_gather(cols_src, None, seed=77 + bl, n_out=4 * bl)
        run_l(1, 0)
        barrier()
        t0 = time.perf_counter()
        run_l(3, bl)
        barrier()
        dtl = time.perf_counter() - t0
        leg.setdefault("large_batch", {})[str(bl)] = {"value": 3 * bl / dtl, "unit": "train interactions/s", "global_batch": bl, "steps": 3,
                                                      "warmup": 1, "ms_per_step": dtl / 3 * 1e3,
                                                      "epoch_seconds_estimated": -(-n_inter_total // bl) * dtl / 3}
    eng.batch_size_cfg = b
    steps_per_epoch = -(-n_inter_total // gb)
    leg["epoch"] = {"steps": steps_per_epoch, "seconds_estimated": steps_per_epoch * dt / K,
                    "note": "steps per epoch x the measured time per step (every step is the same full-graph work)"}
    del eng
    torch.cuda.empty_cache()
    # LayerGCN on the same graph, the reference's defaults (LayerGCN.py:25-33: 4 layers, batch 2048, reg 1e-2,
    # dropout 0): full-graph propagation with the cosine layer refinement, forward and backward per mini-batch, dense Adam
    if world == 1 and not getattr(args, "no_layergcn", False):
        from skrec.parallel import ShardedLayerGCN
        bl_, Kl, Wl = 2048, max(3, min(K, 10)), 3
        if (Kl + Wl) * bl_ <= nnz:
            item0_l = (torch.rand(nI, D, generator=torch.Generator().manual_seed(8)) * 2 - 1) * (6.0 / (nI + D)) ** 0.5
            eng_l = ShardedLayerGCN(ctx, full["users"].long(), full["items"].long(), nU, nI, user0, item0_l, 4, 1e-3, 1e-2, device=dev)
            sampler.sample_epoch_exact(nI, end_user, rp, cols_src[1], nnz, 1, neg)
            uu_l, ii_l, jj_l = _hip.shuffle_gather(cols_src, None, seed=4242, n_out=(Kl + Wl) * bl_)

            def run_layer(lo, hi):
                for s_ in range(lo, hi):
                    sl = slice(s_ * bl_, (s_ + 1) * bl_)
                    eng_l.train_step(uu_l[sl], ii_l[sl], jj_l[sl])
            run_layer(0, Wl)
            barrier()
            t0 = time.perf_counter()
            run_layer(Wl, Wl + Kl)
            barrier()
            dtl = time.perf_counter() - t0
            leg["layergcn"] = {"value": Kl * bl_ / dtl, "unit": "train interactions/s", "steps": Kl, "warmup": Wl, "global_batch": bl_,
                               "ms_per_step": dtl / Kl * 1e3,
                               "config": {"workload": "LayerGCN 4-layer d=64 on the same graph, batch 2048, reg 1e-2, dropout 0 (the reference's "
                                                      "defaults): full-graph propagation + cosine layer refinement fwd+bwd per mini-batch, "
                                                      "dense Adam"},
                               "loss": [float(x) for x in eng_l.loss.tolist()]}
            del eng_l
            torch.cuda.empty_cache()
    return leg


def gru_leg(args, world, rank, dev, dist, K, W, cpu_baseline):
    """BASELINE configs[4]: GRU4RecPlus d=128, histories of 50 events, the reference's session-parallel loop
    (GRU4RecPlus.py:202-254: 128 sessions advance together, logits against the batch's own next items + 2048
    popularity^0.75 negatives, bpr_max, TF-style dense Adam over both item tables); at N > 1 the 128 parallel sessions are
    split over the ranks (skrec.recommender.GRU4RecPlus.ShardedSessionGRU: ONE compact exchange per step).  The data set
    is `--sessions` synthetic sessions (10 M in configs[4]); only the sessions the W + K steps advance are materialised
    (128 at a time, in order), the rest of an epoch repeats the same step.  PARITY UNPINNED (TensorFlow absent)."""
    from skrec import _hip
    from skrec.parallel import DistContext
    from skrec.recommender.GRU4RecPlus import SessionGRU, ShardedSessionGRU
    L_, st = _hip.lib(), _hip.stream
    nI, d, b, n_s, T = args.items, 128, 128, 2048, 50
    assert b % world == 0
    g = torch.Generator().manual_seed(5)
    E_in = torch.nn.init.trunc_normal_(torch.empty(nI, d), std=0.01, a=-0.02, b=0.02, generator=g)
    E_out = torch.nn.init.trunc_normal_(torch.empty(nI, d), std=0.01, a=-0.02, b=0.02, generator=g)
    lim_g, lim_c = (6.0 / (d + 3 * d)) ** 0.5, (6.0 / (d + 2 * d)) ** 0.5
    cells = [((torch.rand(2 * d, 2 * d, generator=g) * 2 - 1) * lim_g, torch.ones(2 * d), (torch.rand(2 * d, d, generator=g) * 2 - 1) * lim_c,
              torch.zeros(d))]
    net_args = (E_in, cells, E_out, torch.zeros(nI), "tanh", "linear", "bpr_max", 1.0, 0.0, 1e-3, dev)
    net = ShardedSessionGRU(DistContext(rank, world), *net_args) if world > 1 else SessionGRU(*net_args)
    lo, hi = net.slots(b) if world > 1 else (0, b)
    # item popularity Zipf(0.9); a session = T items drawn from it; negatives from popularity^0.75 (numpy uniforms, as the
    # reference draws them; the search runs on the device: skr_pop_sample)
    pop = 1.0 / np.arange(1, nI + 1) ** 0.9
    pop = pop[np.random.default_rng(3).permutation(nI)]
    cs_items = torch.from_numpy(np.cumsum(pop) / pop.sum()).to(dev)
    p75 = np.cumsum(pop ** 0.75)
    cs_neg = torch.from_numpy(p75 / p75[-1]).to(dev)
    n_blocks = (W + K + 8 + T - 2) // (T - 1) + 1                 # groups of b sessions the steps walk through
    items = torch.empty(n_blocks * b * T, dtype=torch.int32, device=dev)
    _hip.check(L_.skr_pop_sample(_hip.ptr(cs_items), nI, None, 1234, items.numel(), _hip.ptr(items), st()))
    items = items.view(n_blocks, b, T)
    np.random.seed(77)                                             # the same negatives on every rank
    state = {"s": net.zero_states(hi - lo), "step": 0}

    kblk = max(1, min(64, int(os.environ.get("SKR_ADAM_BLOCK", "32"))))
    net.opt.cold_timing = cold_log = []

    def run(n_steps):
        """the steps are prepared kblk at a time, as GRU4RecPlus.train_epoch does: inputs / targets gathered for the block, the
        negatives' uniforms drawn in the reference's order (one np.random.rand(n_sample) per step) and searched on the
        device, the dense TF-Adam blocked over the k steps (SessionGRU.begin_block)"""
        done = 0
        while done < n_steps:
            k = min(kblk, n_steps - done)
            s0 = state["step"]
            bt = [divmod(s0 + j, T - 1) for j in range(k)]
            xs = torch.stack([items[blk, :, t] for blk, t in bt]).contiguous()
            u = torch.from_numpy(np.random.rand(k * n_s)).to(dev)
            neg = torch.empty(k * n_s, dtype=torch.int32, device=dev)
            _hip.check(L_.skr_pop_sample(_hip.ptr(cs_neg), nI, _hip.ptr(u), 0, k * n_s, _hip.ptr(neg), st()))
            ys = torch.cat([torch.stack([items[blk, :, t + 1] for blk, t in bt]), neg.view(k, n_s)], dim=1).contiguous()
            if kblk > 1:
                net.begin_block(xs, ys)
            for j, (blk, t) in enumerate(bt):
                if t == 0 and s0 + j > 0:                          # the b sessions ended together: fresh states
                    state["s"] = net.zero_states(hi - lo)
                state["s"] = net.train_step(xs[j], ys[j], state["s"])
            state["step"] += k
            done += k
        net.end_blocks()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    run(W)
    barrier()
    t0 = time.perf_counter()
    run(K)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss = float(net.loss.cpu())
    opt = net.opt
    torch.cuda.synchronize()
    if kblk > 1 and cold_log:
        # the optimiser's large kernel: the cold pass of the blocked TF-Adam, ONE sweep per kblk steps over every 64-float
        # block no step of the block names (20 B per cold parameter: p, m, v read, m, v written), on a side stream beside the
        # steps' launches.  The step itself is a chain of small latency-bound launches (GRU cell, logits, their gradients, the
        # hot rows' Adam): no single kernel of it is near a roofline.
        ms_ = float(np.mean([a_.elapsed_time(z_) for a_, z_, kk_ in cold_log if kk_ == kblk] or [a_.elapsed_time(z_) for a_, z_, _ in cold_log]))
        n_hot = int((opt._blk_tag == opt._blk_serial).sum())
        cold_bytes = float(opt.flat.numel() - 64 * n_hot) * 20.0
        roof = {"kernel": f"adam_cold_rows_kernel<4> (TF-arithmetic dense Adam over [E_in | E_out | b_out | GRU kernels] blocked over {kblk} "
                          f"steps: the rows no step of the block names, one pass per block on a side stream)",
                "bound": "hbm", "achieved": cold_bytes / (ms_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": cold_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms_, "launches_averaged": len(cold_log),
                "algorithmic_bytes_per_launch": cold_bytes, "hot_blocks": n_hot, "adam_block": kblk,
                "ms_per_step_of_the_pass": ms_ / kblk, "share_of_step": (ms_ / kblk) / (dt / K * 1e3),
                "overlapped_with_step_kernels": True}
    else:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for a_, z_ in ev:
            a_.record()
            _hip.check(L_.skr_adam_step_tf(_hip.ptr(opt.flat), _hip.ptr(opt.grad), _hip.ptr(opt.m), _hip.ptr(opt.v), opt.flat.numel(), 0.0, 0.9,
                                           0.999, 1e-8, opt.t + 1, 1, _hip.ptr(opt.touch), st()))
            z_.record()
        torch.cuda.synchronize()
        adam_ms = float(np.mean([a_.elapsed_time(z_) for a_, z_ in ev]))
        adam_bytes = opt.flat.numel() * 28.0
        roof = {"kernel": "adam_kernel (dense TF-semantics Adam over [E_in | E_out | b_out | GRU kernels], one launch per step)",
                "bound": "hbm", "achieved": adam_bytes / (adam_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": adam_bytes / (adam_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": adam_ms,
                "launches_averaged": len(ev), "algorithmic_bytes_per_launch": adam_bytes, "share_of_step": adam_ms / (dt / K * 1e3)}
    leg = {"value": K * b / dt, "unit": "train events/s (one event = one session advancing one item)", "n_gpus": world, "steps": K, "warmup": W,
           "ms_per_step": dt / K * 1e3, "scaling": "strong", "dtype": "f32", "parity": "unpinned (TensorFlow 1.14 absent; DESIGN.md 7)",
           "last_loss": loss,
           "config": {"workload": f"BASELINE configs[4]: GRU4RecPlus d={d}, {args.sessions} synthetic sessions of {T} events over {nI} items, "
                                  f"session-parallel loop, {b} parallel sessions + {n_s} popularity^0.75 negatives, bpr_max, dense Adam",
                      "parallel_sessions": b, "sessions_per_rank": hi - lo, "steps_per_epoch": args.sessions * (T - 1) // b,
                      "adam_block": kblk,
                      "sharding": f"the {b} parallel sessions split over {world} ranks; one compact all-gather per step "
                                  f"({(b + n_s) * (d + 1) + b * d + 3 * 2 * d * d + 3 * d + 1} floats per rank), summed in rank order"},
           "roofline": roof}
    if cpu_baseline and world == 1:
        from oracle import gru4rec as G
        o = G.GRU4RecOracle(E_in.numpy(), [tuple(w.numpy() for w in cells[0])], E_out.numpy(), np.zeros(nI, np.float32), loss="bpr_max",
                            bpr_reg=1.0, reg=0.0, lr=1e-3)
        st_o = [torch.zeros(b, d)]
        rng = np.random.default_rng(1)
        ts = []
        for s_ in range(4):
            X = rng.integers(0, nI, b).astype(np.int32)
            Y = rng.integers(0, nI, b + n_s).astype(np.int32)
            t1 = time.perf_counter()
            _, st_o = o.train_step(X, Y, st_o)
            ts.append(time.perf_counter() - t1)
        t_step = float(np.mean(ts[1:]))
        leg["cpu_baseline"] = {"value": b / t_step, "unit": "train events/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"3 steps of the torch-CPU restatement of the graph (oracle/gru4rec.py: autograd + dense "
                                         f"TF-style Adam) at the same sizes: {t_step * 1e3:.1f} ms per step"}
    del net
    torch.cuda.empty_cache()
    return leg


def pmc_lookup():
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary (separate --pmc FETCH_SIZE / WRITE_SIZE passes of
    bench.py: tools/profile_bench.sh -> tools/summarize_profiles.py -> profiles/<round>_pmc_summary.json).  Counters cannot
    be read from inside the run, so the figure is a RECORDED one: the second value names the file and the command line
    it was captured with."""
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary.json")))[-1]
        doc = json.load(open(latest))
        pmc = doc.get("kernels", {})
        source = f"{os.path.relpath(latest, REPO)} (recorded; command: {doc.get('command', 'bench.py, see tools/profile_bench.sh')})"
    except Exception:
        pmc, source = {}, None

    def entry(name):
        return pmc.get(name)
    pmc_lookup.entry = entry

    def lookup(key):
        """`kernel` or `kernel@state` (e.g. @epoch3: the launches of the third whole epoch)"""
        base, _, state = key.partition("@")
        for name, ent in pmc.items():
            n_base, _, n_state = name.partition("@")
            if n_base.startswith(base) and n_state == state and "hbm_bytes_per_launch" in ent:
                return ent["hbm_bytes_per_launch"]
        return None
    return lookup, source


def cold_roofline(n_par, n_hot_blocks, kblk, cold_ms, alone_ms, overlapped, traffic, traffic_source):
    """Roofline entry of adam_cold_rows_kernel, the dominant train kernel in blocked mode: ONE pass per k-step block over
    every 64-float block no batch of the block touches.  `cold_ms` = [(ms, steps applied, phase)] of EVERY cold pass of the
    run's pre-steps, warm-up and timed steps (HIP events on the side stream the pass is launched on) -- the same set
    rocprofv3 --stats of this command averages over.
    Algorithmic bytes per LAUNCH: p, m, v read (12 B) and m, v written (8 B) per cold parameter = 20 B; a
    launch applies up to k optimiser steps.  SURVEY 8(d)'s per-step figure (28 B per parameter and step) is what the pass
    replaces k times over: `dense_equivalent_GBps`, not `achieved`.  The kernel does not write back rows whose moments did
    not change, so it MOVES fewer bytes than the algorithmic count: `achieved_by_traffic` is the counter bytes over the
    same time."""
    ms = float(np.mean([t for t, _, _ in cold_ms]))
    steps = float(np.mean([k for _, k, _ in cold_ms]))
    full = [t for t, k, _ in cold_ms if k == kblk]
    cold_par = float(n_par - 64 * n_hot_blocks)
    cold_bytes = cold_par * 20.0
    ach = cold_bytes / (ms * 1e-3) / 1e9
    r = {"kernel": f"adam_cold_rows_kernel<4> (temporally blocked dense Adam: up to {kblk} zero-gradient steps per pass over the "
                   f"blocks no batch of the block touches; rows at rest skip the square root and divisions; bit-identical to a "
                   f"dense launch per step)",
         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
         "achieved_by_traffic": (traffic / (ms * 1e-3) / 1e9) if traffic else None,
         "frac_by_traffic": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
         "avg_launch_ms": ms, "launches_averaged": len(cold_ms),
         "launches_by_phase": {ph: sum(1 for _, _, p_ in cold_ms if p_ == ph) for ph in sorted({p_ for _, _, p_ in cold_ms})},
         "avg_ms_full_k_launches": float(np.mean(full)) if full else None, "full_k_launches": len(full),
         "timed_region_launches": [{"ms": t, "optimizer_steps": k} for t, k, p_ in cold_ms if p_ == "timed"],
         "algorithmic_bytes_per_launch": cold_bytes, "algorithmic_bytes_per_parameter": 20.0, "cold_parameters": cold_par,
         "hot_blocks": n_hot_blocks, "optimizer_steps_per_launch": steps, "adam_block": kblk,
         "overlapped_with_step_kernels": overlapped,
         "dense_equivalent_GBps": float(n_par) * 28.0 * steps / (ms * 1e-3) / 1e9}
    if alone_ms:
        r["alone"] = {"avg_launch_ms": alone_ms, "achieved": cold_bytes / (alone_ms * 1e-3) / 1e9,
                      "frac": cold_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": f"the same pass ({kblk} steps) with nothing beside it (copies of the buffers after the timed region, "
                              "same tags): `achieved` / `frac` above are the live figures, with the pass held to 4 workgroups per CU "
                              "underneath the step kernels"}
    return r


def bprmf_large_batch_leg(bl, dev, nU, nI, prefix, sample_slice):
    """Separately labelled leg (b = 1 024 is the reference's default, batch_size is its config knob, BPRMF.py:28):
    configs[1]'s job at batch `bl`, N = 1, on tables of its own.  Two forms of the same step are timed:
      * `blocked`: what BPRMF.train_epoch runs -- one launch per step (skr_bpr_fused_step) with the dense Adam blocked over
        k = min(32, 2^20 / (5 bl)) steps (the fused workspace has 2^20 row slots);
      * `dense`: skr_bpr_step_spread + one dense skr_adam_step per step, each launch bracketed by HIP events: the HBM-bound
        regime of K1 (1 564 B per interaction) and K2 (28 B per parameter and step)."""
    from skrec import _hip
    from skrec.recommender.base import DenseAdam
    from skrec.recommender.fused import FusedBlocks
    L, st, S = _hip.lib(), _hip.stream, _hip.SKR_LOSS_SLOTS
    n_par = nU * D + nI * D + nI
    flat = torch.zeros(n_par, device=dev)
    flat[:(nU + nI) * D].normal_(0.0, 0.01, generator=torch.Generator(device=dev).manual_seed(99))
    opt = DenseAdam(flat, lr=1e-3, track_touch=True)
    k = max(1, min(32, (1 << 20) // (5 * bl)))
    nb_w, nb_t = 2, 8
    n_dense = 12
    sl = prefix(((nb_w + nb_t) * k + n_dense) * bl, 0)
    neg = sample_slice(sl)
    uu, ii, jj = _hip.shuffle_gather([sl["users"], sl["items"], neg], None, seed=4242 + bl, n_out=((nb_w + nb_t) * k + n_dense) * bl)
    pu, pi, pj = uu.data_ptr(), ii.data_ptr(), jj.data_ptr()
    loss = torch.zeros(((nb_w + nb_t) * k + n_dense, S, 2), device=dev)
    res = {"batch": bl, "unit": "train interactions/s"}
    if k >= 2:
        fb = FusedBlocks(opt, 0, nU, nU + nI, 1e-3)
        fb.run_blocks(pu, pi, pj, nb_w, k, bl, loss.data_ptr(), 8 * S)
        opt.end_blocks()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = 4 * nb_w * k * bl
        fb.run_blocks(pu + o, pi + o, pj + o, nb_t, k, bl, loss.data_ptr() + 8 * S * nb_w * k, 8 * S)
        opt.end_blocks()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["blocked"] = {"value": nb_t * k * bl / dt, "ms_per_step": dt / (nb_t * k) * 1e3, "steps": nb_t * k, "adam_block": k,
                          "step_launches": "1: skr_bpr_fused_step + one cold pass per block on the side stream"}
        del fb
    # dense form, launch by launch
    P = {n_: t_.data_ptr() for n_, t_ in dict(flat=flat, grad=opt.grad, m=opt.m, v=opt.v, touch=opt.touch).items()}
    pU, pV, pb = P["flat"], P["flat"] + 4 * nU * D, P["flat"] + 4 * (nU + nI) * D
    gU, gV, gb = P["grad"], P["grad"] + 4 * nU * D, P["grad"] + 4 * (nU + nI) * D
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(n_dense)]
    for t_ in ev:
        for e_ in t_:
            e_.record()
    o0 = (nb_w + nb_t) * k
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(n_dense):
        o = 4 * (o0 + s_) * bl
        ev[s_][0].record()
        rc = L.skr_bpr_step_spread(pU, pV, pb, pU, pV, pu + o, pi + o, pj + o, bl, 1.0, 1e-3, 1.0, gU, gV, gb, gU, gV,
                                   loss.data_ptr() + 8 * S * (o0 + s_), P["touch"], P["grad"], st())
        ev[s_][1].record()
        opt.t += 1
        rc |= L.skr_adam_step(P["flat"], P["grad"], P["m"], P["v"], n_par, 1e-3, 0.9, 0.999, 1e-8, opt.t, 1, P["touch"], st())
        ev[s_][2].record()
        if rc:
            _hip.check(rc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bpr_ms = float(np.mean([a.elapsed_time(b_) for a, b_, _ in ev[2:]]))
    adam_ms = float(np.mean([b_.elapsed_time(c) for _, b_, c in ev[2:]]))
    k1 = 1564.0 * bl / (bpr_ms * 1e-3) / 1e9
    k2 = 28.0 * n_par / (adam_ms * 1e-3) / 1e9
    res["dense"] = {"value": n_dense * bl / dt, "ms_per_step": dt / n_dense * 1e3, "steps": n_dense,
                    "step_launches": "2: skr_bpr_step_spread + skr_adam_step (dense, one launch per step)",
                    "roofline_K1": {"kernel": "bpr_step_kernel", "bound": "hbm", "achieved": k1, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": k1 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": bpr_ms,
                                    "algorithmic_bytes_per_launch": 1564.0 * bl, "launches_averaged": n_dense - 2},
                    "roofline_K2": {"kernel": "adam_kernel (dense, touch bytes skip the reads of zero gradients)", "bound": "hbm",
                                    "achieved": k2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2 / HBM_PEAK_GBS, "traffic": None,
                                    "avg_launch_ms": adam_ms, "algorithmic_bytes_per_launch": 28.0 * n_par, "launches_averaged": n_dense - 2}}
    best = max((res[k_]["value"], k_) for k_ in ("blocked", "dense") if k_ in res)
    res["value"], res["form"] = best
    del opt, flat
    torch.cuda.empty_cache()
    return res


def step_roofline(b, kblk, under_log, alone_log, traffic, traffic_source):
    """Roofline entry of bpr_fused_step_kernel, the launch the main stream of a BPRMF step consists of (one per step; the
    cold pass runs beside it on the side stream).  Algorithmic bytes: 1 564 B per interaction.  Timed by
    HIP events on the stream the launches go to: around the k step launches of a block and around its end launch
    (bpr_fused_end_kernel: the block's hot rows back into the dense tables), for blocks of the third whole epoch (under
    the cold pass) and for blocks with the cold pass in front of them on the same stream (alone on the chip)."""
    def ms(log):
        st_ = float(np.mean([e0.elapsed_time(e1) / k_ for e0, e1, _, k_ in log]))
        en_ = float(np.mean([e1.elapsed_time(e2) for _, e1, e2, _ in log]))
        return st_, en_
    us, end_us = (x * 1e3 for x in ms(under_log))
    alg = 1564.0 * b
    ach = alg / (us * 1e-6) / 1e9
    r = {"kernel": "bpr_fused_step_kernel (one launch per step: BPR forward / backward of the batch + the lazily evaluated Adam "
                   "updates of the rows it names; bound by the exact catch-up arithmetic and launch latency, not by HBM)",
         "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
         "traffic_over_algorithmic": (traffic / alg) if traffic else None,
         "avg_launch_us": us, "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_interaction": 1564.0, "interactions_per_launch": b,
         "state": "under the cold pass (third whole epoch, blocks 257-768); start-to-start of consecutive launches on their stream",
         "blocks_averaged": len(under_log), "launches_averaged": len(under_log) * kblk,
         "end_launch_us_per_block": end_us, "end_launch_us_per_step": end_us / kblk}
    if alone_log:
        a_us, a_end = (x * 1e3 for x in ms(alone_log))
        r["alone"] = {"avg_launch_us": a_us, "end_launch_us_per_block": a_end, "achieved": alg / (a_us * 1e-6) / 1e9,
                      "frac": alg / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "blocks_averaged": len(alone_log),
                      "note": "the same launches with the block's cold pass in front of them on the same stream"}
    return r


def eval_leg(args, world, rank, dev, dist, U, V, bias, ds, nU, nI, out):
    """fused GEMM(MFMA) + train mask + top-K + metrics over a block of this rank's users -> out["eval"], out["roofline_eval*"]"""
    from skrec import _hip
    L, st = _hip.lib(), _hip.stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    # ---- eval leg: fused GEMM(MFMA)+mask+top-K over a block of this rank's users -------------------
    if not args.no_eval:
        ne = min(args.eval_users, nU)
        users_e = torch.arange(ne, dtype=torch.int32, device=dev)
        ids = torch.empty((ne, args.top_k), dtype=torch.int32, device=dev)
        rows = torch.empty((ne, 2 * args.top_k), dtype=torch.float32, device=dev)
        sums = torch.zeros(2 * args.top_k, dtype=torch.float64, device=dev)
        ws = int(L.skr_eval_fused_workspace(ne, args.top_k))
        work = torch.empty(ws, dtype=torch.uint8, device=dev)
        test_ptr = torch.arange(nU + 1, dtype=torch.long, device=dev)   # one held-out item per user
        margs = _hip.metric_array([2, 4])                               # Recall (= HR on leave-one-out), NDCG

        def eval_once():
            _hip.check(L.skr_eval_fused_topk(_hip.ptr(U), _hip.ptr(users_e), ne, _hip.ptr(V), _hip.ptr(bias), nI, D,
                                             _hip.ptr(ds["rowptr"]), _hip.ptr(ds["items"]), args.top_k, _hip.ptr(ids), None,
                                             _hip.ptr(work), ws, st()))
            _hip.check(L.skr_rank_metrics(_hip.ptr(ids), ne, args.top_k, _hip.ptr(users_e), _hip.ptr(test_ptr),
                                          _hip.ptr(ds["test_item"]), margs, 2, _hip.ptr(rows), _hip.ptr(sums), st()))
        wu = min(ne, 4096)   # short warm-up launch
        _hip.check(L.skr_eval_fused_topk(_hip.ptr(U), _hip.ptr(users_e), wu, _hip.ptr(V), _hip.ptr(bias), nI, D,
                                         _hip.ptr(ds["rowptr"]), _hip.ptr(ds["items"]), args.top_k, _hip.ptr(ids), None,
                                         _hip.ptr(work), ws, st()))
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        eval_once()
        e1.record()
        barrier()
        te = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([te], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            te = float(tmax)
        flops = 2.0 * ne * nI * D
        k_ms = e0.elapsed_time(e1)
        tf = flops / (k_ms * 1e-3) / 1e12
        hr = (sums.cpu().numpy() / ne).reshape(2, args.top_k)[:, -1]
        out["eval"] = {"users_per_sec": ne * world / te, "users": ne * world, "top_k": args.top_k, "seconds": te,
                       f"HR@{args.top_k}": float(hr[0]), f"NDCG@{args.top_k}": float(hr[1])}
        mode = os.environ.get("SKR_FUSED_MODE", "f16x2")
        if mode == "fp32":
            out["roofline_eval"] = {"kernel": "fused_topk_kernel_v3 (FP32 MFMA GEMM + mask + top-K)", "bound": "mfma",
                                    "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                                    "traffic": None, "avg_launch_ms": k_ms, "algorithmic_flop_per_launch": flops}
        else:
            # every fp32 product is formed from `products` half-precision MFMA products with fp32 accumulation (f16x2: three
            # fp16 x fp16, bf16x3: six bf16 x bf16), so the matrix pipe issues that many times the algorithmic flops: the peak for
            # THIS arithmetic is the dense 16-bit MFMA peak (the same for fp16 and bf16) / products
            products = 3 if mode == "f16x2" else 6
            kname = {"f16x2": "fused_topk_kernel_v7 (v_mfma_f32_16x16x32_f16, steps of 16 items, threshold tests between the MFMAs, one item "
                              "ring per workgroup; fp32 operands scaled by a power of two per table and split into 2 fp16 pieces, 3 fp16 "
                              "MFMAs per fp32 product, fp32 accumulate; per-user guard, rejected rows recomputed by the bf16x3 kernel in "
                              "the same call; GEMM + mask + top-K)",
                     }.get(
                mode, "fused_topk_kernel_v6 (v_mfma_f32_16x16x32_bf16, steps of 16 items, threshold tests between the MFMAs, "
                      "one item ring per workgroup; ")
            if mode != "f16x2":
                kname += "fp32 operands split into 3 bf16 pieces, 6 bf16 MFMAs per fp32 product, fp32 accumulate; GEMM + mask + top-K)"
            peak = MFMA_BF16_PEAK_TF / products
            out["roofline_eval"] = {"kernel": kname, "bound": "mfma",
                                    "achieved": tf, "peak": peak, "unit": "TFLOP/s", "unit_note": "fp32-equivalent (algorithmic 2*B*I*64 flop)",
                                    "frac": tf / peak, "traffic": None, "avg_launch_ms": k_ms,
                                    "algorithmic_flop_per_launch": flops, "mfma_products_per_fp32_product": products,
                                    "mfma_issued_tflops": products * tf, "mfma_peak_tflops": MFMA_BF16_PEAK_TF,
                                    "frac_if_priced_as_bf16x3": tf / (MFMA_BF16_PEAK_TF / 6.0),
                                    "accuracy": "error vs float64 relative to sum|u_i v_i| (tools/fused_accuracy.py): f16x2 max 2.1e-7 / "
                                                "mean 2.8e-8, bf16x3 2.6e-7 / 3.1e-8, FP32-MFMA kernel 3.5e-7 / 5.1e-8"}
            if mode == "f16x2":
                import ctypes
                n_rej = ctypes.c_int32(-1)
                _hip.check(_hip.lib().skr_eval_fused_rejected(ctypes.byref(n_rej), _hip.stream()))
                out["roofline_eval"]["rows_rejected_by_the_guard_in_the_last_call"] = n_rej.value
            if world == 1:   # the FP32-MFMA kernel on the same inputs, for comparison
                mode_was = os.environ.get("SKR_FUSED_MODE")
                os.environ["SKR_FUSED_MODE"] = "fp32"
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ids_bf = ids.clone()
                f0.record()
                eval_once()
                f1.record()
                torch.cuda.synchronize()
                if mode_was is None:
                    del os.environ["SKR_FUSED_MODE"]
                else:
                    os.environ["SKR_FUSED_MODE"] = mode_was
                ms32 = f0.elapsed_time(f1)
                tf32 = flops / (ms32 * 1e-3) / 1e12
                out["roofline_eval_fp32"] = {"kernel": "fused_topk_kernel_v3 (FP32 MFMA)", "bound": "mfma", "achieved": tf32,
                                             "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf32 / MFMA_F32_PEAK_TF,
                                             "avg_launch_ms": ms32,
                                             "top_k_lists_identical_to_the_default_kernel": float((ids_bf == ids).all(dim=1).float().mean())}



def cpu_leg(args, U, V, bias, ds, nU, nI, b, out):
    """the reference's way on this box's host cores (rank 0, N = 1 only; a bounded sample) -> out["cpu_baseline"]"""
    world = 1
    # ---- CPU baseline (rank 0, N = 1 only): the reference's way on this box's host cores -----------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline as CB
        rp = ds["rowptr"].cpu().numpy()
        it = ds["items"].cpu().numpy()
        us = ds["users"].cpu().numpy()
        rate_s, kind_s, sample_s = CB.time_sampler(nI, rp, it)
        nb = 16
        rng = np.random.default_rng(3)
        idx = rng.integers(0, len(it), nb * b)
        neg = rng.integers(0, nI, nb * b).astype(np.int32)
        t_step, cores = CB.time_bprmf_steps(nU, nI, D, us[idx], it[idx], neg, b, steps=nb - 2, warmup=2)
        cpu_value = b / (t_step + b / rate_s)
        out["cpu_baseline"] = {"value": cpu_value, "unit": "train interactions/s", "cores": cores,
                               "kind": "port", "sample": f"{nb - 2} BPRMF steps of {b} at full table size with the "
                               f"reference's torch-CPU op sequence ({t_step * 1e3:.1f} ms/step) + sampler share at "
                               f"{rate_s / 1e6:.2f} M negatives/s ({kind_s}: {sample_s})"}
        if not args.no_eval:
            ev_rate, ev_kind = CB.time_eval_batches(U.cpu().numpy(), V.cpu().numpy(), bias.cpu().numpy(), rp, it,
                                                    ds["test_item"].cpu().numpy(), np.arange(256, dtype=np.int32),
                                                    K=args.top_k)
            out["cpu_baseline"]["eval_users_per_sec"] = ev_rate
            out["cpu_baseline"]["eval_kind"] = ev_kind
            out["cpu_baseline"]["eval_sample"] = "4 batches of 64 users: torch-CPU matmul + numpy masking + native top-K (4 threads)"


def finish(args, world, rank, dev, dist, full, V, bias, out):
    """replica check (N > 1), the secondary LightGCN leg, the ONE JSON line"""
    if world > 1:
        # the replicated item table must be bit-identical on every rank after the timed steps
        chk = torch.stack([V.double().sum(), bias.double().sum(), V.view(torch.int32).long().sum().double()])
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        out["config"]["item_table_replicas_identical"] = bool(torch.equal(lo_, hi_))
    if not args.no_lightgcn:
        out["lightgcn"] = lightgcn_leg(args, world, rank, dev, dist, full, args.lightgcn_steps, args.lightgcn_warmup,
                                       not args.no_cpu_baseline)
    if not args.no_gru and 128 % world == 0:
        out["gru4rec"] = gru_leg(args, world, rank, dev, dist, args.gru_steps, 10, not args.no_cpu_baseline)
    out.update(dist_info(world, dist), compute_stream=compute_stream_note())
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()




def bprmf_strong(args, world, rank, dev, dist, full, ds):
    """N > 1, --scaling strong: the SAME training job on more GPUs.  Global batch = --batch; every rank replays the same
    exact-stream negatives and the same shuffle of the same epoch slice, keeps the triples of its own users (u % N), and
    the item gradient's touched rows are exchanged once per step -- skrec.parallel.ShardedBPRMF.train_block, the engine
    the drop-in BPRMF runs under torchrun, with the temporally blocked Adam (every rank sees the whole global block, so its
    hot rows are known without communication)."""
    from skrec import _hip
    from skrec.parallel import DistContext, ShardedBPRMF
    from skrec.utils.py.random import DeviceSampler
    ctx = DistContext(rank, world)
    nUg, nI, b, K, W = args.users, args.items, args.batch, args.steps, args.warmup
    n_inter_total = int(full["rowptr"][-1])
    user0 = torch.randn(nUg, D, generator=torch.Generator().manual_seed(2021)) * 0.01
    item0 = torch.randn(nI, D, generator=torch.Generator().manual_seed(7)) * 0.01         # identical on every rank
    eng = ShardedBPRMF(ctx, user0, item0, torch.zeros(nI), 1e-3, 1e-3, dev)
    del user0
    eng.optimizer.t = int(args.start_step)
    kblk = eng.adam_block
    timing = eng.optimizer.cold_timing = []

    def prefix(n_need, start_user):
        lo = int(full["rowptr"][start_user])
        end_user = min(int(torch.searchsorted(full["rowptr"], torch.tensor(lo + n_need, device=dev))) + 1, nUg)
        hi = int(full["rowptr"][end_user])
        return dict(rowptr=(full["rowptr"][start_user:end_user + 1] - lo).contiguous(), users=full["users"][lo:hi].contiguous(),
                    items=full["items"][lo:hi].contiguous(), n_users=end_user - start_user, nnz=hi - lo, end_user=end_user)
    n_pre = max(0, min(args.pre_steps, (n_inter_total // b - K - W) // 2))
    pre = prefix(n_pre * b, 0) if n_pre > 0 else None
    warm = prefix(W * b, pre["end_user"] if pre else 0) if W > 0 else None
    timed = prefix(K * b, warm["end_user"] if warm else (pre["end_user"] if pre else 0))
    assert timed["nnz"] >= K * b, "dataset too small for --steps"
    sampler = DeviceSampler(2020)
    shuffles = [0]

    def run_slice(sl, n_steps):
        neg = torch.empty(sl["nnz"], dtype=torch.int32, device=dev)
        sampler.sample_epoch_exact(nI, sl["n_users"], sl["rowptr"], sl["items"], sl["nnz"], 1, neg)
        shuffles[0] += 1
        uu, ii, jj = _hip.shuffle_gather([sl["users"], sl["items"], neg], None, seed=11 * 1000003 + shuffles[0], n_out=n_steps * b)
        bounds = [(s_ * b, (s_ + 1) * b) for s_ in range(n_steps)]
        losses = torch.zeros((n_steps, 2), dtype=torch.float32, device=dev)
        if kblk > 1:
            for s0 in range(0, n_steps, kblk):
                eng.train_block(uu, ii, jj, bounds[s0:s0 + kblk], losses[s0:s0 + kblk])
        else:
            for a_, z_ in bounds:
                eng.train_step(uu[a_:z_], ii[a_:z_], jj[a_:z_])

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    marks = [0]
    if pre is not None:
        run_slice(pre, n_pre)
    marks.append(len(timing))
    if W > 0:
        run_slice(warm, W)
    marks.append(len(timing))
    barrier()
    t0 = time.perf_counter()
    run_slice(timed, K)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    marks.append(len(timing))
    sparse = eng.exchange == "sparse" or (eng.exchange == "auto" and world * 2 * b * 66 < nI * 65)
    out = {
        "metric": "train interactions/sec + eval users/sec (HR@10/NDCG@10) at 1/2/4/8 MI355X",
        "value": K * b / dt, "unit": "train interactions/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: BPRMF d=64, synthetic {args.users}-user/{args.items}-item/"
                               f"{args.interactions}-interaction (MovieLens-shaped), exact-stream sampler + fused BPR "
                               f"step + dense Adam; eval = fused MFMA top-{args.top_k}",
                   "users": args.users, "items": args.items, "train_interactions": n_inter_total,
                   "global_batch": b, "batch_per_gpu": f"~{b // world} (this rank's users' triples of every global batch)",
                   "engine": "skrec.parallel.ShardedBPRMF.train_block",
                   "sharding": f"users u%{world}, item table replicated + RCCL "
                               + (f"all-gather of the touched item-gradient rows per step ({2 * b * 66 * 4 / 1e6:.2f} MB per rank)"
                                  if sparse else "all-reduce of the dense item gradient per step (26 MB)")},
    }
    if kblk > 1 and timing:
        phase = lambda i: "pre" if i < marks[1] else ("warmup" if i < marks[2] else "timed")   # noqa: E731
        cold_ms = [(a.elapsed_time(z), k, phase(i)) for i, (a, z, k) in enumerate(timing)]
        opt = eng.optimizer
        n_hot = int((opt._blk_tag == opt._blk_serial).sum())
        pmc_traffic, pmc_source = pmc_lookup()
        out["roofline"] = cold_roofline(eng.flat.numel(), n_hot, kblk, cold_ms, None, True, None, pmc_source)
    if not args.no_eval:
        eval_leg(args, world, rank, dev, dist, eng.user_rows, eng.item_rows, eng.item_bias, ds, eng.n_local, nI, out)
    finish(args, world, rank, dev, dist, full, eng.item_rows, eng.item_bias, out)


def dist_info(world, dist):
    """what the collectives of this run really were: rccl_ranks = the group size torch.distributed saw on backend "nccl"
    (= RCCL on ROCm); 0 for a gloo rehearsal; 1 for a single process (no group)"""
    if world > 1 and dist.is_initialized():
        be = dist.get_backend()
        return {"dist_backend": be, "rccl_ranks": dist.get_world_size() if be == "nccl" else 0,
                "gpus_visible": torch.cuda.device_count()}
    return {"dist_backend": None, "rccl_ranks": 1, "gpus_visible": torch.cuda.device_count()}


def compute_stream_note():
    return "own" if torch.cuda.current_stream() != torch.cuda.default_stream() else "null"


def launch_command(n_ranks, argv, port, script=None):
    """the driver's own N > 1 form: one fresh process per GPU under torch.distributed.run, rendezvous on 127.0.0.1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_ranks, argv, script=None, env=None):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process becomes the launcher.  It has made NO
    HIP call (torch.cuda.device_count() counts devices without initialising one) and makes none: it starts N fresh children
    (never an exec of a process that has touched the GPU), relays their output -- rank 0 prints the JSON line -- and returns
    their exit status.  With fewer visible GPUs than ranks RCCL cannot form the group (one process per GPU), so unless
    SKR_DIST_BACKEND says otherwise the ranks rehearse on gloo, sharing the card(s); the line then carries rccl_ranks = 0."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    n_dev = torch.cuda.device_count()
    if n_dev < n_ranks:
        if "SKR_DIST_BACKEND" not in env:
            print(f"[bench] {n_ranks} ranks but {n_dev} visible GPU(s): rehearsing on gloo (SKR_DIST_BACKEND=gloo); nothing in "
                  "this run is an RCCL / xGMI measurement", file=sys.stderr)
            env["SKR_DIST_BACKEND"] = "gloo"
        # processes time-slicing one card: every cross-stream wait becomes milliseconds; keep each on one stream
        env.setdefault("SKR_ADAM_OVERLAP", "0")
        env.setdefault("SKR_SAMPLER_ONE_STREAM", "1")
    cmd = launch_command(n_ranks, argv, port, script)
    # the ranks get a process group of their own, and a deadline: a rank that never returns from a collective (a peer that
    # died, a rendezvous that cannot complete) must not hold the caller for ever.  Only THIS group is ever signalled.
    deadline = float(env.get("SKR_BENCH_RANKS_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    import signal
    import threading
    timed_out = []

    def expire():
        timed_out.append(True)
        print(f"[bench] the {n_ranks} ranks did not finish within {deadline:.0f} s: ending their process group", file=sys.stderr)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                return
            time.sleep(5 if sig == signal.SIGTERM else 0)
    timer = threading.Timer(deadline, expire)
    timer.daemon = True
    timer.start()
    try:
        for line in proc.stdout:            # relayed as it comes; stderr goes straight through
            sys.stdout.write(line)
            sys.stdout.flush()
        rc = proc.wait()
    finally:
        timer.cancel()
    return 124 if timed_out else rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["bprmf", "lightgcn", "gru4rec"], default="bprmf")
    ap.add_argument("--sessions", type=int, default=10_000_000, help="gru4rec leg: sessions of the synthetic data set")
    ap.add_argument("--no-gru", action="store_true", help="skip the GRU4RecPlus leg (BASELINE configs[4])")
    ap.add_argument("--gru-steps", type=int, default=100)
    ap.add_argument("--start-step", type=int, default=0, help="optimiser step count the run starts from (0 = a fresh model; "
                    "past ~16 600 steps Adam's second bias correction is exactly 1 and its division is skipped)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--interactions", type=int, default=50_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--eval-users", type=int, default=262144)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-epoch", action="store_true", help="skip the whole-epoch leg (N = 1)")
    ap.add_argument("--pre-steps", type=int, default=2048, help="untimed training steps before the warm-up (see main)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the same job (global batch = --batch) on more GPUs; weak = --batch per GPU")
    ap.add_argument("--no-lightgcn", action="store_true", help="skip the secondary LightGCN leg")
    ap.add_argument("--no-layergcn", action="store_true", help="skip the LayerGCN steps at the end of the LightGCN leg")
    ap.add_argument("--lightgcn-steps", type=int, default=10)
    ap.add_argument("--lightgcn-warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=5, help="the K timed steps are run this many times (fresh slices of the epoch each "
                    "time, barrier + synchronize around each); `value` is the median repeat, all of them are listed")
    ap.add_argument("--large-batches", default="16384,65536", help="batch sizes of the separately labelled large-batch legs "
                    "(SURVEY 8d: batch_size is a config knob of the reference); empty = none")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (no HIP call has been made in this process)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n_dev = max(torch.cuda.device_count(), 1)
    local_dev = local_rank % n_dev           # one process per GPU; the modulo only matters for rehearsals on one card
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # the legs run on a compute stream of their own, as fit() does (skrec/recommender/base.py: on_compute_stream -- launches on
    # the device's null stream cost the host more: the 20-step slice 35.0 -> 36.6 M interactions/s, an epoch 1.14 -> 1.07 s
    # on the same box).  SKR_COMPUTE_STREAM=0: the null stream; SKR_BENCH_STEP_PRIORITY=1: that stream at high queue
    # priority (measured: 17.6 M interactions/s, the step launch 45 us instead of 18.5 -- not used)
    if os.environ.get("SKR_COMPUTE_STREAM", "1") != "0":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1 if os.environ.get("SKR_BENCH_STEP_PRIORITY") == "1" else 0))
    import torch.distributed as dist
    if world > 1:
        # "nccl" is RCCL on ROCm.  SKR_DIST_BACKEND=gloo rehearses the N > 1 code path on a single-GPU box.
        backend = os.environ.get("SKR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from skrec import _hip
    from skrec.utils.py.random import DeviceSampler
    L = _hip.lib()
    st = _hip.stream

    if args.workload == "gru4rec":       # the GRU4RecPlus leg as the headline
        leg = gru_leg(args, world, rank, dev, dist, args.steps, args.warmup, not args.no_cpu_baseline)
        out = {"metric": "train interactions/sec + eval users/sec (HR@10/NDCG@10) at 1/2/4/8 MI355X", "higher_is_better": True,
               "vs_baseline": None, "data": "synthetic"}
        out.update(leg)
        out.update(dist_info(world, dist), compute_stream=compute_stream_note())
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return
    full = same_on_every_rank(synth_dataset(args.users, args.items, args.interactions, 20260101, dev), rank, world, dev, dist)
    if args.workload == "lightgcn":      # the LightGCN leg as the headline
        leg = lightgcn_leg(args, world, rank, dev, dist, full, args.steps, args.warmup, not args.no_cpu_baseline)
        out = {"metric": "train interactions/sec + eval users/sec (HR@10/NDCG@10) at 1/2/4/8 MI355X", "higher_is_better": True,
               "vs_baseline": None, "data": "synthetic"}
        out.update(leg)
        out.update(dist_info(world, dist), compute_stream=compute_stream_note())
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return
    n_inter_total = int(full["rowptr"][-1])
    ds, _ = shard(full, rank, world, dev)
    strong = world > 1 and args.scaling == "strong"
    if strong:
        return bprmf_strong(args, world, rank, dev, dist, full, ds)
    nU, nI = len(ds["rowptr"]) - 1, args.items
    b, K, W = args.batch, args.steps, args.warmup

    # ---- model state: the reference's BPRMF tables + dense Adam ----------------------------------
    # one flat buffer [U | V | b] (tables are views) => ONE adam launch per step, as in skrec.recommender.BPRMF
    n_par = nU * D + nI * D + nI
    flat = torch.zeros(n_par, device=dev)
    U, V, bias = flat[:nU * D].view(nU, D), flat[nU * D:(nU + nI) * D].view(nI, D), flat[(nU + nI) * D:]
    U.copy_(torch.randn(nU, D, generator=torch.Generator().manual_seed(2021 + rank)) * 0.01)
    V.copy_(torch.randn(nI, D, generator=torch.Generator().manual_seed(7)) * 0.01)   # identical on every rank
    grad, m1, m2 = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    gU, gV, gb = grad[:nU * D].view(nU, D), grad[nU * D:(nU + nI) * D].view(nI, D), grad[(nU + nI) * D:]
    g_item = grad[nU * D:]                                  # [V | b] gradients: the all-reduced part
    touch = torch.zeros((n_par + 63) // 64, dtype=torch.uint8, device=dev)
    # N > 1, the path's one exchange step: the item table is replicated, its gradient has to be summed over
    # the ranks.  A step touches at most 2*b of the I item rows, so by default the ranks exchange packed rows
    # (skr_pack_grad_rows -> all-gather -> skr_unpack_grad_rows, ~0.54 MB per rank and step) instead of
    # all-reducing the dense [I, 65] block (26 MB); SKR_EXCHANGE=dense keeps the all-reduce.
    exchange = os.environ.get("SKR_EXCHANGE", "sparse") if world > 1 else "none"
    assert exchange in ("none", "sparse", "dense")
    if exchange == "dense":
        touch[nU:] = 2                                      # all-reduced item gradients are read every step
    if exchange == "sparse":
        from skrec.parallel import unique_padded_rows
        pack_buf = torch.empty((2 * b, D + 2), device=dev)
        gather_buf = torch.empty((world, 2 * b, D + 2), device=dev)
        gather_views = [gather_buf[r] for r in range(world)]
    loss = torch.zeros(2 * 32, device=dev)      # skr_bpr_step_spread: SKR_LOSS_SLOTS pairs of loss words

    # ---- the slice of the epoch these W+K steps consume: a user prefix of the local shard ---------
    def prefix(n_need, start_user):
        lo = int(ds["rowptr"][start_user])
        end_user = int(torch.searchsorted(ds["rowptr"], torch.tensor(lo + n_need, device=dev))) + 1
        end_user = min(end_user, nU)
        hi = int(ds["rowptr"][end_user])
        rp = (ds["rowptr"][start_user:end_user + 1] - lo).contiguous()
        return dict(rowptr=rp, users=ds["users"][lo:hi], items=ds["items"][lo:hi], n_users=end_user - start_user,
                    nnz=hi - lo, end_user=end_user)
    # --pre-steps real training steps before the W warm-up steps (untimed, own user range): every kernel and torch
    # helper of the loop has run at full queue depth, and the moments of the rows they touch are no longer all-zero
    n_pre = max(0, min(args.pre_steps, (int(ds["rowptr"][-1]) // b - K - W) // 2))
    if world > 1:   # every step holds collectives: all ranks must run the same number (shards differ in size)
        t_pre = torch.tensor([n_pre], device=dev, dtype=torch.int64)
        dist.all_reduce(t_pre, op=dist.ReduceOp.MIN)
        n_pre = int(t_pre)
    pre = prefix(n_pre * b, 0) if n_pre > 0 else None
    warm = prefix(W * b, pre["end_user"] if pre else 0) if W > 0 else None
    R = max(1, args.repeats)
    timed_slices, at = [], (warm["end_user"] if warm else (pre["end_user"] if pre else 0))
    for _ in range(R):          # every repeat trains on interactions of its own (its sampler call is inside its timed region)
        timed_slices.append(prefix(K * b, at))
        at = timed_slices[-1]["end_user"]
        assert timed_slices[-1]["nnz"] >= K * b, "dataset too small for --steps x --repeats"
    timed = timed_slices[0]
    sampler = DeviceSampler(2020)
    def sample_slice(sl, smp=None):
        neg = torch.empty(sl["nnz"], dtype=torch.int32, device=dev)
        (smp or sampler).sample_epoch_exact(nI, sl["n_users"], sl["rowptr"], sl["items"], sl["nnz"], 1, neg)
        if os.environ.get("SKR_BENCH_SYNC_AFTER_SAMPLER") == "1":     # diagnosis of shared-GPU rehearsals only
            torch.cuda.synchronize()
        return neg

    host_probe = os.environ.get("SKR_BENCH_HOST_PROBE") == "1"      # diagnosis: host-side time of each part of a timed slice

    def run_slice(sl, n_steps, phase=None, neg=None):
        tp0 = time.perf_counter()
        if neg is None:
            neg = sample_slice(sl)
        tp1 = time.perf_counter()
        # device shuffle + batch assembly: ONE launch of the library's own kernel (a keyed bijection of the
        # slice's interactions, evaluated per output row) -- the first n_steps * b rows of the shuffled epoch slice
        run_slice.shuffles += 1
        uu, ii, jj = _hip.shuffle_gather([sl["users"], sl["items"], neg], None, seed=(11 + rank) * 1000003 + run_slice.shuffles,
                                         n_out=n_steps * b)
        # host side of a step = three ctypes calls on cached integer addresses (no tensor slicing, no
        # data_ptr() calls): keeps the launch rate above the kernel rate also at N = 8
        pu, pi, pj = uu.data_ptr(), ii.data_ptr(), jj.data_ptr()
        stream = st()
        if exchange == "sparse":   # per step: the distinct item ids its 2*b gradient rows belong to (-1 = duplicate)
            step_ids = unique_padded_rows(torch.cat([ii.view(n_steps, b), jj.view(n_steps, b)], dim=1))
            pids = step_ids.data_ptr()
        ev = event_pool[:n_steps] if (phase == "timed" and kblk <= 1) else None    # created outside the timed region
        if kblk > 1:
            # Temporally blocked dense Adam (csrc/train.hip K2b; at N = 1 what skrec.recommender.BPRMF.train_epoch does).
            # Per block of kblk steps: rows no batch of the block touches get their kblk zero-gradient updates in ONE
            # pass (adam_cold_rows_kernel), touched rows are advanced when a batch is about to read them or has written their gradient (adam_hot_kernel).
            # Every parameter receives every update in the same arithmetic -- bit-identical to a dense launch per step.
            def block_ids(lo, hi, kk):
                # 64-float blocks of the flat [U | V | bias] buffer the batches lo..hi touch, step-major (5b per step): a hot
                # step names the rows of its own batch and of the next one
                ub, bi, bj = uu[lo:hi].view(kk, b), ii[lo:hi].view(kk, b), jj[lo:hi].view(kk, b)
                return torch.cat([ub, bi + nU, bj + nU, (bi >> 6) + (nU + nI), (bj >> 6) + (nU + nI)], dim=1).reshape(-1)
            if fused:
                # ONE launch per step (csrc/train.hip K2c), through the class BPRMF.train_epoch uses: the next block's words and
                # tags on a stream of their own, the cold pass and the write-back of the rows the next block does not touch
                # on the side stream
                tp2 = time.perf_counter()
                f_opt.t = run_slice.t
                f_opt.cold_timing = [] if phase is not None else None
                if phase == "epoch3":       # blocks 256.. of the third whole epoch: the state an epoch runs in
                    f_opt.cold_event_pool, f_opt.cold_timing_skip = epoch_cold_pool, 256
                    fb.block_timing, fb.block_event_pool, fb.block_timing_skip = epoch_block_log, epoch_block_pool, 256
                else:
                    f_opt.cold_event_pool, f_opt.cold_timing_skip = cold_pool, 0
                nfull, rem = divmod(n_steps, kblk)
                fb.run_blocks(pu, pi, pj, nfull, kblk, b, P["loss"], 0)
                if rem:
                    o = 4 * nfull * kblk * b
                    fb.run_blocks(pu + o, pi + o, pj + o, 1, rem, b, P["loss"], 0)
                f_opt.end_blocks()
                if host_probe and phase == "timed":
                    tp3 = time.perf_counter()
                    print("[bench] host us: sampler call %.0f, shuffle + addresses %.0f, blocks %.0f" % (
                        (tp1 - tp0) * 1e6, (tp2 - tp1) * 1e6, (tp3 - tp2) * 1e6), file=sys.stderr)
                run_slice.t = f_opt.t
                if phase is not None:
                    cold_log.extend(((e0_, e1_), kk_, phase) for e0_, e1_, kk_ in f_opt.cold_timing)
                if phase == "epoch3":
                    fb.block_timing = None
                return
            if world == 1 and not fused:   # every full block of the slice in one vectorised op
                nfull = n_steps // kblk
                blk_all = block_ids(0, nfull * kblk * b, nfull * kblk).view(nfull, kblk * 5 * b)
            for s0 in range(0, n_steps, kblk):
                kk = min(kblk, n_steps - s0)
                lo, hi = s0 * b, (s0 + kk) * b
                if world > 1:
                    # the item table is replicated: its hot rows are those ANY rank's batches of the block touch -- the
                    # ranks exchange the block's item ids once (kk * 2b int32 each), then every rank tags the same item
                    # rows.  Step-major like the N = 1 list: per step b own users + 2b item ids of every rank (+ their
                    # bias words), so a hot step can name just the rows of batch s and s + 1 of ALL ranks.
                    mine = torch.cat([ii[lo:hi].view(kk, b), jj[lo:hi].view(kk, b)], dim=1).contiguous()      # [kk, 2b]
                    every = torch.empty((world, kk, 2 * b), dtype=torch.int32, device=dev)
                    if gather_into:
                        dist.all_gather_into_tensor(every, mine)
                    else:
                        dist.all_gather([every[r] for r in range(world)], mine)
                    # per step the DISTINCT item ids of all ranks (sorted, -1 = empty slot): at N = 8 a step's 16 k gathered
                    # ids name ~2x fewer rows, and every entry of the hot list costs a wavefront
                    every = unique_padded_rows(every.permute(1, 0, 2).reshape(kk, world * 2 * b))
                    if bias_blocks is not None:   # fewer bias blocks than item ids per step: name them all, once each
                        blk = torch.cat([uu[lo:hi].view(kk, b), torch.where(every < 0, every, every + nU),
                                         bias_blocks.expand(kk, -1)], dim=1).reshape(-1)
                        per = b + 2 * world * b + bias_blocks.shape[1]
                    else:
                        blk = torch.cat([uu[lo:hi].view(kk, b), torch.where(every < 0, every, every + nU),
                                         torch.where(every < 0, every, (every >> 6) + (nU + nI))], dim=1).view(-1)
                        per = b + 4 * world * b
                else:
                    blk = blk_all[s0 // kblk] if s0 // kblk < blk_all.shape[0] else block_ids(lo, hi, kk)
                    per = 5 * b
                run_slice.serial += 1
                cur = torch.cuda.current_stream()
                if run_slice.serial > 1:
                    cur.wait_event(ev_cold)      # the previous cold pass still reads the tags / writes cold rows
                t0 = run_slice.t
                rc = L.skr_adam_block_mark(blk.data_ptr(), blk.numel(), 0, 64, blk_tag.data_ptr(), run_slice.serial,
                                           blk_claim.data_ptr(), t0, stream)
                # the cold pass touches no row this block's batches read or write: side stream, under the small launches
                ev_marked.record(cur)
                side.wait_event(ev_marked)
                # every cold pass of the pre-steps, the warm-up and the timed steps is bracketed by HIP events on the stream
                # it is launched on (pairs created before the timed region): the roofline averages over all of them, like
                # rocprofv3 --stats of this command does
                pair = cold_pool.pop() if (phase is not None and cold_pool) else None
                if phase is None and epoch_probe is not None and 200 <= s0 // kblk < 400:    # diagnosis: blocks of a whole epoch
                    pair = mk_pair()
                    blk_pair = mk_pair()
                    blk_pair[0].record(cur)
                    epoch_probe.append((pair, blk_pair))
                else:
                    blk_pair = None
                if pair is not None:
                    pair[0].record(side)
                rc |= L.skr_adam_block_cold(P["flat"], P["m1"], P["m2"], n_par, 1e-3, 0.9, 0.999, 1e-8, run_slice.t, kk,
                                            blk_tag.data_ptr(), run_slice.serial, side.cuda_stream)
                if pair is not None:
                    pair[1].record(side)
                    if phase is not None:
                        cold_log.append((pair, kk, phase))
                ev_cold.record(side)
                pblk, nblk = blk.data_ptr(), blk.numel()
                for s in range(s0, s0 + kk):
                    o = s * b * 4
                    rc |= L.skr_bpr_step_spread(P["U"], P["V"], P["bias"], P["U"], P["V"], pu + o, pi + o, pj + o, b, 1.0, 1e-3, 1.0,
                                         P["gU"], P["gV"], P["gb"], P["gU"], P["gV"], P["loss"], None, None, stream)
                    run_slice.t += 1
                    if world > 1:   # the step's one exchange: packed item-gradient rows, summed in rank order on every rank
                        rc |= L.skr_pack_grad_rows(pids + s * 2 * b * 4, 2 * b, P["gV"], P["gb"], D, pack_buf.data_ptr(), stream)
                        if gather_into:
                            dist.all_gather_into_tensor(gather_buf, pack_buf)
                        else:
                            dist.all_gather(gather_views, pack_buf)
                        rc |= L.skr_unpack_grad_rows_sorted(gather_buf.data_ptr(), 2 * b, world, P["gV"], P["gb"], D, None, None, stream)
                    if s < s0 + kk - 1:      # the block's last step names every hot row: all end at t0 + kk
                        rc |= L.skr_adam_block_hot(P["flat"], P["grad"], P["m1"], P["m2"], n_par, 1e-3, 0.9, 0.999, 1e-8, t0,
                                                   run_slice.t, pblk + 4 * per * (s - s0), 2 * per, 0, 64, blk_claim.data_ptr(), stream)
                    else:
                        rc |= L.skr_adam_block_hot(P["flat"], P["grad"], P["m1"], P["m2"], n_par, 1e-3, 0.9, 0.999, 1e-8, t0,
                                                   run_slice.t, pblk, nblk, 0, 64, blk_claim.data_ptr(), stream)
                if rc:
                    _hip.check(rc)
                if blk_pair is not None:
                    blk_pair[1].record(cur)
                keep_alive.append(blk)
            torch.cuda.current_stream().wait_event(ev_cold)
            return
        for s in range(n_steps):
            o = s * b * 4
            rc = L.skr_bpr_step_spread(P["U"], P["V"], P["bias"], P["U"], P["V"], pu + o, pi + o, pj + o, b, 1.0, 1e-3, 1.0,
                                P["gU"], P["gV"], P["gb"], P["gU"], P["gV"], P["loss"], P["touch"], P["grad"], stream)
            run_slice.t += 1
            if world > 1:
                # the exchange runs on RCCL's stream while Adam sweeps the (local) user part of the flat buffer
                if exchange == "sparse":
                    rc |= L.skr_pack_grad_rows(pids + s * 2 * b * 4, 2 * b, P["gV"], P["gb"], D, pack_buf.data_ptr(), stream)
                    if gather_into:
                        work = dist.all_gather_into_tensor(gather_buf, pack_buf, async_op=True)
                    else:
                        work = dist.all_gather(gather_views, pack_buf, async_op=True)
                else:
                    work = dist.all_reduce(g_item, async_op=True)
                rc |= L.skr_adam_step(P["flat"], P["grad"], P["m1"], P["m2"], n_user_par, 1e-3, 0.9, 0.999, 1e-8,
                                      run_slice.t, 1, P["touch"], stream)
                work.wait()
                if exchange == "sparse":
                    rc |= L.skr_unpack_grad_rows_sorted(gather_buf.data_ptr(), 2 * b, world, P["gV"], P["gb"], D, P["touch"],
                                                 P["grad"], stream)
                if ev is not None:
                    ev[s][0].record()
                rc |= L.skr_adam_step(P["flat"] + 4 * n_user_par, P["grad"] + 4 * n_user_par, P["m1"] + 4 * n_user_par,
                                      P["m2"] + 4 * n_user_par, n_par - n_user_par, 1e-3, 0.9, 0.999, 1e-8, run_slice.t, 1,
                                      P["touch"] + nU, stream)
            else:
                if ev is not None:
                    ev[s][0].record()
                rc |= L.skr_adam_step(P["flat"], P["grad"], P["m1"], P["m2"], n_par, 1e-3, 0.9, 0.999, 1e-8, run_slice.t, 1,
                                      P["touch"], stream)
            if ev is not None:
                ev[s][1].record()
            if rc:
                _hip.check(rc)
        if ev is not None:
            step_events.extend(ev)
    n_user_par = nU * D
    # SKR_ADAM_BLOCK = k (default 32 = the most; N > 1 needs the sparse exchange): look k batches ahead and block the dense Adam; 1 = classic
    kblk = max(1, min(64, int(os.environ.get("SKR_ADAM_BLOCK", "32")))) if (world == 1 or exchange == "sparse") else 1
    # N = 1: the BPR batch and the hot rows' Adam in one launch per step (SKR_BPR_FUSED=0: two dependent launches)
    fused = world == 1 and kblk > 1 and os.environ.get("SKR_BPR_FUSED", "1") != "0" and kblk * 5 * b <= (1 << 20)
    if fused:
        from skrec.recommender.base import DenseAdam
        from skrec.recommender.fused import FusedBlocks
        f_opt = DenseAdam(flat, lr=1e-3)
        f_opt.grad, f_opt.m, f_opt.v = grad, m1, m2        # the loop's own buffers
        fb = FusedBlocks(f_opt, 0, nU, nU + nI, 1e-3)
    blk_tag = torch.zeros((n_par + 63) // 64, dtype=torch.int32, device=dev)
    blk_claim = torch.zeros_like(blk_tag)
    keep_alive = []
    # the cold pass (one sweep over nearly all parameters per k steps, HBM-bound) runs on a side stream underneath the
    # small, latency-bound bpr / hot-step launches of its block, as in BPRMF.train_epoch: 30 vs 24 M interactions/s
    # (SKR_ADAM_OVERLAP=0 turns it off)
    side = torch.cuda.Stream(device=dev) if os.environ.get("SKR_ADAM_OVERLAP", "1") != "0" else torch.cuda.current_stream()
    ev_marked, ev_cold = torch.cuda.Event(), torch.cuda.Event()
    run_slice.serial = 0
    run_slice.shuffles = 0
    n_bias_blocks = (nI + 63) // 64
    bias_blocks = (torch.arange(n_bias_blocks, dtype=torch.int32, device=dev) + (nU + nI)).view(1, -1) \
        if n_bias_blocks <= 2 * b * world else None
    gather_into = world > 1 and dist.get_backend() == "nccl"    # gloo rehearsals use the list form
    if exchange == "sparse":
        # probe the collective once outside the timed region; every rank takes the same branch because a
        # failing collective fails on all of them
        try:
            if gather_into:
                dist.all_gather_into_tensor(gather_buf, pack_buf.zero_())
            else:
                dist.all_gather(gather_views, pack_buf.zero_())
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- fall back to the dense all-reduce rather than lose the run
            if rank == 0:
                print(f"[bench] sparse exchange unavailable ({type(e).__name__}: {e}); using the dense all-reduce",
                      file=sys.stderr)
            exchange = "dense"
            kblk = 1
            touch[nU:] = 2
    P = {k: t.data_ptr() for k, t in dict(U=U, V=V, bias=bias, gU=gU, gV=gV, gb=gb, loss=loss, touch=touch, grad=grad,
                                          flat=flat, m1=m1, m2=m2).items()}
    run_slice.t = int(args.start_step)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # size the sampler's scratch for the timed slices outside the timed regions (consumes stream words), and let it take every
    # slice's row statistics here: they are a property of the CSR, read back ONCE per CSR (in fit() the CSR is the same every
    # epoch, so only the first epoch pays that round trip) -- a timed region must not contain the first sight of its slice
    for sl_ in timed_slices[1:]:
        _scratch = torch.empty(sl_["nnz"], dtype=torch.int32, device=dev)
        sampler.sample_epoch_exact(nI, sl_["n_users"], sl_["rowptr"], sl_["items"], sl_["nnz"], 1, _scratch)
    _scratch = torch.empty(timed["nnz"], dtype=torch.int32, device=dev)
    sampler.sample_epoch_exact(nI, timed["n_users"], timed["rowptr"], timed["items"], timed["nnz"], 1, _scratch)
    # ... and run the slice preparation once at the timed slice's size (first use of a kernel loads its code object)
    _cols = _hip.shuffle_gather([timed["users"], timed["items"], _scratch], None, seed=1, n_out=K * b)
    if exchange == "sparse":
        unique_padded_rows(torch.cat([_cols[1].view(K, b), _cols[2].view(K, b)], dim=1))
    del _scratch, _cols
    # HIP-event pairs around the Adam launches of the timed steps (the roofline's per-launch time).  Created HERE:
    # building 2K timing events costs ~30 us each once a pool of ~1000 is used up (+56 ms inside the region at K = 960)
    mk_pair = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))   # noqa: E731
    event_pool = [mk_pair() for _ in range(K)] if kblk <= 1 else []
    cold_pool = [mk_pair() for _ in range((n_pre + W) // max(kblk, 1) + R * (K // max(kblk, 1) + 1) + 6)] if kblk > 1 else []
    # the third whole epoch: 512 of its blocks (from the 257th on) get their cold pass and their step launches bracketed
    epoch_cold_pool = [mk_pair() for _ in range(512)] if (kblk > 1 and world == 1 and not args.no_epoch) else []
    mk_trip = lambda: tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))   # noqa: E731
    epoch_block_pool = [mk_trip() for _ in range(512)] if epoch_cold_pool else []
    alone_block_pool = [mk_trip() for _ in range(48)] if epoch_cold_pool else []
    epoch_block_log, alone_block_log = [], []
    for pair in event_pool + cold_pool + epoch_cold_pool + epoch_block_pool + alone_block_pool:
        for e_ in pair:
            e_.record()                          # first record creates the HIP event
    cold_log, step_events = [], []
    # SKR_BENCH_EPOCH_PROBE=1: HIP events around the cold pass and around the step launches of 200 blocks of each whole epoch
    epoch_probe = [] if os.environ.get("SKR_BENCH_EPOCH_PROBE") == "1" else None
    if pre is not None:
        run_slice(pre, n_pre, "pre")
    if W > 0:
        run_slice(warm, W, "warmup")
    dts = []
    for r_ in range(R):         # EXACTLY K steps per repeat, barrier + synchronize on both sides, max over the ranks
        barrier()
        t0 = time.perf_counter()
        run_slice(timed_slices[r_], K, "timed")
        barrier()
        dt_r = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_r], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_r = float(tmax)
        dts.append(dt_r)
    dt = float(np.median(dts))
    if kblk > 1:
        n_fresh_log = len(cold_log)              # what follows in the log belongs to the whole-epoch leg
        cold_ms = [(a.elapsed_time(z), kk, ph) for (a, z), kk, ph in cold_log]
        adam_ms = float(np.mean([t for t, _, _ in cold_ms]))
        steps_per_launch = float(np.mean([kk for _, kk, _ in cold_ms]))
    else:
        adam_ms = float(np.mean([a.elapsed_time(z) for a, z in step_events]))
    value = K * b * world / dt
    if fused:       # the tags of the last block FusedBlocks ran
        blk_tag, run_slice.serial = fb.tags[fb.last_q], fb.serial
    n_hot_blocks = int((blk_tag == run_slice.serial).sum()) if kblk > 1 else 0     # hot blocks of the last timed k-step block
    # the same pass ALONE on the chip (copies of the buffers, same tags, same step count): what the kernel does when it
    # does not share HBM and CUs with the step kernels -- its own quality, next to the live (overlapped) figure
    cold_alone_ms = None
    if kblk > 1 and world == 1:
        cp, cm, cv = flat.clone(), m1.clone(), m2.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(4):
            e0.record()
            _hip.check(L.skr_adam_block_cold(cp.data_ptr(), cm.data_ptr(), cv.data_ptr(), n_par, 1e-3, 0.9, 0.999, 1e-8,
                                             run_slice.t + rep * kblk, kblk, blk_tag.data_ptr(), run_slice.serial, st()))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        cold_alone_ms = float(np.mean(ts[1:]))
        del cp, cm, cv
    # ---- one WHOLE epoch through the same loop (N = 1): sampling of every user's negatives, the permutation, all
    # nnz/b steps.  The K timed steps above draw their batches from a user prefix (so that their share of the sampling
    # sits inside the timed region); an epoch's batches spread over all users -- more distinct hot rows per block, and
    # moments of every age.  Reported beside `value`, never instead of it.
    epoch_leg = None
    n_hot_epoch = n_hot_blocks
    if world == 1 and not args.no_epoch:
        whole = prefix(int(ds["rowptr"][-1]), 0)
        n_ep = whole["nnz"] // b
        keep_alive.clear()
        barrier()
        t0e = time.perf_counter()
        run_slice(whole, n_ep)
        barrier()
        te = time.perf_counter() - t0e
        # third epoch, pipelined the way BPRMF.fit() runs (skrec/io/data_iterator.py::_EpochAhead): the NEXT epoch's
        # negatives -- a serial chain on one compute unit -- are drawn on a helper thread / side stream while this
        # epoch trains.  Epoch 2's were drawn during epoch 1 below; the timed epoch draws epoch 3's.
        import threading
        ahead_stream = torch.cuda.Stream(device=dev)
        box = {}

        # the helper draws with a generator of its own: in the second epoch the main thread samples in line at the same time, and
        # a sampler handle is not re-entrant (in fit() the two never overlap: the epoch in hand was drawn during the previous one)
        sampler_ahead = DeviceSampler(2021)

        def draw_ahead():
            with torch.cuda.stream(ahead_stream):
                box["neg"] = sample_slice(whole, sampler_ahead)
        th = threading.Thread(target=draw_ahead, daemon=True)
        t1e = time.perf_counter()
        th.start()
        run_slice(whole, n_ep)
        th.join()
        barrier()
        te2 = time.perf_counter() - t1e
        neg_ahead = box.pop("neg")
        neg_ahead.record_stream(torch.cuda.current_stream())
        th = threading.Thread(target=draw_ahead if os.environ.get("SKR_BENCH_NO_AHEAD3") != "1" else (lambda: None), daemon=True)
        t2e = time.perf_counter()
        th.start()
        run_slice(whole, n_ep, phase="epoch3", neg=neg_ahead)
        te3_host = time.perf_counter() - t2e      # every launch of the epoch queued (the GPU may still be working)
        th.join()
        barrier()
        te3 = time.perf_counter() - t2e
        del neg_ahead
        box.clear()
        keep_alive.clear()
        if fused:
            n_hot_epoch = int((fb.tags[fb.last_q] == fb.serial).sum())        # hot blocks of the epoch's last block
            # the step launches ALONE on the chip: the same loop with the cold pass in front of each block's steps on the same
            # stream instead of beside them (40 blocks of a slice of their own, the last 36 bracketed)
            side_, f_opt._side = f_opt._side, torch.cuda.current_stream()
            fb.block_timing, fb.block_event_pool, fb.block_timing_skip = alone_block_log, alone_block_pool, 4
            n_alone = min(40, n_ep // kblk)
            run_slice(prefix(n_alone * kblk * b, 0), n_alone * kblk)
            torch.cuda.synchronize()
            fb.block_timing, f_opt._side = None, side_
        if epoch_probe:
            torch.cuda.synchronize()
            last = epoch_probe[-200:]            # the third epoch's blocks
            print("[bench] epoch probe (third epoch, blocks 200-399): cold pass %.3f ms, step launches of a block %.3f ms, "
                  "block start to next block start %.3f ms" % (
                      float(np.mean([a_.elapsed_time(z_) for (a_, z_), _ in last])),
                      float(np.mean([a_.elapsed_time(z_) for _, (a_, z_) in last])),
                      float(np.mean([last[i][1][0].elapsed_time(last[i + 1][1][0]) for i in range(len(last) - 1)]))), file=sys.stderr)
        epoch_leg = {"interactions_per_sec": n_ep * b / te3, "seconds": te3, "steps": n_ep, "first_epoch_seconds": te,
                     "first_epoch_interactions_per_sec": n_ep * b / te,
                     "unpipelined_seconds": te, "epoch_drawing_ahead_too_seconds": te2, "host_queued_after_seconds": te3_host,
                     "epochs_seconds": [te, te2, te3],
                     "note": "third of three consecutive full epochs (every row's moments aged by real training), pipelined as "
                             "BPRMF.fit() runs: its negatives were drawn during the previous epoch and it draws the next "
                             "epoch's while training; includes the epoch permutation.  first_epoch = sampling in line.  (The third epoch's time "
                             "varies from run to run on shared boxes, 1.04-1.6 s, with or without the helper thread: "
                             "SKR_BENCH_NO_AHEAD3=1; the first two do not.)"}

    out = {
        "metric": "train interactions/sec + eval users/sec (HR@10/NDCG@10) at 1/2/4/8 MI355X",
        "value": value, "unit": "train interactions/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": args.scaling if world == 1 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "repeats": {"n": R, "statistic": "value / ms_per_step are the MEDIAN of n timed regions of exactly `steps` steps each",
                    "seconds": dts, "value_min": K * b * world / max(dts), "value_max": K * b * world / min(dts)},
        "config": {"workload": f"BASELINE configs[1]: BPRMF d=64, synthetic {args.users}-user/{args.items}-item/"
                               f"{args.interactions}-interaction (MovieLens-shaped), exact-stream sampler + fused BPR "
                               f"step + dense Adam; eval = fused MFMA top-{args.top_k}",
                   "users": args.users, "items": args.items, "train_interactions": n_inter_total,
                   "batch_per_gpu": b, "global_batch": b * world,
                   "step_launches": ("1: skr_bpr_fused_step (BPR batch + the touched rows' Adam, evaluated lazily; bit-identical)" if fused
                                     else "2: skr_bpr_step_spread + skr_adam_block_hot" if kblk > 1 else "2: skr_bpr_step_spread + skr_adam_step"),
                   "sharding": f"users u%{world}, item table replicated"
                   + ({"sparse": f" + RCCL all-gather of the touched item-gradient rows per step ({2 * b * 66 * 4 / 1e6:.2f} MB per rank)",
                       "dense": " + RCCL all-reduce of the dense item gradient per step (26 MB)", "none": ""}[exchange])},
    }
    if epoch_leg is not None:
        out["full_epoch"] = epoch_leg
    pmc_traffic, pmc_source = pmc_lookup()
    if world > 1:
        pmc_traffic = None       # the recorded counters belong to the N = 1 command: a shard's pass moves other bytes
    # ---- roofline of the dominant kernel (adam_kernel over the flat parameter buffer) ---------------
    # 7 fp32 per parameter per step (p,g,m,v in; p,m,v out).  For N > 1 the timed launch is the
    # replicated [V | b] part (the user part overlaps the all-reduce and is not bracketed by the events).
    if kblk > 1:
        # dominant train kernel in blocked mode: adam_cold_rows_kernel, ONE pass per kblk steps over every 64-float block no
        # batch of the k-step block touches.  Algorithmic bytes per LAUNCH: p, m, v read (12 B) and m, v
        # written (8 B) per cold parameter = 20 B; blocks not at rest also write p (4 B more -- not counted: their share
        # depends on the state of the moments).  The launch applies kblk optimiser steps.  SURVEY 8(d)'s per-step figure
        # (28 B per parameter and step) is what this pass replaces kblk times over: `dense_equivalent_GBps`, not `achieved`.
        fresh = cold_roofline(n_par, n_hot_blocks, kblk, cold_ms, cold_alone_ms, side != torch.cuda.current_stream(),
                              pmc_traffic("adam_cold_rows_kernel") if args.users == 1_000_000 else None, pmc_source)
        aged_ms = [(a.elapsed_time(z), kk, ph) for (a, z), kk, ph in cold_log[n_fresh_log:] if ph == "epoch3"]
        if aged_ms:
            # THE roofline entry: the cold passes of the third whole epoch (blocks 257..768) -- every row's moments aged by two
            # epochs of real training, nothing skipped because a moment is still zero: the state an epoch runs in
            out["roofline"] = cold_roofline(n_par, n_hot_epoch, kblk, aged_ms, None, True,
                                            pmc_traffic("adam_cold_rows_kernel@epoch3") if args.users == 1_000_000 else None, pmc_source)
            out["roofline"]["state"] = ("third consecutive whole epoch (full_epoch leg), cold passes of blocks 257-768, every one "
                                        "bracketed by HIP events on the side stream it runs on, underneath the block's step launches")
            fresh["state"] = ("fresh model: the passes of this run's pre-steps, warm-up and timed steps -- most rows' moments are still "
                              "zero and are not written back, which flatters the pass; kept for comparison with rounds 1-2")
            out["roofline_fresh_model"] = fresh
        else:
            fresh["state"] = "fresh model (no whole-epoch leg in this run): flattering, see DESIGN.md 5"
            out["roofline"] = fresh
        if epoch_block_log:
            out["roofline_step"] = step_roofline(b, kblk, epoch_block_log, alone_block_log,
                                                 pmc_traffic("bpr_fused_step_kernel@epoch3") if args.users == 1_000_000 else None, pmc_source)
    else:
        adam_bytes = float(n_par if world == 1 else n_par - nU * D) * 28.0
        ach = adam_bytes / (adam_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "adam_kernel<true> (dense Adam over the flat [U|V|b] buffer, one launch per step)", "bound": "hbm",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": pmc_traffic("adam_kernel") if world == 1 and args.users == 1_000_000 else None,
                           "traffic_source": pmc_source, "avg_launch_ms": adam_ms, "launches_averaged": len(step_events),
                           "algorithmic_bytes_per_launch": adam_bytes}

    if world == 1 and args.large_batches:
        out["large_batch"] = {str(bl): bprmf_large_batch_leg(bl, dev, nU, nI, prefix, sample_slice)
                              for bl in (int(x) for x in args.large_batches.split(",") if x)
                              if (10 * max(1, min(32, (1 << 20) // (5 * bl))) + 12) * bl <= int(ds["rowptr"][-1])}
    if not args.no_eval:
        eval_leg(args, world, rank, dev, dist, U, V, bias, ds, nU, nI, out)
    if world == 1 and not args.no_cpu_baseline:
        cpu_leg(args, U, V, bias, ds, nU, nI, b, out)
    keep_alive.clear()
    finish(args, world, rank, dev, dist, full, V, bias, out)


if __name__ == "__main__":
    main()
