"""bench.py -- the hot path of BASELINE.json configs[1] on N MI355X:

    BPRMF d=64, synthetic MovieLens-shaped 1M users / 100K items / ~50M interactions,
    exact-stream negative sampling + fused BPR step + dense Adam (train), fused MFMA top-K (eval).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one mini-batch of 1024 interactions per rank through the whole training path: its share
of the epoch's negative sampling (the sampler call that produces exactly the negatives these K steps
consume sits INSIDE the timed region), skr_bpr_step, the exchange of the item gradient (N > 1: packed touched rows,
all-gathered, summed in rank order on every rank), and the step's dense Adam update of every parameter of the flat
[U|V|b] buffer (the reference's dense-Adam semantics) in its temporally blocked, bit-identical form: one cold pass per
32 steps over the rows no batch of the block touches + one hot launch per step (SKR_ADAM_BLOCK=1: one skr_adam_step
per step).  Users are sharded u % N; the item table and bias are replicated.  Per-rank batch fixed at 1024, global
batch 1024*N ("weak" in the contract's terms; the user shard per rank is 1/N of the fixed dataset).  Beside the K
timed steps, `full_epoch` reports whole epochs through the same loop (N = 1).
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
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


def lightgcn_main(args, world, rank, dev, dist, full):
    """--workload lightgcn: BASELINE configs[2] (N = 1) / configs[3] (N > 1).  A step is the reference's
    step: full-graph 3-layer propagation forward AND backward for one global batch of 1024*N, fused BPR
    on the propagated tables, dense Adam over [U_local; I]; users sharded, one all-reduce of the [I, 64]
    block per layer and direction (skrec.parallel.ShardedLightGCN)."""
    from skrec import _hip
    from skrec.parallel import DistContext, ShardedLightGCN
    from skrec.utils.py.random import DeviceSampler
    ctx = DistContext(rank, world)
    nU, nI, b, K, W = args.users, args.items, args.batch, args.steps, args.warmup
    n_inter_total = int(full["rowptr"][-1])
    mine = torch.from_numpy(ctx.owned_users(nU)).to(dev)
    g0 = torch.Generator().manual_seed(2021)
    bound = (6.0 / (nU + D)) ** 0.5
    user0 = ((torch.rand(nU, D, generator=g0) * 2 - 1) * bound)[mine.cpu()]
    item0 = (torch.rand(nI, D, generator=torch.Generator().manual_seed(7)) * 2 - 1) * (6.0 / (nI + D)) ** 0.5
    eng = ShardedLightGCN.from_device_edges(ctx, full["users"], full["items"], nU, nI, user0, item0, 3, 1e-3, 1e-3, b)
    gb = b * world
    # the epoch slice these steps consume: a user prefix, sampled with the exact stream on every rank
    need = (W + K) * gb
    end_user = min(int(torch.searchsorted(full["rowptr"], torch.tensor(need, device=dev))) + 1, nU)
    nnz = int(full["rowptr"][end_user])
    assert nnz >= need, "dataset too small for --steps"
    rp = full["rowptr"][:end_user + 1].contiguous()
    sampler = DeviceSampler(2020)
    neg = torch.empty(nnz, dtype=torch.int32, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    gperm = torch.Generator(device=dev).manual_seed(11)
    spmm_events = []

    def run(n_steps, offset, timed):
        sampler.sample_epoch_exact(nI, end_user, rp, full["items"][:nnz], nnz, 1, neg)
        perm = torch.randperm(nnz, generator=gperm, device=dev)[:n_steps * gb]
        uu, ii, jj = (t.index_select(0, perm).contiguous() for t in (full["users"][:nnz], full["items"][:nnz], neg))
        for s_ in range(n_steps):
            sl = slice(s_ * gb, (s_ + 1) * gb)
            if timed and s_ % 4 == 0:   # bracket one forward propagation (3 SpMM per side) every few steps
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.propagate()
                e1.record()
                spmm_events.append((e0, e1))
            eng.train_step(uu[sl], ii[sl], jj[sl])
    if W:
        run(W, 0, False)
    barrier()
    t0 = time.perf_counter()
    run(K, W, False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    run(8, 0, True)    # untimed extra steps only to bracket the propagation with events
    barrier()
    prop_ms = float(np.mean([a.elapsed_time(z) for a, z in spmm_events]))
    nnz_loc = eng.a_ui.nnz
    # algorithmic bytes of one forward propagation on this rank: per SpMM nnz*8 + rows*8 + X in + Y out
    def spmm_bytes(csr, n_x):
        return csr.nnz * 8 + (csr.shape[0] + 1) * 8 + n_x * 256 + csr.shape[0] * 256
    alg = 3 * (spmm_bytes(eng.a_ui, nI) + spmm_bytes(eng.a_iu, eng.n_local))
    ach = alg / (prop_ms * 1e-3) / 1e9
    out = {
        "metric": "train interactions/sec + eval users/sec (HR@10/NDCG@10) at 1/2/4/8 MI355X",
        "value": K * gb / dt, "unit": "train interactions/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{2 if world == 1 else 3}]: LightGCN 3-layer d=64, synthetic {nU}-user/{nI}-item/"
                               f"{args.interactions}-interaction graph, full-graph propagation fwd+bwd per mini-batch "
                               f"(reference semantics), exact-stream sampler, dense Adam",
                   "users": nU, "items": nI, "train_interactions": n_inter_total, "global_batch": gb,
                   "sharding": f"users u%{world}, item block all-reduced per layer ({2 * 3 + 1} x {nI * 256 / 1e6:.1f} MB per step)"},
        "roofline": {"kernel": "spmm_main_kernel (one forward propagation = 6 launches: 3 layers x {user side, item side})",
                     "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": None, "avg_launch_ms": prop_ms / 6, "algorithmic_bytes_per_launch": alg / 6,
                     "note": "gather-bound: nnz*256 B of row gathers come from L2/Infinity Cache at ~8 TB/s (DESIGN.md 4)",
                     "local_nnz": nnz_loc},
    }
    if world == 1 and not args.no_cpu_baseline:
        # the reference's step on the host: torch.sparse.mm x3 forward + autograd backward + dense Adam
        import torch.nn as nn
        idx = torch.stack([torch.cat([full["users"].long(), full["items"].long() + nU]),
                           torch.cat([full["items"].long() + nU, full["users"].long()])]).cpu()
        deg = torch.bincount(idx[0], minlength=nU + nI).float()
        dinv = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
        A = torch.sparse_coo_tensor(idx, dinv[idx[0]] * dinv[idx[1]], (nU + nI, nU + nI)).coalesce()
        E = nn.Parameter(torch.randn(nU + nI, D) * 0.01)
        opt = torch.optim.Adam([E], lr=1e-3)
        us, it_, ng = (t[:3 * b].cpu().long() for t in (full["users"], full["items"], neg if nnz >= 3 * b else full["items"]))
        times = []
        for s_ in range(3):
            t1 = time.perf_counter()
            x, layers = E, [E]
            for _ in range(3):
                x = torch.sparse.mm(A, x)
                layers.append(x)
            fin = torch.stack(layers, 1).mean(1)
            u_, i_, j_ = us[s_ * b:(s_ + 1) * b], it_[s_ * b:(s_ + 1) * b] + nU, ng[s_ * b:(s_ + 1) * b] + nU
            yui, yuj = (fin[u_] * fin[i_]).sum(-1), (fin[u_] * fin[j_]).sum(-1)
            loss = (-torch.nn.functional.logsigmoid(yui - yuj)).mean() + 1e-3 * 0.5 * (E[u_].pow(2).sum() + E[i_].pow(2).sum()
                                                                                       + E[j_].pow(2).sum()) / b
            opt.zero_grad()
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t1)
        t_step = float(np.mean(times[1:]))
        out["cpu_baseline"] = {"value": b / t_step, "unit": "train interactions/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"2 LightGCN steps of {b} with the reference's torch-CPU op sequence "
                               f"(sparse.mm x3 + autograd + dense Adam) on the full graph: {t_step:.2f} s/step"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["bprmf", "lightgcn"], default="bprmf")
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
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n_dev = max(torch.cuda.device_count(), 1)
    local_dev = local_rank % n_dev           # one process per GPU; the modulo only matters for rehearsals on one card
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
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

    full = synth_dataset(args.users, args.items, args.interactions, 20260101, dev)
    if args.workload == "lightgcn":
        return lightgcn_main(args, world, rank, dev, dist, full)
    n_inter_total = int(full["rowptr"][-1])
    ds, _ = shard(full, rank, world, dev)
    if world > 1:
        del full
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
    timed = prefix(K * b, warm["end_user"] if warm else (pre["end_user"] if pre else 0))
    assert timed["nnz"] >= K * b, "dataset too small for --steps"
    sampler = DeviceSampler(2020)
    gperm = torch.Generator(device=dev).manual_seed(11 + rank)

    def sample_slice(sl):
        neg = torch.empty(sl["nnz"], dtype=torch.int32, device=dev)
        sampler.sample_epoch_exact(nI, sl["n_users"], sl["rowptr"], sl["items"], sl["nnz"], 1, neg)
        return neg

    def run_slice(sl, n_steps, events=None, neg=None):
        if neg is None:
            neg = sample_slice(sl)
        perm = torch.randperm(sl["nnz"], generator=gperm, device=dev)[:n_steps * b]
        uu = sl["users"].index_select(0, perm).contiguous()
        ii = sl["items"].index_select(0, perm).contiguous()
        jj = neg.index_select(0, perm).contiguous()
        # host side of a step = three ctypes calls on cached integer addresses (no tensor slicing, no
        # data_ptr() calls): keeps the launch rate above the kernel rate also at N = 8
        pu, pi, pj = uu.data_ptr(), ii.data_ptr(), jj.data_ptr()
        stream = st()
        if exchange == "sparse":   # per step: the distinct item ids its 2*b gradient rows belong to (-1 = duplicate)
            step_ids = unique_padded_rows(torch.cat([ii.view(n_steps, b), jj.view(n_steps, b)], dim=1))
            pids = step_ids.data_ptr()
        ev = event_pool[:n_steps] if events is not None else None    # created outside the timed region
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
            if world == 1:   # every full block of the slice in one vectorised op (as BPRMF.train_epoch does)
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
                if ev is not None:
                    ev[s0][0].record(side)
                rc |= L.skr_adam_block_cold(P["flat"], P["m1"], P["m2"], n_par, 1e-3, 0.9, 0.999, 1e-8, run_slice.t, kk,
                                            blk_tag.data_ptr(), run_slice.serial, side.cuda_stream)
                if ev is not None:
                    ev[s0][1].record(side)
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
                keep_alive.append(blk)
            torch.cuda.current_stream().wait_event(ev_cold)
            if events is not None:
                events.extend(ev[s0] for s0 in range(0, n_steps, kblk))
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
        if events is not None:
            events.extend(ev)
    n_user_par = nU * D
    # SKR_ADAM_BLOCK = k (default 32 = the most; N > 1 needs the sparse exchange): look k batches ahead and block the dense Adam; 1 = classic
    kblk = max(1, min(32, int(os.environ.get("SKR_ADAM_BLOCK", "32")))) if (world == 1 or exchange == "sparse") else 1
    blk_tag = torch.zeros((n_par + 63) // 64, dtype=torch.int32, device=dev)
    blk_claim = torch.zeros_like(blk_tag)
    keep_alive = []
    # the cold pass (one sweep over nearly all parameters per k steps, HBM-bound) runs on a side stream underneath the
    # small, latency-bound bpr / hot-step launches of its block, as in BPRMF.train_epoch: 30 vs 24 M interactions/s
    # (SKR_ADAM_OVERLAP=0 turns it off)
    side = torch.cuda.Stream(device=dev) if os.environ.get("SKR_ADAM_OVERLAP", "1") != "0" else torch.cuda.current_stream()
    ev_marked, ev_cold = torch.cuda.Event(), torch.cuda.Event()
    run_slice.serial = 0
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

    # size the sampler's scratch for the timed slice outside the timed region (consumes stream words)
    _scratch = torch.empty(timed["nnz"], dtype=torch.int32, device=dev)
    sampler.sample_epoch_exact(nI, timed["n_users"], timed["rowptr"], timed["items"], timed["nnz"], 1, _scratch)
    # ... and run the slice preparation (permutation, gathers) once at the timed slice's size: above ~10^6 elements torch
    # switches to other kernels, whose first use costs ~55 ms of code loading that is not part of a step
    _p = torch.randperm(timed["nnz"], generator=torch.Generator(device=dev).manual_seed(1), device=dev)[:K * b]
    _cols = [c.index_select(0, _p).contiguous() for c in (timed["users"], timed["items"], _scratch)]
    if exchange == "sparse":
        unique_padded_rows(torch.cat([_cols[1].view(K, b), _cols[2].view(K, b)], dim=1))
    del _scratch, _p, _cols
    # HIP-event pairs around the Adam launches of the timed steps (the roofline's per-launch time).  Created HERE:
    # building 2K timing events costs ~30 us each once a pool of ~1000 is used up (+56 ms inside the region at K = 960)
    event_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if (kblk <= 1 or s % kblk == 0) else None
                  for s in range(K)]
    for pair in event_pool:
        if pair is not None:
            pair[0].record(); pair[1].record()       # first record creates the HIP event
    if pre is not None:
        run_slice(pre, n_pre)
    if W > 0:
        run_slice(warm, W)
    barrier()
    events = []
    t0 = time.perf_counter()
    run_slice(timed, K, events)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    adam_ms = float(np.mean([a.elapsed_time(z) for a, z in events]))
    value = K * b * world / dt
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

        def draw_ahead():
            with torch.cuda.stream(ahead_stream):
                box["neg"] = sample_slice(whole)
        th = threading.Thread(target=draw_ahead, daemon=True)
        t1e = time.perf_counter()
        th.start()
        run_slice(whole, n_ep)
        th.join()
        barrier()
        te2 = time.perf_counter() - t1e
        neg_ahead = box.pop("neg")
        neg_ahead.record_stream(torch.cuda.current_stream())
        th = threading.Thread(target=draw_ahead, daemon=True)
        t2e = time.perf_counter()
        th.start()
        run_slice(whole, n_ep, neg=neg_ahead)
        th.join()
        barrier()
        te3 = time.perf_counter() - t2e
        del neg_ahead
        box.clear()
        keep_alive.clear()
        epoch_leg = {"interactions_per_sec": n_ep * b / te3, "seconds": te3, "steps": n_ep, "first_epoch_seconds": te,
                     "first_epoch_interactions_per_sec": n_ep * b / te,
                     "unpipelined_seconds": te, "epoch_drawing_ahead_too_seconds": te2,
                     "note": "third of three consecutive full epochs (every row's moments aged by real training), pipelined as "
                             "BPRMF.fit() runs: its negatives were drawn during the previous epoch and it draws the next "
                             "epoch's while training; includes the epoch permutation.  first_epoch = sampling in line."}

    out = {
        "metric": "train interactions/sec + eval users/sec (HR@10/NDCG@10) at 1/2/4/8 MI355X",
        "value": value, "unit": "train interactions/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: BPRMF d=64, synthetic {args.users}-user/{args.items}-item/"
                               f"{args.interactions}-interaction (MovieLens-shaped), exact-stream sampler + fused BPR "
                               f"step + dense Adam; eval = fused MFMA top-{args.top_k}",
                   "users": args.users, "items": args.items, "train_interactions": n_inter_total,
                   "batch_per_gpu": b, "global_batch": b * world, "sharding": f"users u%{world}, item table replicated"
                   + ({"sparse": f" + RCCL all-gather of the touched item-gradient rows per step ({2 * b * 66 * 4 / 1e6:.2f} MB per rank)",
                       "dense": " + RCCL all-reduce of the dense item gradient per step (26 MB)", "none": ""}[exchange])},
    }
    if epoch_leg is not None:
        out["full_epoch"] = epoch_leg
    # HBM traffic per launch comes from the rocprofv3 PMC passes of this same command (separate runs:
    # tools/profile_bench.sh -> tools/summarize_profiles.py -> profiles/<round>_pmc_summary.json)
    pmc = {}
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary.json")))[-1]
        pmc = json.load(open(latest)).get("kernels", {})
    except Exception:
        pmc = {}

    def pmc_traffic(prefix):
        for name, ent in pmc.items():
            if name.startswith(prefix) and "hbm_bytes_per_launch" in ent:
                return ent["hbm_bytes_per_launch"]
        return None
    # ---- roofline of the dominant kernel (adam_kernel over the flat parameter buffer) ---------------
    # 7 fp32 per parameter per step (p,g,m,v in; p,m,v out).  For N > 1 the timed launch is the
    # replicated [V | b] part (the user part overlaps the all-reduce and is not bracketed by the events).
    if kblk > 1:
        # dominant train kernel in blocked mode: adam_cold_rows_kernel, ONE pass per kblk steps over every 64-float block no
        # batch of the k-step block touches.  Algorithmic bytes per LAUNCH: p, m, v read (12 B) and m, v
        # written (8 B) per cold parameter = 20 B; blocks not at rest also write p (4 B more -- not counted: their share
        # depends on the state of the moments).  The launch applies kblk optimiser steps.  SURVEY 8(d)'s per-step figure
        # (28 B per parameter and step) is what this pass replaces kblk times over: `dense_equivalent_GBps`, not `achieved`.
        cold_par = float(n_par - 64 * n_hot_blocks)
        cold_bytes = cold_par * 20.0
        ach = cold_bytes / (adam_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": f"adam_cold_rows_kernel<4> (temporally blocked dense Adam: {kblk} zero-gradient steps per pass over "
                                     f"the blocks no batch of the block touches; rows at rest skip the square root and divisions; "
                                     f"bit-identical to a dense launch per step)",
                           "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": pmc_traffic("adam_cold_rows_kernel") if args.users == 1_000_000 else None,
                           "avg_launch_ms": adam_ms, "algorithmic_bytes_per_launch": cold_bytes,
                           "algorithmic_bytes_per_parameter": 20.0, "cold_parameters": cold_par, "hot_blocks": n_hot_blocks,
                           "optimizer_steps_per_launch": kblk,
                           "overlapped_with_step_kernels": side != torch.cuda.current_stream(),
                           "dense_equivalent_GBps": float(n_par) * 28.0 * kblk / (adam_ms * 1e-3) / 1e9}
        if cold_alone_ms:
            out["roofline"]["alone"] = {"avg_launch_ms": cold_alone_ms, "achieved": cold_bytes / (cold_alone_ms * 1e-3) / 1e9,
                                        "frac": cold_bytes / (cold_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "note": "the same pass with nothing beside it (copies of the buffers after the timed "
                                                "region, same tags): `achieved` / `frac` above are the live figures, with "
                                                "the pass held to 4 workgroups per CU underneath the step kernels"}
    else:
        adam_bytes = float(n_par if world == 1 else n_par - nU * D) * 28.0
        ach = adam_bytes / (adam_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "adam_kernel<true> (dense Adam over the flat [U|V|b] buffer, one launch per step)", "bound": "hbm",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": pmc_traffic("adam_kernel") if world == 1 and args.users == 1_000_000 else None,
                           "avg_launch_ms": adam_ms, "algorithmic_bytes_per_launch": adam_bytes}

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
        mode = os.environ.get("SKR_FUSED_MODE", "bf16x3")
        if mode == "fp32":
            out["roofline_eval"] = {"kernel": "fused_topk_kernel_v3 (FP32 MFMA GEMM + mask + top-K)", "bound": "mfma",
                                    "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                                    "traffic": None, "avg_launch_ms": k_ms, "algorithmic_flop_per_launch": flops}
        else:
            # every fp32 product is formed from six bf16 x bf16 MFMA products with fp32 accumulation, so the matrix
            # pipe issues 6x the algorithmic flops: the peak for THIS arithmetic is the dense bf16 peak / 6
            out["roofline_eval"] = {"kernel": "fused_topk_kernel_v4 (fp32 operands split into 3 bf16 pieces, 6 bf16 MFMAs per "
                                              "fp32 product, fp32 accumulate; GEMM + mask + top-K)", "bound": "mfma",
                                    "achieved": tf, "peak": MFMA_BF16_PEAK_TF / 6.0, "unit": "TFLOP/s", "unit_note": "fp32-equivalent (algorithmic 2*B*I*64 flop)",
                                    "frac": tf / (MFMA_BF16_PEAK_TF / 6.0), "traffic": None, "avg_launch_ms": k_ms,
                                    "algorithmic_flop_per_launch": flops, "mfma_issued_tflops": 6.0 * tf,
                                    "mfma_peak_tflops": MFMA_BF16_PEAK_TF,
                                    "accuracy": "error vs float64 relative to sum|u_i v_i|: max 2.9e-7 (FP32-MFMA kernel: 3.5e-7), "
                                                "tools/fused_accuracy.py"}
            if world == 1:   # the FP32-MFMA kernel on the same inputs, for comparison
                os.environ["SKR_FUSED_MODE"] = "fp32"
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ids_bf = ids.clone()
                f0.record()
                eval_once()
                f1.record()
                torch.cuda.synchronize()
                del os.environ["SKR_FUSED_MODE"]
                ms32 = f0.elapsed_time(f1)
                tf32 = flops / (ms32 * 1e-3) / 1e12
                out["roofline_eval_fp32"] = {"kernel": "fused_topk_kernel_v3 (FP32 MFMA)", "bound": "mfma", "achieved": tf32,
                                             "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf32 / MFMA_F32_PEAK_TF,
                                             "avg_launch_ms": ms32,
                                             "top_k_lists_identical_to_bf16x3": float((ids_bf == ids).all(dim=1).float().mean())}

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
    if world > 1:
        # the replicated item table must be bit-identical on every rank after the timed steps
        chk = torch.stack([V.double().sum(), bias.double().sum(), V.view(torch.int32).long().sum().double()])
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        out["config"]["item_table_replicas_identical"] = bool(torch.equal(lo_, hi_))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
