"""oracle/cpu_baseline.py -- the CPU leg of bench.py (TEST/BENCH INFRASTRUCTURE, never on the product
path).  It times the reference's way of doing the hot path on the host cores of the GPU box:

* sampler   ``oracle/_ref/libskrec_ref.so`` -- the reference's own randint.h compiled where it lay
            (kind "reference"); if that file is absent, our C restatement (kind "port");
* BPR step  the reference's torch-CPU op sequence restated with torch modules: three nn.Embedding
            gathers per score, inner_product, -logsigmoid sum, l2_loss over five gathers, dense
            autograd backward, torch.optim.Adam over every row (BPRMF.py:110-127) -- kind "port";
* eval      torch-CPU matmul + numpy -inf masking + the reference's cpp_evaluate_matrix through
            oracle/_ref (evaluator.py:191-203, test_batch_size 64, 4 threads).
"""
import time

import numpy as np

from . import oracle as O


def time_sampler(num_items, rowptr, pos, budget_s=8.0):
    """negatives/s of _sampling_negative_items at the native level over a user prefix"""
    n_users = len(rowptr) - 1
    take = min(n_users, 20000)
    fn = O.ref_sample_epoch if O.have_ref() else (lambda I, r, p, k: O.Sampler(2020).sample_epoch(I, r, p, k))
    kind = "reference" if O.have_ref() else "port"
    t_used, done = 0.0, 0
    while True:
        rp = rowptr[:take + 1].astype(np.int64) - rowptr[0]
        p = np.ascontiguousarray(pos[:int(rp[-1])])
        t0 = time.perf_counter()
        fn(num_items, rp, p, 1)
        dt = time.perf_counter() - t0
        t_used += dt
        done = int(rp[-1])
        rate = done / dt
        if t_used > budget_s or take >= n_users or dt > 2.0:
            return rate, kind, f"{take} users / {done} negatives, native loop only (no Python per-user overhead)"
        take = min(n_users, take * 4)


def time_bprmf_steps(num_users, num_items, dim, users, pos, neg, batch, steps=12, warmup=2, lr=1e-3, reg=1e-3):
    """seconds per training step of the reference's torch-CPU sequence at full table size"""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    ue, ie, be = nn.Embedding(num_users, dim), nn.Embedding(num_items, dim), nn.Embedding(num_items, 1)
    nn.init.normal_(ue.weight, 0, 0.01)
    nn.init.normal_(ie.weight, 0, 0.01)
    nn.init.zeros_(be.weight)
    opt = torch.optim.Adam(list(ue.parameters()) + list(ie.parameters()) + list(be.parameters()), lr=lr)

    def score(u, i):
        return torch.sum(ue(u) * ie(i), dim=-1) + be(i).squeeze()
    times = []
    for s in range(warmup + steps):
        sl = slice(s * batch, (s + 1) * batch)
        t0 = time.perf_counter()
        u = torch.from_numpy(users[sl]).long()
        i = torch.from_numpy(pos[sl]).long()
        j = torch.from_numpy(neg[sl]).long()
        loss = (-F.logsigmoid(score(u, i) - score(u, j))).sum()
        l2 = 0.5 * sum(torch.sum(torch.pow(w, 2)) for w in (ue(u), ie(i), ie(j), be(i), be(j)))
        loss = loss + reg * l2
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if s >= warmup:
            times.append(dt)
    return float(np.mean(times)), torch.get_num_threads()


def time_eval_batches(user_table, item_table, bias, train_rowptr, train_items, test_items, users, K=10, batches=4,
                      threads=4):
    """users/s of the reference evaluator loop: 64-user batches, CPU matmul, numpy masking, native top-K"""
    import torch
    ut, it = torch.from_numpy(user_table), torch.from_numpy(item_table)
    b = torch.from_numpy(bias) if bias is not None else None
    t_tot, n = 0.0, 0
    for k in range(batches):
        bu = users[k * 64:(k + 1) * 64]
        if len(bu) == 0:
            break
        t0 = time.perf_counter()
        sc = torch.matmul(ut[torch.from_numpy(bu).long()], it.T)
        if b is not None:
            sc += b
        sc = sc.numpy()
        for r, u in enumerate(bu):
            sc[r][train_items[train_rowptr[u]:train_rowptr[u + 1]]] = -np.inf
        tests = [test_items[u:u + 1] for u in bu]
        if O.have_ref():
            O.ref_eval_score_matrix(sc, tests, [2, 4], K, thread_num=threads)
        else:
            O.eval_score_matrix(sc, tests, [2, 4], K)
        t_tot += time.perf_counter() - t0
        n += len(bu)
    return n / t_tot, ("reference" if O.have_ref() else "port")


def time_lightgcn_layer(rowptr, col, val, n_nodes, dim, n_layers=3, budget_s=30.0):
    """The reference's LightGCN step on the host (LightGCN.py:89-100,180-199): K x torch.sparse.mm on the normalised COO
    adjacency, the autograd backward of each, dense Adam.  A whole step at BASELINE size takes ~45 s on 128 cores, so
    ONE layer is timed forward + backward (two sparse products) plus one dense Adam update, and the step is extrapolated
    as n_layers x that layer + Adam.  -> (seconds per step [extrapolated], cores, description)"""
    import torch
    import torch.nn as nn
    rows = np.repeat(np.arange(n_nodes, dtype=np.int64), np.diff(rowptr))
    idx = torch.from_numpy(np.stack([rows, col.astype(np.int64)]))
    A = torch.sparse_coo_tensor(idx, torch.from_numpy(val), (n_nodes, n_nodes), is_coalesced=True)
    E = nn.Parameter(torch.randn(n_nodes, dim) * 0.01)
    opt = torch.optim.Adam([E], lr=1e-3)
    t0 = time.perf_counter()
    y = torch.sparse.mm(A, E)
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    y.sum().backward()
    t_bwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    opt.step()
    t_adam = time.perf_counter() - t0
    t_step = n_layers * (t_fwd + t_bwd) + t_adam
    return t_step, torch.get_num_threads(), (f"one layer of the reference's torch-CPU sequence on the full graph: sparse.mm forward "
                                             f"{t_fwd:.2f} s + autograd backward {t_bwd:.2f} s, dense Adam {t_adam:.2f} s; step "
                                             f"extrapolated as {n_layers} x layer + Adam = {t_step:.1f} s")
