"""GPU parity at BASELINE.json's full size (1 M users / 100 k items / ~48 M interactions) through
size-independent properties and cross-checks between independent paths."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
U, I = 1_000_000, 100_000


@pytest.fixture(scope="module")
def big():
    import torch
    import bench
    return bench.synth_dataset(U, I, 50_000_000, 20260101, torch.device("cuda", 0))


def test_exact_epoch_full_size(big):
    """(1) the prefix of the epoch equals the reference stream bit for bit (the stream is sequential,
    so the first users' negatives do not depend on the rest); (2) no negative is a train positive of
    its user, all in range; (3) words consumed = slots + rejections >= slots; (4) a second epoch
    continues the stream (differs from the first)."""
    import torch
    from skrec.utils.py.random import DeviceSampler
    rowptr, items = big["rowptr"], big["items"]
    nnz = int(rowptr[-1])
    s = DeviceSampler(2020)
    neg = torch.empty(nnz, dtype=torch.int32, device="cuda")
    s.sample_epoch_exact(I, U, rowptr, items, nnz, 1, neg)
    draws1 = s.draws
    n_pref = 5000
    rp = rowptr[:n_pref + 1].cpu().numpy()
    want = O.Sampler(2020).sample_epoch(I, rp, items[:int(rp[-1])].cpu().numpy(), 1)
    assert np.array_equal(neg[:len(want)].cpu().numpy(), want)
    assert int(neg.min()) >= 0 and int(neg.max()) < I
    key_pos = big["users"].long() * I + items.long()            # sorted by construction
    key_neg = big["users"].long() * I + neg.long()
    idx = torch.searchsorted(key_pos, key_neg).clamp(max=nnz - 1)
    assert not bool((key_pos[idx] == key_neg).any())
    assert nnz <= draws1 <= nnz * 1.01
    neg2 = torch.empty_like(neg)
    s.sample_epoch_exact(I, U, rowptr, items, nnz, 1, neg2)
    assert s.draws > draws1 + nnz - 1 and float((neg2 != neg).float().mean()) > 0.99
    # roughly uniform over the catalogue
    hist = torch.bincount(neg.long(), minlength=I).float()
    assert float(hist.std() / hist.mean()) < 0.1


@pytest.mark.parametrize("K,n_eval", [(10, 1_000_000), (100, 262_144)])
def test_fused_eval_full_size_against_independent_path(big, K, n_eval, fused_mode):
    """fused MFMA top-K over 1 M users (top-10) / 262 144 users (top-100) vs the dense path (skr_score_matrix -> skr_mask_train ->
    skr_eval_scores) on a random sample of users: same ids except where fp32 summation order flips a
    near-tie; every fused list is sorted, masked and duplicate-free; metric sums add up."""
    import torch
    from skrec import _hip
    L = _hip.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    Ut = torch.randn(U, 64, generator=g, device=dev) * 0.1
    Vt = torch.randn(I, 64, generator=g, device=dev) * 0.1
    bias = torch.randn(I, generator=g, device=dev) * 0.05
    nu = n_eval                     # the first n_eval users of the data set
    rowptr, items = big["rowptr"], big["items"]
    users = torch.arange(nu, dtype=torch.int32, device=dev)
    ids = torch.empty((nu, K), dtype=torch.int32, device=dev)
    sc = torch.empty((nu, K), dtype=torch.float32, device=dev)
    chunk = 1 << 18
    ws = int(L.skr_eval_fused_workspace(chunk, K))
    work = torch.empty(ws, dtype=torch.uint8, device=dev)
    for s in range(0, nu, chunk):
        b = min(chunk, nu - s)
        _hip.check(L.skr_eval_fused_topk(_hip.ptr(Ut), _hip.ptr(users[s:s + b]), b, _hip.ptr(Vt), _hip.ptr(bias), I, 64,
                                         _hip.ptr(rowptr), _hip.ptr(items), K, _hip.ptr(ids[s:s + b]), _hip.ptr(sc[s:s + b]),
                                         _hip.ptr(work), ws, _hip.stream()))
    torch.cuda.synchronize()
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    assert int(ids.min()) >= 0 and int(ids.max()) < I
    srt = torch.sort(ids, dim=1).values
    assert not bool((srt[:, 1:] == srt[:, :-1]).any())                       # no duplicates in a list
    key_pos = big["users"].long() * I + items.long()
    key_top = (users.long()[:, None] * I + ids.long()).reshape(-1)
    idx = torch.searchsorted(key_pos, key_top).clamp(max=len(key_pos) - 1)
    assert not bool((key_pos[idx] == key_top).any())                         # train items never ranked
    # independent dense path on a sample
    sample = torch.randperm(nu, generator=g, device=dev)[:2048].int().contiguous()
    dense = _hip.score_matrix(Ut, sample.cpu().numpy(), Vt, bias)
    _hip.check(L.skr_mask_train(_hip.ptr(dense), 2048, I, I, _hip.ptr(sample), _hip.ptr(rowptr), _hip.ptr(items), _hip.stream()))
    ids2 = torch.empty((2048, K), dtype=torch.int32, device=dev)
    _hip.check(L.skr_eval_scores(_hip.ptr(dense), 2048, I, I, None, None, None, 0, K, None, _hip.ptr(ids2), None, _hip.stream()))
    torch.cuda.synchronize()
    a, b2 = ids[sample.long()].cpu().numpy(), ids2.cpu().numpy()
    agree = (a == b2).all(1)
    assert agree.mean() > (0.97 if K <= 10 else 0.5)     # with 100 ranks some neighbouring pair is within fp32 noise more often
    sa, sb = sc[sample.long()].cpu().numpy(), np.take_along_axis(dense.cpu().numpy(), b2.astype(np.int64), 1)
    np.testing.assert_allclose(sa, sb, rtol=2e-5, atol=2e-6)                 # even where ids swap, scores tie
    for r in np.flatnonzero(~agree):
        assert set(a[r]) == set(b2[r]) or abs(sa[r, -1] - sb[r, -1]) < 2e-6
    # metrics: HR@10 from the fused lists == HR computed on the host from the same lists
    test_ptr = torch.arange(nu + 1, dtype=torch.long, device=dev)
    rows = torch.empty((nu, 2 * K), dtype=torch.float32, device=dev)
    sums = torch.zeros(2 * K, dtype=torch.float64, device=dev)
    _hip.check(L.skr_rank_metrics(_hip.ptr(ids), nu, K, _hip.ptr(users), _hip.ptr(test_ptr), _hip.ptr(big["test_item"]),
                                  _hip.metric_array([2, 4]), 2, _hip.ptr(rows), _hip.ptr(sums), _hip.stream()))
    torch.cuda.synchronize()
    hit = (ids == big["test_item"][:nu, None]).any(1).double().sum()
    assert abs(float(sums[K - 1]) - float(hit)) < 0.5                        # Recall@10 == HR@10 on leave-one-out
    np.testing.assert_allclose(sums.cpu().numpy(), rows.double().sum(0).cpu().numpy(), rtol=1e-9)


def test_blocked_adam_full_size_is_bit_identical(big):
    """the training path at full table size (1 M user rows + 100 k item rows + biases = 70.5 M parameters), batches
    of 1024 real interactions of the data set with exact-stream negatives: 96 steps with the temporally blocked Adam
    (k = 32 -- the shipped and benchmarked default -- hot steps naming their own and the next batch) == one dense Adam launch per step, BIT FOR BIT, after
    moments have been aged so that rows at rest, ordinary and lively rows all occur.  Gradient atomics are ordered
    differently from launch to launch, so both runs take their gradients from the same recorded buffers."""
    import torch
    from skrec import _hip
    from skrec.recommender.base import DenseAdam
    from skrec.utils.py.random import DeviceSampler
    L, st = _hip.lib(), _hip.stream
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    n_par = (U + I) * 64 + I
    b, k, n_steps = 1024, 32, 96
    init = torch.randn(n_par, generator=g, device=dev) * 0.05
    # aged moments: every row last touched Exp(mean 977) steps ago (as in tools/microbench_cold.py "steady")
    age = torch.empty(U + I, device=dev).exponential_(1.0 / 977.0, generator=g)
    age = torch.cat([age.repeat_interleave(64), torch.zeros(I, device=dev)])
    m0 = torch.randn(n_par, generator=g, device=dev) * 1e-3 * torch.exp(age * float(np.log(0.9)))
    v0 = torch.rand(n_par, generator=g, device=dev) * 1e-6 * torch.exp(age * float(np.log(0.999)))
    del age
    # batches: a random sample of the data set's interactions + exact-stream negatives of a user prefix
    n_pref = 40_000
    nnz = int(big["rowptr"][n_pref])
    neg = torch.empty(nnz, dtype=torch.int32, device=dev)
    DeviceSampler(2020).sample_epoch_exact(I, n_pref, big["rowptr"][:n_pref + 1].contiguous(), big["items"][:nnz], nnz, 1, neg)
    perm = torch.randperm(nnz, generator=g, device=dev)[:n_steps * b]
    u = big["users"][:nnz][perm].view(n_steps, b).contiguous()
    i = big["items"][:nnz][perm].view(n_steps, b).contiguous()
    j = neg[perm].view(n_steps, b).contiguous()
    t_start = 20_000

    def views(t):
        return t[:U * 64].view(U, 64), t[U * 64:(U + I) * 64].view(I, 64), t[(U + I) * 64:]
    # reference: dense launch per step; the gradient buffer of every step is recorded sparsely (touched blocks only)
    a = DenseAdam(init.clone(), lr=1e-3)
    a.m.copy_(m0)
    a.v.copy_(v0)
    a.t = t_start
    loss = torch.zeros(2, device=dev)
    recorded = []
    for s in range(n_steps):
        (P, Q, Bi), (gP, gQ, gB) = views(a.flat), views(a.grad)
        _hip.check(L.skr_bpr_step(_hip.ptr(P), _hip.ptr(Q), _hip.ptr(Bi), _hip.ptr(P), _hip.ptr(Q), _hip.ptr(u[s]), _hip.ptr(i[s]),
                                  _hip.ptr(j[s]), b, 1.0, 1e-3, 1.0, _hip.ptr(gP), _hip.ptr(gQ), _hip.ptr(gB), _hip.ptr(gP),
                                  _hip.ptr(gQ), _hip.ptr(loss), None, None, st()))
        idx = a.grad.nonzero().flatten()
        recorded.append((idx, a.grad[idx].clone()))
        a.step()
    # blocked: same parameters at every step (bit-identical so far, by induction), so the recorded gradients are its own
    c = DenseAdam(init.clone(), lr=1e-3)
    c.m.copy_(m0)
    c.v.copy_(v0)
    c.t = t_start
    for s0 in range(0, n_steps, k):
        uu, ii, jj = (t[s0:s0 + k] for t in (u, i, j))
        ids = torch.cat([uu, ii + U, jj + U, (ii >> 6) + (U + I), (jj >> 6) + (U + I)], dim=1).reshape(-1)
        c.begin_block(ids, k, per_step=5 * b)
        for s in range(s0, s0 + k):
            idx, val = recorded[s]
            c.grad[idx] = val
            c.hot_step()
    c.end_blocks()
    torch.cuda.synchronize()
    assert c.t == a.t == t_start + n_steps
    for x, y in ((a.flat, c.flat), (a.m, c.m), (a.v, c.v)):
        assert int((x.view(torch.int32) != y.view(torch.int32)).sum()) == 0
    assert float(c.grad.abs().max()) == 0.0
    moved = (a.flat != init).view(-1)[:U * 64].view(U, 64).any(1).float().mean()
    assert 0.01 < float(moved) < 0.9          # rows at rest did not move, touched and lively rows did


def test_fused_step_full_size(big):
    """the shipped training step at full table size: 64 steps of 1024 real interactions (exact-stream negatives) through
    skr_bpr_fused_plan / _step / _end (k = 32, two blocks, the cold pass beside them) against one skr_bpr_step + dense
    skr_adam_step per batch, from aged moments.  Rows shared inside a batch (popular items, bias blocks) sum their float
    atomics in a launch-dependent order in BOTH paths, and every row a batch names is scored against them: named rows are
    compared to rounding; the rows no batch names (their k zero-gradient updates per block in one cold pass) BIT FOR BIT.
    (Bit-identity of the named rows, with gradients that do not depend on the order of atomics: test_fused_step_is_bit_identical.)"""
    import torch
    from skrec import _hip
    from skrec.recommender.base import DenseAdam
    from skrec.recommender.fused import FusedBlocks
    from skrec.utils.py.random import DeviceSampler
    L, st = _hip.lib(), _hip.stream
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    n_par = (U + I) * 64 + I
    b, k, n_steps = 1024, 32, 64
    init = torch.randn(n_par, generator=g, device=dev) * 0.05
    age = torch.empty(U + I, device=dev).exponential_(1.0 / 977.0, generator=g)
    age = torch.cat([age.repeat_interleave(64), torch.zeros(I, device=dev)])
    m0 = torch.randn(n_par, generator=g, device=dev) * 1e-3 * torch.exp(age * float(np.log(0.9)))
    v0 = torch.rand(n_par, generator=g, device=dev) * 1e-6 * torch.exp(age * float(np.log(0.999)))
    del age
    n_pref = 40_000
    nnz = int(big["rowptr"][n_pref])
    neg = torch.empty(nnz, dtype=torch.int32, device=dev)
    DeviceSampler(2020).sample_epoch_exact(I, n_pref, big["rowptr"][:n_pref + 1].contiguous(), big["items"][:nnz], nnz, 1, neg)
    perm = torch.randperm(nnz, generator=g, device=dev)[:n_steps * b]
    u = big["users"][:nnz][perm].contiguous()
    i = big["items"][:nnz][perm].contiguous()
    j = neg[perm].contiguous()
    t_start = 20_000

    def views(t):
        return t[:U * 64].view(U, 64), t[U * 64:(U + I) * 64].view(I, 64), t[(U + I) * 64:]
    a = DenseAdam(init.clone(), lr=1e-3)
    a.m.copy_(m0)
    a.v.copy_(v0)
    a.t = t_start
    la = torch.zeros((n_steps, 2), device=dev)
    (P, Q, Bi), (gP, gQ, gB) = views(a.flat), views(a.grad)
    for s in range(n_steps):
        sl = slice(s * b, (s + 1) * b)
        _hip.check(L.skr_bpr_step(_hip.ptr(P), _hip.ptr(Q), _hip.ptr(Bi), _hip.ptr(P), _hip.ptr(Q), _hip.ptr(u[sl]), _hip.ptr(i[sl]),
                                  _hip.ptr(j[sl]), b, 1.0, 1e-3, 1.0, _hip.ptr(gP), _hip.ptr(gQ), _hip.ptr(gB), _hip.ptr(gP),
                                  _hip.ptr(gQ), _hip.ptr(la[s]), None, None, st()))
        a.step()
    c = DenseAdam(init.clone(), lr=1e-3)
    c.m.copy_(m0)
    c.v.copy_(v0)
    c.t = t_start
    S = _hip.SKR_LOSS_SLOTS
    lc = torch.zeros((n_steps, S, 2), device=dev)
    fb = FusedBlocks(c, 0, U, U + I, 1e-3)
    fb.run_blocks(u.data_ptr(), i.data_ptr(), j.data_ptr(), n_steps // k, k, b, lc.data_ptr(), 8 * S)
    c.end_blocks()
    torch.cuda.synchronize()
    assert c.t == a.t == t_start + n_steps
    np.testing.assert_allclose(lc.sum(1).cpu().numpy(), la.cpu().numpy(), rtol=1e-5)
    # rows no batch names take only zero-gradient updates: the same bits in both runs
    uu = u.view(n_steps, b).long()
    cnt = torch.zeros(U, dtype=torch.int64, device=dev).index_add_(0, uu.reshape(-1), torch.ones(n_steps * b, dtype=torch.int64, device=dev))
    cold = cnt == 0
    for x, y in ((a.flat, c.flat), (a.m, c.m), (a.v, c.v)):
        xu, yu = x[:U * 64].view(U, 64)[cold], y[:U * 64].view(U, 64)[cold]
        assert int((xu.view(torch.int32) != yu.view(torch.int32)).sum()) == 0
    assert int(cold.sum()) > 900_000 and int((cnt > 0).sum()) > 20_000
    # everything else to rounding
    for x, y, atol in ((a.flat, c.flat, 2e-6), (a.m, c.m, 1e-7), (a.v, c.v, 1e-9)):
        assert bool(((x - y).abs() <= atol + 1e-5 * y.abs()).all()), float((x - y).abs().max())
    assert float(fb.work[:, 6 * fb.cap * 64:].abs().max()) == 0.0


def test_lightgcn_full_size_propagation_and_step(big):
    """BASELINE configs[2] at size: LightGCN, 3 layers, 'pre' adjacency of the 1 M-user / 100 k-item / 48 M-interaction
    graph (1.1 M rows, 97 M non-zeros), reference LightGCN.py:89-100,180-199.
    (1) the propagation kernel: skr_spmm_plan_* (and the plan-free skr_csr_spmm) on the full adjacency vs a float64 host
        computation of sampled rows -- user rows, the ten longest item rows (~10^5..10^6 entries, cut into thousands of
        column-blocked tasks), short item rows, an empty row;
    (2) the layer mean of `propagate()` on all 1.1 M rows vs the oracle's fp32 propagation (oracle.lightgcn_propagate);
    (3) one train_step of the user-sharded engine (world = 1): its (bpr mean, l2) against the oracle's BPR maths on the
        oracle's propagated tables, 1e-5 relative; its gradient of the ego table against the oracle's explicit backward
        (oracle.lightgcn_step) on sampled rows."""
    import scipy.sparse as sp
    import torch
    from skrec import _hip
    from skrec.parallel import DistContext, ShardedLightGCN
    from skrec.recommender.LightGCN import build_adjacency_device
    from skrec.utils.py.random import DeviceSampler
    dev = torch.device("cuda", 0)
    N = U + I
    users, items = big["users"], big["items"]
    adj, adj_t = build_adjacency_device(users, items, U, I, "pre", dev)
    assert adj_t is adj and adj.shape == (N, N) and adj.nnz == 2 * users.numel()
    rp, col, val = (t.cpu().numpy() for t in (adj.rowptr, adj.col, adj.val))
    A = sp.csr_matrix((val, col, rp), shape=(N, N))
    g = torch.Generator(device=dev).manual_seed(11)
    X = torch.randn(N, 64, generator=g, device=dev) * 0.1
    Xh = X.cpu().numpy()
    lens = np.diff(rp)
    rng = np.random.default_rng(3)
    rows = np.unique(np.concatenate([rng.integers(0, U, 1500), U + rng.integers(0, I, 1500), np.argsort(lens)[-10:],
                                     np.flatnonzero(lens == 0)[:3], np.flatnonzero((lens >= 500) & (lens <= 520))[:20]]))
    want = np.stack([(val[rp[r]:rp[r + 1]].astype(np.float64)[:, None] * Xh[col[rp[r]:rp[r + 1]]].astype(np.float64)).sum(0) for r in rows])
    mass = np.stack([(np.abs(val[rp[r]:rp[r + 1]]).astype(np.float64)[:, None] * np.abs(Xh[col[rp[r]:rp[r + 1]]])).sum(0) for r in rows])
    info = adj.plan_info()
    assert info["long_rows"] == int((lens >= 512).sum()) > 1000 and info["tasks"] > info["long_rows"]
    Y = torch.empty_like(X)
    for plan in ("1", "0"):
        import os
        os.environ["SKR_SPMM_PLAN"] = plan
        try:
            Y.fill_(7.0)
            adj.spmm(X, Y)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("SKR_SPMM_PLAN")
        got = Y[torch.from_numpy(rows).to(dev)].cpu().numpy()
        assert np.all(np.abs(got - want) <= 2e-6 * mass + 1e-7), (plan, np.abs(got - want).max())
    # (2) + (3): the engine
    ctx = DistContext(0, 1)
    bound_u, bound_i = (6.0 / (U + 64)) ** 0.5, (6.0 / (I + 64)) ** 0.5
    user0 = (torch.rand(U, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1) * bound_u
    item0 = (torch.rand(I, 64, generator=torch.Generator().manual_seed(2)) * 2 - 1) * bound_i
    b, reg = 1024, 1e-3
    eng = ShardedLightGCN.from_device_edges(ctx, users, items, U, I, user0, item0, 3, 1e-3, reg, b)
    E0 = torch.cat([user0, item0]).numpy()
    Ebar = O.lightgcn_propagate(A, E0, 3)                       # the oracle's fp32 propagation (scipy, ~10 s per layer)
    final = eng.propagate().cpu().numpy()
    np.testing.assert_allclose(final, Ebar, rtol=0, atol=5e-6 * np.abs(Ebar).max())
    n_pref = 2000
    nnz = int(big["rowptr"][n_pref])
    neg = torch.empty(nnz, dtype=torch.int32, device=dev)
    DeviceSampler(2020).sample_epoch_exact(I, n_pref, big["rowptr"][:n_pref + 1].contiguous(), items[:nnz], nnz, 1, neg)
    sel = torch.from_numpy(rng.permutation(nnz)[:b]).to(dev)
    bu, bi, bj = users[:nnz][sel].contiguous(), items[:nnz][sel].contiguous(), neg[sel].contiguous()
    g_before = eng._g_ego.clone()
    assert float(g_before.abs().max()) == 0.0
    # take the gradient before Adam consumes it: run the step with the optimiser's step() disabled
    step = eng.optimizer.step
    eng.optimizer.step = lambda: None
    try:
        eng.train_step(bu, bi, bj)
    finally:
        eng.optimizer.step = step
    torch.cuda.synchronize()
    hu, hi, hj = bu.cpu().numpy(), bi.cpu().numpy(), bj.cpu().numpy()
    loss, l2, _, _, _, _, _ = O.bpr_batch(Ebar[:U], Ebar[U:], None, E0[:U], E0[U:], hu, hi, hj, 1.0 / b, reg, 1.0 / b)
    got = eng.loss.cpu().numpy()
    np.testing.assert_allclose(got[0], loss, rtol=1e-5)
    np.testing.assert_allclose(got[1], l2, rtol=1e-5)
    # the oracle's explicit backward (oracle.lightgcn_step's steps, re-using Ebar; A is symmetric, so A^T = A)
    _, _, gPu, gQi, _, gRu, gRi = O.bpr_batch(Ebar[:U], Ebar[U:], None, E0[:U], E0[U:], hu, hi, hj, 1.0 / b, reg, 1.0 / b)
    H = (np.concatenate([gPu, gQi], 0) * np.float32(0.25)).astype(np.float32)
    G = H
    for _ in range(3):
        G = (A @ G).astype(np.float32) + H
    G = G + np.concatenate([gRu, gRi], 0)
    probe = np.unique(np.concatenate([hu[:200], U + hi[:200], U + hj[:200], rng.integers(0, N, 2000), np.argsort(lens)[-5:]]))
    gg = eng._g_ego[torch.from_numpy(probe).to(dev)].cpu().numpy()
    scale = np.abs(G).max()
    np.testing.assert_allclose(gg, G[probe], rtol=2e-5, atol=2e-6 * scale)
    # the step above skipped what is never read / sums of zeros (row and column masks); the fully dense step agrees
    import os
    g_masked = eng._g_ego.clone()
    eng._g_ego.zero_()
    os.environ["SKR_LIGHTGCN_DENSE"] = "1"
    eng.optimizer.step = lambda: None
    try:
        eng.train_step(bu, bi, bj)
    finally:
        eng.optimizer.step = step
        os.environ.pop("SKR_LIGHTGCN_DENSE")
    torch.cuda.synchronize()
    np.testing.assert_allclose(eng.loss.cpu().numpy(), got, rtol=1e-6)
    assert float((eng._g_ego - g_masked).abs().max()) <= 2e-6 * scale
