"""GPU parity, T rows: skr_bpr_step, skr_adam_step, skr_csr_spmm, skr_layer_refine_* against the
oracle's explicit-gradient maths (itself pinned to the reference trajectories) and torch-CPU fp32."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("n,with_bias,split", [(1, True, False), (257, True, False), (1024, False, True), (3000, True, True)])
def test_bpr_step_vs_oracle(n, with_bias, split):
    import torch
    from gpu_utils import to_dev, dev
    from skrec import _hip
    rng = np.random.default_rng(n)
    nU, nI = 300, 200
    P = (rng.standard_normal((nU, 64)) * 0.3).astype(np.float32)
    Q = (rng.standard_normal((nI, 64)) * 0.3).astype(np.float32)
    bias = (rng.standard_normal(nI) * 0.2).astype(np.float32) if with_bias else None
    RP = (rng.standard_normal((nU, 64)) * 0.3).astype(np.float32) if split else P
    RQ = (rng.standard_normal((nI, 64)) * 0.3).astype(np.float32) if split else Q
    u = rng.integers(0, nU, n).astype(np.int32)
    i = rng.integers(0, nI, n).astype(np.int32)
    j = rng.integers(0, nI, n).astype(np.int32)
    ls, reg, rs = (1.0 / n, 1e-3, 1.0 / 1024) if split else (1.0, 1e-3, 1.0)
    loss, l2, gP, gQ, gb, gRP, gRQ = O.bpr_batch(P, Q, bias, RP, RQ, u, i, j, ls, reg, rs)
    dP, dQ, dRP, dRQ = to_dev(P), to_dev(Q), to_dev(RP), to_dev(RQ)
    db = to_dev(bias) if with_bias else None
    z = lambda a: torch.zeros_like(to_dev(a))  # noqa: E731
    ggP, ggQ = z(P), z(Q)
    ggRP, ggRQ = (z(RP), z(RQ)) if split else (ggP, ggQ)
    ggb = z(bias) if with_bias else None
    dl = torch.zeros(2, dtype=torch.float32, device=dev())
    du, di, dj = to_dev(u), to_dev(i), to_dev(j)  # keep the tensors alive across the launch
    _hip.check(_hip.lib().skr_bpr_step(_hip.ptr(dP), _hip.ptr(dQ), _hip.ptr(db), _hip.ptr(dRP if split else dP),
                                       _hip.ptr(dRQ if split else dQ), _hip.ptr(du), _hip.ptr(di),
                                       _hip.ptr(dj), n, ls, reg, rs, _hip.ptr(ggP), _hip.ptr(ggQ), _hip.ptr(ggb),
                                       _hip.ptr(ggRP), _hip.ptr(ggRQ), _hip.ptr(dl), None, None, _hip.stream()))
    torch.cuda.synchronize()
    got = dl.cpu().numpy()
    assert abs(got[0] - loss) <= 1e-5 * abs(loss) and abs(got[1] - l2) <= 1e-5 * abs(l2)
    _close(ggP.cpu().numpy(), gP)
    _close(ggQ.cpu().numpy(), gQ)
    if split:
        _close(ggRP.cpu().numpy(), gRP)
        _close(ggRQ.cpu().numpy(), gRQ)
    if with_bias:
        _close(ggb.cpu().numpy(), gb)


@pytest.mark.parametrize("use_touch", [False, True])
def test_adam_matches_torch_cpu(use_touch):
    """dense torch.optim.Adam on the CPU vs skr_adam_step; with touch bytes the kernel must skip only
    gradient blocks that really are zero and give the same numbers"""
    import torch
    from gpu_utils import to_dev
    from skrec import _hip
    rng = np.random.default_rng(0)
    n = 64 * 1000 + 3
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([pt], lr=1e-3)
    dp, dm, dv = to_dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    nb = (n + 63) // 64
    for t in range(1, 6):
        blocks = rng.random(nb) < 0.3                     # gradient lives in ~30 % of the 64-float blocks
        g = (rng.standard_normal(n) * np.repeat(blocks, 64)[:n]).astype(np.float32)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        dg = to_dev(g)
        touch = to_dev(np.where(blocks, 1 + (np.arange(nb) % 2), 0).astype(np.uint8)) if use_touch else None
        _hip.check(_hip.lib().skr_adam_step(_hip.ptr(dp), _hip.ptr(dg), _hip.ptr(dm), _hip.ptr(dv), n, 1e-3, 0.9, 0.999,
                                            1e-8, t, 1, _hip.ptr(touch), _hip.stream()))
        torch.cuda.synchronize()
        assert float(dg.abs().max()) == 0.0  # zero_grad fused
        if use_touch:  # 1 -> cleared, 2 -> sticky, 0 stays
            want = np.where(blocks, np.where(np.arange(nb) % 2 == 1, 2, 0), 0)
            assert np.array_equal(touch.cpu().numpy(), want)
        _close(dp.cpu().numpy(), pt.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_bpr_step_marks_touched_blocks():
    import torch
    from gpu_utils import to_dev, dev
    from skrec import _hip
    rng = np.random.default_rng(9)
    nU, nI, n = 50, 40, 30
    flat = torch.zeros(nU * 64 + nI * 64 + nI, device=dev())
    g = torch.zeros_like(flat)
    touch = torch.zeros((flat.numel() + 63) // 64, dtype=torch.uint8, device=dev())
    P, Q, b = flat[:nU * 64].view(nU, 64), flat[nU * 64:(nU + nI) * 64].view(nI, 64), flat[(nU + nI) * 64:]
    flat.normal_()
    gP, gQ, gb = g[:nU * 64].view(nU, 64), g[nU * 64:(nU + nI) * 64].view(nI, 64), g[(nU + nI) * 64:]
    u, i, j = (rng.integers(0, m, n).astype(np.int32) for m in (nU, nI, nI))
    du, di, dj = to_dev(u), to_dev(i), to_dev(j)
    loss = torch.zeros(2, device=dev())
    _hip.check(_hip.lib().skr_bpr_step(_hip.ptr(P), _hip.ptr(Q), _hip.ptr(b), _hip.ptr(P), _hip.ptr(Q), _hip.ptr(du),
                                       _hip.ptr(di), _hip.ptr(dj), n, 1.0, 1e-3, 1.0, _hip.ptr(gP), _hip.ptr(gQ), _hip.ptr(gb),
                                       _hip.ptr(gP), _hip.ptr(gQ), _hip.ptr(loss), _hip.ptr(touch), _hip.ptr(g), _hip.stream()))
    torch.cuda.synchronize()
    nz = (g.cpu().numpy() != 0)
    nz = np.pad(nz, (0, (-len(nz)) % 64)).reshape(-1, 64).any(1)
    t = touch.cpu().numpy().astype(bool)
    assert np.all(t[nz]) and t.sum() <= len(set(u)) + 2 * len(set(i) | set(j)) + 2


@pytest.mark.parametrize("plan", ["0", "1"])
@pytest.mark.parametrize("n_rows,density,heavy", [(50, 0.2, 0), (3000, 0.004, 2), (20000, 0.0005, 3)])
def test_csr_spmm_vs_scipy(n_rows, density, heavy, plan, monkeypatch):
    """both forms of the product: the plan-free kernel (plan = "0") and skr_spmm_plan_* (plan = "1": the heavy rows
    here exceed 512 entries, so they are cut into column-blocked tasks and re-assembled from partial rows)"""
    import torch
    from gpu_utils import dev
    from skrec.recommender.LightGCN import DeviceCSR
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)
    rng = np.random.default_rng(n_rows)
    A = sp.random(n_rows, n_rows, density=density, format="lil", random_state=rng, dtype=np.float32)
    for h in range(heavy):  # rows far longer than one 512-nnz chunk -> split rows + atomics
        cols = rng.choice(n_rows, min(n_rows, 2000 + 700 * h), replace=False)
        A[h * 7, cols] = rng.standard_normal(len(cols)).astype(np.float32)
    A = sp.csr_matrix(A)
    A[n_rows // 2, :] = 0  # an empty row
    A.eliminate_zeros()
    X = rng.standard_normal((n_rows, 64)).astype(np.float32)
    add = rng.standard_normal((n_rows, 64)).astype(np.float32)
    acc0 = rng.standard_normal((n_rows, 64)).astype(np.float32)
    csr = DeviceCSR(A, dev())
    dX, dadd, dacc = (torch.from_numpy(a).to(dev()) for a in (X, add, acc0.copy()))
    Y = torch.full((n_rows, 64), 7.0, device=dev())
    csr.spmm(dX, Y)
    torch.cuda.synchronize()
    want = (A.astype(np.float64) @ X.astype(np.float64))
    # fp32 accumulation noise grows with the row's absolute mass (rows of 2000+ terms here)
    mass = (abs(A).astype(np.float64) @ np.abs(X).astype(np.float64))
    tol = 2e-6 * mass + 1e-6

    def close(got, ref):
        assert np.all(np.abs(got - ref) <= tol + 2e-6 * np.abs(ref)), np.abs(got - ref).max()
    close(Y.cpu().numpy(), want)
    csr.spmm(dX, Y, addend=dadd, accum=dacc, accum_scale=0.25)
    torch.cuda.synchronize()
    close(Y.cpu().numpy(), want + add)
    close(dacc.cpu().numpy(), acc0 + 0.25 * (want + add))
    if plan == "1":
        info = csr.plan_info()
        assert info["long_rows"] == heavy and (info["tasks"] > 0) == (heavy > 0), info


@pytest.mark.parametrize("windows", [None, "3"])
@pytest.mark.parametrize("n_rows,n_cols,long_from", [(300, 40_000, 2), (1000, 70_000, 8), (64, 200_000, 100), (5000, 33_000, 0), (7, 5, 2)])
def test_spmm_plan_rectangular_blocks_and_determinism(n_rows, n_cols, long_from, windows, monkeypatch):
    """skr_spmm_plan_* on rectangular matrices whose long rows span several 16 384-column blocks (tasks of every length
    1..256, rows of exactly the threshold, empty rows, a row touching only the last block), with the long-row threshold
    lowered so that small cases exercise the task path; repeated runs are BIT-identical (partial rows are added in a
    fixed order -- no float atomics), epilogues included; argument checks.  windows = "3": the short rows gather from X in
    three column windows, one launch each (what the plan does by itself when X exceeds the Infinity Cache)"""
    import ctypes as C
    if windows:
        monkeypatch.setenv("SKR_SPMM_WINDOWS", windows)
    else:
        monkeypatch.delenv("SKR_SPMM_WINDOWS", raising=False)
    import torch
    from gpu_utils import dev, to_dev
    from skrec import _hip
    L = _hip.lib()
    rng = np.random.default_rng(n_rows * 7 + long_from)
    lens = np.minimum(rng.integers(0, 40, n_rows) ** 2 // 3, n_cols)
    lens[rng.integers(0, n_rows, max(1, n_rows // 50))] = min(n_cols, 3000)          # long rows across many blocks
    lens[0] = 0
    if long_from >= 2 and n_rows > 3:
        lens[1], lens[2] = min(long_from, n_cols), min(long_from, n_cols) - 1          # exactly at / just under the threshold
    rowptr = np.zeros(n_rows + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    col = np.concatenate([np.sort(rng.choice(n_cols, l, replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    if n_rows > 5 and lens[5] > 0:
        b, e = rowptr[5], rowptr[6]
        col[b:e] = np.arange(n_cols - (e - b), n_cols)                                 # only the last columns
    val = rng.standard_normal(len(col)).astype(np.float32)
    nnz = len(col)
    A = sp.csr_matrix((val, col, rowptr), shape=(n_rows, n_cols))
    X = rng.standard_normal((n_cols, 64)).astype(np.float32)
    add = rng.standard_normal((n_rows, 64)).astype(np.float32)
    acc0 = rng.standard_normal((n_rows, 64)).astype(np.float32)
    d_rp, d_col, d_val = to_dev(rowptr), to_dev(col if nnz else np.zeros(1, np.int32)), to_dev(val if nnz else np.zeros(1, np.float32))
    dX, dadd = to_dev(X), to_dev(add)
    h = C.c_void_p()
    _hip.check(L.skr_spmm_plan_create(n_rows, n_cols, _hip.ptr(d_rp), _hip.ptr(d_col), _hip.ptr(d_val), nnz, long_from, C.byref(h), _hip.stream()))
    info = (C.c_int64 * 4)()
    _hip.check(L.skr_spmm_plan_info(h, info))
    thr = long_from or 512
    # columns per block (spmm.hip plan_default_cblk): a multiple of 8 blocks of ~28 k columns, equal widths (multiples of 64)
    if n_cols < 8 * 256:
        cblk = 16384
    else:
        nb = 8 * max(1, (n_cols + 4 * 28672) // (8 * 28672))
        cblk = max(((-(-n_cols // nb)) + 63) // 64 * 64, 2048)
    assert (info[0] & 0xffffffff) == int((lens >= thr).sum()) and info[2] == -(-n_cols // cblk) and (info[3] & 0xffffffff) == thr
    assert (info[3] >> 32) == (int(windows) if windows else 1)
    # tasks: one per 256 entries of every (long row, column block) segment
    want_tasks = 0
    for r in np.flatnonzero(lens >= thr):
        blk = col[rowptr[r]:rowptr[r + 1]] // cblk
        want_tasks += int(sum(-(-c // 256) for c in np.bincount(blk)))
    assert info[1] == want_tasks
    outs = []
    for rep in range(3):
        Y = torch.full((n_rows, 64), 7.0, device=dev())
        acc = to_dev(acc0.copy())
        _hip.check(L.skr_spmm_plan_run(h, _hip.ptr(dX), 64, _hip.ptr(dadd), _hip.ptr(Y), _hip.ptr(acc), 0.5, _hip.stream()))
        torch.cuda.synchronize()
        outs.append((Y.cpu().numpy(), acc.cpu().numpy()))
    for y, a in outs[1:]:
        assert np.array_equal(y.view(np.int32), outs[0][0].view(np.int32)) and np.array_equal(a.view(np.int32), outs[0][1].view(np.int32))
    want = A.astype(np.float64) @ X.astype(np.float64) + add
    mass = abs(A).astype(np.float64) @ np.abs(X).astype(np.float64) + np.abs(add)
    assert np.all(np.abs(outs[0][0] - want) <= 2e-6 * mass + 1e-6)
    assert np.all(np.abs(outs[0][1] - (acc0 + 0.5 * want)) <= 2e-6 * (mass + np.abs(acc0)) + 1e-6)
    # masks: entries of columns whose X rows are zero may be skipped; rows that are not needed may be left untouched
    cmask = (rng.random(n_cols) < 0.3).astype(np.uint8)
    rmask = (rng.random(n_rows) < 0.4).astype(np.uint8)
    X2 = X * cmask[:, None]
    dX2, d_cm, d_rm = to_dev(X2), to_dev(cmask), to_dev(rmask)
    Yfull = torch.empty((n_rows, 64), device=dev())
    _hip.check(L.skr_spmm_plan_run(h, _hip.ptr(dX2), 64, _hip.ptr(dadd), _hip.ptr(Yfull), None, 1.0, _hip.stream()))
    for use_r, use_c in ((False, True), (True, False), (True, True)):
        Y = torch.full((n_rows, 64), 7.0, device=dev())
        acc = to_dev(acc0.copy())
        _hip.check(L.skr_spmm_plan_run_masked(h, _hip.ptr(dX2), 64, _hip.ptr(dadd), _hip.ptr(Y), _hip.ptr(acc), 0.5,
                                              _hip.ptr(d_rm) if use_r else None, _hip.ptr(d_cm) if use_c else None, _hip.stream()))
        torch.cuda.synchronize()
        y, f, a = Y.cpu().numpy(), Yfull.cpu().numpy(), acc.cpu().numpy()
        need = rmask.astype(bool) if use_r else np.ones(n_rows, bool)
        mass2 = abs(A).astype(np.float64) @ np.abs(X2).astype(np.float64) + np.abs(add)
        assert np.all(np.abs(y[need] - f[need]) <= 1e-6 * mass2[need] + 1e-7)        # same sums, possibly another order
        assert np.all(y[~need] == 7.0) and np.array_equal(a[~need], acc0[~need])     # skipped rows are untouched
        assert np.all(np.abs(a[need] - (acc0[need] + 0.5 * f[need])) <= 1e-6 * (mass2[need] + np.abs(acc0[need])) + 1e-7)
    mk = torch.zeros(n_cols + 3, dtype=torch.uint8, device=dev())
    ids = to_dev(np.array([0, -1, n_cols - 1, 0], np.int32))
    _hip.check(L.skr_mark_ids(_hip.ptr(ids), 4, 3, _hip.ptr(mk), _hip.stream()))
    assert mk.nonzero().flatten().tolist() == sorted({3, n_cols + 2})
    Y = torch.empty((n_rows, 64), device=dev())
    assert L.skr_spmm_plan_run(h, _hip.ptr(dX), 32, None, _hip.ptr(Y), None, 1.0, _hip.stream()) == -1
    assert L.skr_spmm_plan_run(h, _hip.ptr(dX), 64, None, _hip.ptr(dX), None, 1.0, _hip.stream()) == -1
    assert L.skr_spmm_plan_run(None, _hip.ptr(dX), 64, None, _hip.ptr(Y), None, 1.0, _hip.stream()) == -1
    _hip.check(L.skr_spmm_plan_destroy(h))
    h2 = C.c_void_p()
    assert L.skr_spmm_plan_create(n_rows, n_cols, _hip.ptr(d_rp), _hip.ptr(d_col), _hip.ptr(d_val), nnz, 1, C.byref(h2), _hip.stream()) == -1
    assert L.skr_spmm_plan_create(n_rows, n_cols, None, _hip.ptr(d_col), _hip.ptr(d_val), nnz, 0, C.byref(h2), _hip.stream()) == -1


def test_layer_refine_fwd_bwd_vs_torch_autograd():
    import torch
    from gpu_utils import to_dev, dev
    from skrec import _hip
    rng = np.random.default_rng(1)
    n = 777
    Y = rng.standard_normal((n, 64)).astype(np.float32)
    E = rng.standard_normal((n, 64)).astype(np.float32)
    Y[5] = 0.0  # zero-degree node: propagated row is exactly zero
    dZ = rng.standard_normal((n, 64)).astype(np.float32)
    ty, te = torch.tensor(Y, requires_grad=True), torch.tensor(E, requires_grad=True)
    w = torch.nn.functional.cosine_similarity(ty, te, dim=-1)
    z = torch.einsum("a,ab->ab", w, ty)
    z.backward(torch.from_numpy(dZ))
    dY_, dE_, dZ_ = to_dev(Y), to_dev(E), to_dev(dZ)
    Z = torch.zeros((n, 64), device=dev())
    W = torch.zeros(n, device=dev())
    acc = torch.ones((n, 64), device=dev())
    L, st = _hip.lib(), _hip.stream()
    _hip.check(L.skr_layer_refine_fwd(_hip.ptr(dY_), _hip.ptr(dE_), n, 64, _hip.ptr(Z), _hip.ptr(W), _hip.ptr(acc), st))
    gY = torch.zeros((n, 64), device=dev())
    gE = torch.full((n, 64), 2.0, device=dev())
    _hip.check(L.skr_layer_refine_bwd(_hip.ptr(dY_), _hip.ptr(dE_), _hip.ptr(W), _hip.ptr(dZ_), n, 64, _hip.ptr(gY),
                                      _hip.ptr(gE), st))
    torch.cuda.synchronize()
    _close(W.cpu().numpy(), w.detach().numpy(), rtol=1e-5, atol=1e-6)
    _close(Z.cpu().numpy(), z.detach().numpy(), rtol=1e-5, atol=1e-6)
    _close(acc.cpu().numpy(), 1.0 + z.detach().numpy(), rtol=1e-5, atol=1e-6)
    _close(gY.cpu().numpy(), ty.grad.numpy(), rtol=2e-5, atol=2e-6)
    _close(gE.cpu().numpy(), 2.0 + te.grad.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("plan", ["0", "1"])
def test_spmm_row_epilogues_equal_the_separate_passes(plan, monkeypatch):
    """skr_spmm_plan_run_ex: the row-local passes in the product's row epilogue (layer mean with its E0 term, buffer
    initialisation, LayerGCN's refinement forward and backward, with row masks) == the plain product followed by the
    stand-alone kernels -- short rows (the row kernel's 16-lane layout) and long rows (the reduce kernel's 64-lane layout);
    plan "0": the plan-free composition DeviceCSR.spmm falls back to on small graphs.  skr_layer_refine_bwd_masked,
    skr_clear_marked_rows."""
    import torch
    from gpu_utils import dev, to_dev
    from skrec import _hip
    from skrec.recommender.LightGCN import DeviceCSR
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)
    L, st = _hip.lib(), _hip.stream()
    rng = np.random.default_rng(31)
    n = 2500
    A = sp.random(n, n, density=0.004, format="lil", random_state=rng, dtype=np.float32)
    for h in range(3):
        cols = rng.choice(n, 1200 + 300 * h, replace=False)
        A[h * 11 + 1, cols] = rng.standard_normal(len(cols)).astype(np.float32)
    A = sp.csr_matrix(A)
    A[n // 3, :] = 0
    A.eliminate_zeros()
    csr = DeviceCSR(A, dev())
    X, E, add, acc0, dE0 = (to_dev(rng.standard_normal((n, 64)).astype(np.float32)) for _ in range(5))
    mask_np = (rng.random(n) < 0.3).astype(np.uint8)
    mask_np[[1, 12, 23]] = [1, 0, 1]
    mask = to_dev(mask_np)
    z = lambda: torch.zeros((n, 64), device=dev())  # noqa: E731
    raw = z()
    csr.spmm(X, raw)
    rawa = raw + add

    def close(a, b, tol=2e-6):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.all(np.abs(a - b) <= tol * (1.0 + np.abs(b))), np.abs(a - b).max()
    # layer mean with its E0 term, and initialisation without reading
    Y, acc = z(), torch.full((n, 64), 9.0, device=dev())
    csr.spmm(X, Y, accum=acc, accum_scale=0.25, accum_base=E)
    close(Y, raw)
    close(acc, 0.25 * E + 0.25 * raw)
    acc = torch.full((n, 64), 9.0, device=dev())
    csr.spmm(X, None, addend=add, accum=acc, accum_scale=0.5, accum_init=True)
    close(acc, 0.5 * rawa)
    # forward refinement
    Zs, Ws, accs = z(), torch.zeros(n, device=dev()), acc0.clone()
    _hip.check(L.skr_layer_refine_fwd(_hip.ptr(raw), _hip.ptr(E), n, 64, _hip.ptr(Zs), _hip.ptr(Ws), _hip.ptr(accs), st))
    Y, Zf, Wf, accf = z(), z(), torch.zeros(n, device=dev()), acc0.clone()
    csr.spmm(X, Y, accum=accf, refine_fwd=(E, Wf, Zf))
    close(Y, raw); close(Zf, Zs); close(Wf, Ws); close(accf, accs)
    accf = torch.full((n, 64), 9.0, device=dev())
    csr.spmm(X, None, accum=accf, accum_init=True, refine_fwd=(E, Wf, Zf))
    close(accf, Zs)
    # ... with a row mask: rows outside it are left as they were (plan) or computed anyway (plan-free)
    Zm, Wm = torch.full((n, 64), 5.0, device=dev()), torch.full((n,), 5.0, device=dev())
    csr.spmm(X, None, row_mask=mask, refine_fwd=(E, Wm, Zm))
    sel = torch.from_numpy(mask_np.astype(bool)).to(dev())
    close(Zm[sel], Zs[sel]); close(Wm[sel], Ws[sel])
    if plan == "1":
        assert bool((Zm[~sel] == 5.0).all()) and bool((Wm[~sel] == 5.0).all())
    # backward refinement: the finished row (product + addend) is dZ
    dYs, dEs = z(), dE0.clone()
    _hip.check(L.skr_layer_refine_bwd(_hip.ptr(Y), _hip.ptr(E), _hip.ptr(Ws), _hip.ptr(rawa), n, 64, _hip.ptr(dYs), _hip.ptr(dEs), st))
    dYf, dEf = z(), dE0.clone()
    csr.spmm(X, dYf, addend=add, refine_bwd=(E, Ws, Y, dEf))
    close(dYf, dYs, 4e-6); close(dEf, dEs, 4e-6)
    # the masked stand-alone backward refinement, both treatments of the skipped rows
    for zs in (0, 1):
        dYm, dEm = torch.full((n, 64), 3.0, device=dev()), dE0.clone()
        _hip.check(L.skr_layer_refine_bwd_masked(_hip.ptr(Y), _hip.ptr(E), _hip.ptr(Ws), _hip.ptr(rawa), n, 64, _hip.ptr(dYm), _hip.ptr(dEm),
                                                 _hip.ptr(mask), zs, st))
        assert torch.equal(dYm[sel], dYs[sel]) and torch.equal(dEm[sel], dEs[sel])
        assert bool((dYm[~sel] == (0.0 if zs else 3.0)).all()) and torch.equal(dEm[~sel], dE0[~sel])
    # accum only where somebody reads it; an addend that is zero outside the marked rows is not read there
    acc = torch.full((n, 64), 9.0, device=dev())
    csr.spmm(X, Y, accum=acc, accum_scale=0.25, accum_base=E, accum_mask=mask)
    close(acc[sel], (0.25 * E + 0.25 * raw)[sel])
    if plan == "1":
        assert bool((acc[~sel] == 9.0).all())
    add_sparse = add * torch.from_numpy(mask_np).to(dev()).float().unsqueeze(1)
    poisoned = torch.where(sel.unsqueeze(1), add, torch.full_like(add, float("nan"))) if plan == "1" else add_sparse
    Y2 = z()
    csr.spmm(X, Y2, addend=poisoned, addend_mask=mask)       # (plan: the rows outside the mask must not even be read)
    close(Y2, raw + add_sparse)
    # the product with the transpose restricted to marked rows (skr_csr_scatter_marked_rows; float atomics: to rounding)
    Yt = add.clone()
    csr.scatter_marked_rows(mask, X, Yt)
    Xm = X.cpu().numpy().astype(np.float64) * mask_np[:, None]
    want_t = add.cpu().numpy() + A.T.astype(np.float64) @ Xm
    mass_t = np.abs(add.cpu().numpy()) + abs(A.T).astype(np.float64) @ np.abs(Xm)
    assert np.all(np.abs(Yt.cpu().numpy() - want_t) <= 2e-6 * mass_t + 1e-6)
    # skr_clear_marked_rows
    T, m2 = torch.ones((n, 64), device=dev()), mask.clone()
    _hip.check(L.skr_clear_marked_rows(_hip.ptr(m2), n, 1, _hip.ptr(T), 64, st))
    assert bool((T[sel] == 0).all()) and bool((T[~sel] == 1).all()) and int(m2.sum()) == 0
    _hip.check(L.skr_clear_marked_rows(_hip.ptr(mask), n, 0, _hip.ptr(T), 64, st))
    assert int(mask.sum()) == int(mask_np.sum())


def test_gather_axpy_scale():
    import torch
    from gpu_utils import to_dev, dev
    from skrec import _hip
    rng = np.random.default_rng(2)
    T = rng.standard_normal((100, 64)).astype(np.float32)
    idx = rng.integers(0, 100, 333).astype(np.int32)
    out = torch.zeros((333, 64), device=dev())
    L, st = _hip.lib(), _hip.stream()
    dT, didx = to_dev(T), to_dev(idx)
    _hip.check(L.skr_gather_rows(_hip.ptr(dT), _hip.ptr(didx), 333, 64, _hip.ptr(out), st))
    y = to_dev(T.copy())
    _hip.check(L.skr_axpy(0.5, _hip.ptr(dT), _hip.ptr(y), T.size, st))
    _hip.check(L.skr_scale(3.0, _hip.ptr(y), T.size, st))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), T[idx])
    _close(y.cpu().numpy(), (T + np.float32(0.5) * T) * np.float32(3.0), rtol=1e-6)


@pytest.mark.parametrize("lazy", [True, False])
@pytest.mark.parametrize("k,n_steps,t0,tf", [(8, 37, 0, False), (1, 5, 0, False), (16, 16, 0, False), (3, 10, 0, False), (8, 20, 16595, False),
                                             (8, 16, 40000, False), (32, 70, 0, False), (32, 64, 16580, False), (32, 32, 300, False),
                                             (64, 130, 0, False), (64, 64, 16570, False),
                                             # tf.train.AdamOptimizer's arithmetic (skr_adam_*_tf): |lr_t| falls until step ~20 and
                                             # rises afterwards -- blocks on both sides of the minimum and across it
                                             (8, 37, 0, True), (32, 70, 0, True), (32, 64, 10, True), (64, 130, 0, True),
                                             (32, 64, 16580, True), (16, 48, 700, True)])
def test_blocked_adam_is_bit_identical(k, n_steps, t0, tf, lazy):
    """temporally blocked dense Adam (cold blocks: k zero-gradient updates in one pass; hot blocks: the ordinary
    update every step -- or, `lazy`, when a batch is about to read them / has written their gradient, catching up
    on the zero-gradient steps in between) == skr_adam_step after every batch, BIT FOR BIT: parameters and both moments"""
    import torch
    from skrec import _hip
    from skrec.recommender.base import DenseAdam
    L, st = _hip.lib(), _hip.stream
    rng = np.random.default_rng(100 + k)
    nU, nI, b = 5000, 777, 256
    n_par = (nU + nI) * 64 + nI
    init = torch.from_numpy((rng.standard_normal(n_par) * 0.05).astype(np.float32)).cuda()
    # rows are distinct within a step (float atomics on a row then happen once: the gradients are deterministic
    # and the comparison isolates the optimiser); across steps they repeat freely
    u = torch.from_numpy(np.stack([rng.permutation(nU)[:b] for _ in range(n_steps)]).astype(np.int32)).cuda()
    ij = np.stack([rng.permutation(nI)[:2 * b] for _ in range(n_steps)]).astype(np.int32)
    i, j = torch.from_numpy(ij[:, :b].copy()).cuda(), torch.from_numpy(ij[:, b:].copy()).cuda()

    def views(opt):
        f, g = opt.flat, opt.grad
        sl = lambda t: (t[:nU * 64].view(nU, 64), t[nU * 64:(nU + nI) * 64].view(nI, 64), t[(nU + nI) * 64:])  # noqa: E731
        return sl(f), sl(g)

    def bpr(opt, s, loss, touch):
        (U, V, bias), (gU, gV, gb) = views(opt)
        _hip.check(L.skr_bpr_step(_hip.ptr(U), _hip.ptr(V), _hip.ptr(bias), _hip.ptr(U), _hip.ptr(V), _hip.ptr(u[s]), _hip.ptr(i[s]),
                                  _hip.ptr(j[s]), b, 1.0, 1e-3, 1.0, _hip.ptr(gU), _hip.ptr(gV), _hip.ptr(gb), _hip.ptr(gU),
                                  _hip.ptr(gV), _hip.ptr(loss), _hip.ptr(touch), _hip.ptr(opt.grad) if touch is not None else None,
                                  st()))

    # classic: one dense launch per step
    # t0: optimiser steps already taken (around 16 600 the second bias correction becomes exactly 1.0f and both forms
    # drop its division; the block starting at 16 595 straddles that point)
    a = DenseAdam(init.clone(), lr=1e-2, track_touch=True, tf_epsilon=tf)
    a.t = t0
    la = torch.zeros(2, device="cuda")
    for s in range(n_steps):
        bpr(a, s, la, a.touch)
        a.step()
    # blocked
    c = DenseAdam(init.clone(), lr=1e-2, tf_epsilon=tf)
    c.t = t0
    lc = torch.zeros(2, device="cuda")
    for s0 in range(0, n_steps, k):
        kk = min(k, n_steps - s0)
        uu, ii, jj = (t[s0:s0 + kk] for t in (u, i, j))                  # [kk, b] each: step-major after the cat on dim 1
        ids = torch.cat([uu, ii + nU, jj + nU, (ii >> 6) + (nU + nI), (jj >> 6) + (nU + nI)], dim=1).reshape(-1)
        c.begin_block(ids, kk, per_step=5 * b if lazy else None)
        for s in range(s0, s0 + kk):
            bpr(c, s, lc, None)
            c.hot_step()
    c.end_blocks()
    torch.cuda.synchronize()
    assert c.t == a.t == t0 + n_steps
    assert int((a.flat != c.flat).sum()) == 0
    assert torch.equal(a.m, c.m) and torch.equal(a.v, c.v)
    assert float(c.grad.abs().max()) == 0.0          # every gradient was consumed
    # the losses are sums of atomically accumulated terms: equal up to summation order
    np.testing.assert_allclose(la.cpu().numpy(), lc.cpu().numpy(), rtol=1e-5)


def _fused_case(k, n_blocks, t0, distinct, seed, nU=5000, nI=777, b=256, lr=1e-2):
    """classic (skr_bpr_step + one dense skr_adam_step per batch) and fused (skr_bpr_fused_step, one launch per batch)
    optimisers after the same n_blocks * k batches"""
    import torch
    from skrec import _hip
    from skrec.recommender.base import DenseAdam
    from skrec.recommender.fused import FusedBlocks
    L, st = _hip.lib(), _hip.stream
    rng = np.random.default_rng(seed)
    n_steps = k * n_blocks
    n_par = (nU + nI) * 64 + nI
    init = torch.from_numpy((rng.standard_normal(n_par) * 0.05).astype(np.float32)).cuda()
    if distinct:      # rows distinct within a step (each float atomic then happens once: deterministic gradients)
        u = np.stack([rng.permutation(nU)[:b] for _ in range(n_steps)])
        ij = np.stack([rng.permutation(nI)[:2 * b] for _ in range(n_steps)])
        i, j = ij[:, :b], ij[:, b:]
    else:             # popular items and repeated users inside a step, like real batches
        u = rng.integers(0, nU // 50, (n_steps, b))
        i = np.minimum((rng.pareto(1.2, (n_steps, b)) * 5).astype(np.int64), nI - 1)
        j = (i + 1 + rng.integers(0, nI - 1, (n_steps, b))) % nI
    u, i, j = (torch.from_numpy(np.ascontiguousarray(x).astype(np.int32)).cuda() for x in (u, i, j))
    a = DenseAdam(init.clone(), lr=lr, track_touch=True)
    a.t = t0
    la = torch.zeros((n_steps, 2), device="cuda")
    f, g = a.flat, a.grad
    sl = lambda t: (t[:nU * 64].view(nU, 64), t[nU * 64:(nU + nI) * 64].view(nI, 64), t[(nU + nI) * 64:])  # noqa: E731
    (U, V, bias), (gU, gV, gb) = sl(f), sl(g)
    for s in range(n_steps):
        _hip.check(L.skr_bpr_step(_hip.ptr(U), _hip.ptr(V), _hip.ptr(bias), _hip.ptr(U), _hip.ptr(V), _hip.ptr(u[s]), _hip.ptr(i[s]),
                                  _hip.ptr(j[s]), b, 1.0, 1e-3, 1.0, _hip.ptr(gU), _hip.ptr(gV), _hip.ptr(gb), _hip.ptr(gU),
                                  _hip.ptr(gV), _hip.ptr(la[s]), _hip.ptr(a.touch), _hip.ptr(a.grad), st()))
        a.step()
    c = DenseAdam(init.clone(), lr=lr)
    c.t = t0
    S = _hip.SKR_LOSS_SLOTS
    lc = torch.zeros((n_steps, S, 2), device="cuda")
    fb = FusedBlocks(c, 0, nU, nU + nI, 1e-3)
    fb.run_blocks(u.data_ptr(), i.data_ptr(), j.data_ptr(), n_blocks, k, b, lc.data_ptr(), 8 * S)
    c.end_blocks()
    torch.cuda.synchronize()
    assert c.t == a.t == t0 + n_steps
    return a, c, la, lc.sum(1), fb


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_blocks,t0", [(8, 3, 0), (1, 3, 0), (3, 4, 0), (32, 2, 0), (32, 2, 16580), (64, 1, 300), (64, 2, 16560),
                                           (16, 2, 40000)])
def test_fused_step_is_bit_identical(k, n_blocks, t0):
    """one launch per step (the hot rows' Adam evaluated lazily inside the BPR kernel, pending gradients applied at the
    row's next naming or by the block's end launch) == skr_bpr_step + a dense skr_adam_step after every batch, BIT FOR
    BIT: parameters and both moments.  Bias blocks are shared by many references of a step (the sharers each recompute
    the block's state, one owner writes it); item and user rows repeat across the steps of a block."""
    import torch
    a, c, la, lc, fb = _fused_case(k, n_blocks, t0, True, 300 + k)
    assert int((a.flat != c.flat).sum()) == 0
    assert torch.equal(a.m, c.m) and torch.equal(a.v, c.v)
    assert float(fb.work[:, 6 * fb.cap * 64:].abs().max()) == 0.0          # every gradient was consumed, the buffers are clear
    np.testing.assert_allclose(la.cpu().numpy(), lc.cpu().numpy(), rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("k,b,nU,nI", [(32, 256, 3000, 500), (64, 64, 100, 40), (1, 128, 50, 70), (7, 100, 20, 1000)])
def test_fused_plan_matches_the_sort_based_derivation(k, b, nU, nI):
    """skr_bpr_fused_plan (bit masks of the naming steps, atomics) against build_fused_meta (sort + scans): the same earlier-naming
    counts, previous steps, one owner per (step, row), and the same slot tables up to the numbering of the slots; the
    scratch is left zero"""
    import torch
    from skrec import _hip
    from skrec.recommender.fused import build_fused_meta
    L, st = _hip.lib(), _hip.stream
    rng = np.random.default_rng(k + b)
    u = torch.from_numpy(rng.integers(0, nU, k * b).astype(np.int32)).cuda()
    i = torch.from_numpy(np.minimum((rng.pareto(1.0, k * b) * 3).astype(np.int64), nI - 1).astype(np.int32)).cuda()
    j = torch.from_numpy(rng.integers(0, nI, k * b).astype(np.int32)).cuda()
    nfb = nU + nI + (nI + 63) // 64
    n_ref = k * 5 * b
    scratch = torch.zeros(28 * nfb // 8 + 1, dtype=torch.int64, device="cuda")
    meta, sb, sf = (torch.full((n_ref,), -7, dtype=torch.int32, device="cuda") for _ in range(3))
    ns = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    for _ in range(2):      # twice: the second call starts from the scratch the first one left
        _hip.check(L.skr_bpr_fused_plan(_hip.ptr(u), _hip.ptr(i), _hip.ptr(j), b, k, 0, nU, nU + nI, nfb, _hip.ptr(scratch),
                                        _hip.ptr(meta), _hip.ptr(sb), _hip.ptr(sf), _hip.ptr(ns), st()))
    torch.cuda.synchronize()
    assert int(scratch[:3 * nfb].abs().sum()) == 0          # the three masks per flat block; the slot numbers behind them may stay
    wm, wsb, wsf, wns = build_fused_meta(u, i, j, 1, k, b, 0, nU, nU + nI)
    n = int(ns)
    assert n == int(wns[0])
    got, want = meta.cpu().numpy().astype(np.int64), wm.reshape(-1).cpu().numpy().astype(np.int64)
    sole, got = (got >> 31) & 1, got & 0x7fffffff
    assert np.array_equal(got >> 24, want >> 24) and np.array_equal((got >> 20) & 7, (want >> 20) & 7)     # prev + 1, n0 mod 6
    gsb, wsb_ = sb.cpu().numpy(), wsb[0].cpu().numpy()
    assert (gsb[n:] == -1).all() and sorted(gsb[:n]) == list(wsb_[:n]) and len(set(gsb[:n])) == n
    # the same row behind every reference's slot; the slots' final words belong to the same rows
    assert np.array_equal(gsb[got & 0xfffff], wsb_[want & 0xfffff])
    order_g, order_w = np.argsort(gsb[:n]), np.argsort(wsb_[:n])
    assert np.array_equal(sf.cpu().numpy()[:n][order_g], wsf[0].cpu().numpy()[:n][order_w])
    # one owner per (step, row)
    step = np.repeat(np.arange(k), 5 * b)
    pair = gsb[got & 0xfffff].astype(np.int64) * 64 + step
    owners = pair[((got >> 23) & 1) == 1]
    assert len(owners) == len(np.unique(pair)) == len(np.unique(owners))
    # "sole": exactly the references whose (step, row) pair occurs once
    _, inv, cnt = np.unique(pair, return_inverse=True, return_counts=True)
    assert np.array_equal(sole == 1, cnt[inv] == 1)
    # skr_bpr_fused_plan2 with the previous block's tags: first namings at a step > 0 of rows that block did not touch carry
    # 7 in the n0 field ("pre-advanced"), every other word is unchanged
    prev_hot = torch.from_numpy(rng.random(nfb) < 0.4).cuda()
    tags = torch.where(prev_hot, torch.full((nfb,), 9, dtype=torch.int32, device="cuda"), torch.full((nfb,), 4, dtype=torch.int32, device="cuda"))
    meta2 = torch.full((n_ref,), -7, dtype=torch.int32, device="cuda")
    _hip.check(L.skr_bpr_fused_plan2(_hip.ptr(u), _hip.ptr(i), _hip.ptr(j), b, k, 0, nU, nU + nI, nfb, _hip.ptr(scratch),
                                     _hip.ptr(meta2), _hip.ptr(sb), _hip.ptr(sf), _hip.ptr(ns), _hip.ptr(tags), 9, st()))
    torch.cuda.synchronize()
    wm2, _, _, _ = build_fused_meta(u, i, j, 1, k, b, 0, nU, nU + nI, prev_hot=prev_hot.view(1, -1))
    got2, want2 = meta2.cpu().numpy().astype(np.int64) & 0x7fffffff, wm2.reshape(-1).cpu().numpy().astype(np.int64)
    assert np.array_equal((got2 >> 20) & 7, (want2 >> 20) & 7) and np.array_equal(got2 >> 24, want2 >> 24)
    flagged = ((got2 >> 20) & 7) == 7
    first = ((want >> 24) == 0) & (((want >> 20) & 7) == 0)                         # first namings (n0 = 0, no previous step)
    blk_of = sb.cpu().numpy()[got2 & 0xfffff]
    assert np.array_equal(flagged, first & (step > 0) & ~prev_hot.cpu().numpy()[blk_of]) and (flagged.any() or k == 1)
    # (bit 23, the owner among a pair's references, is decided by an atomic: it may fall on another reference in another call)
    assert np.array_equal((got2[~flagged] >> 20) & ~8, (got[~flagged] >> 20) & ~8)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_blocks", [(32, 2), (8, 4)])
def test_fused_step_with_rows_shared_inside_a_step(k, n_blocks):
    """popular items and repeated users inside one batch: several wavefronts of a launch read (and recompute) the same
    row, one of them writes it, all of them add into its gradient.  The float atomics then sum in an order that differs
    from launch to launch -- in the two-launch path too -- so the comparison is to rounding, not to the bit."""
    a, c, la, lc, fb = _fused_case(k, n_blocks, 0, False, 77 + k, lr=1e-3)
    np.testing.assert_allclose(c.flat.cpu().numpy(), a.flat.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(c.m.cpu().numpy(), a.m.cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(c.v.cpu().numpy(), a.v.cpu().numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(la.cpu().numpy(), lc.cpu().numpy(), rtol=1e-4)
    assert float(fb.work[:, 6 * fb.cap * 64:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("k,t0,eps,lr", [(8, 0, 1e-8, 1e-3), (16, 20000, 1e-8, 1e-3), (5, 16596, 1e-8, 5e-2), (8, 100, 0.0, 1e-3),
                                         (8, 3, 1e-3, 1e-3), (32, 0, 1e-8, 1e-3), (32, 16580, 1e-8, 1e-3), (32, 777, 1e-8, 1e-2),
                                         (32, 50000, 0.0, 1e-3), (64, 0, 1e-8, 1e-3), (64, 16570, 1e-8, 1e-3), (64, 900, 1e-8, 1e-2)])
def test_cold_pass_rest_regime_is_bit_identical(k, t0, eps, lr):
    """cold pass (rows at rest skip the square root and the divisions of the update, see adam_cold_rows_kernel) == k
    calls of skr_adam_step with a zero gradient, BIT FOR BIT, over rows of every age (moments decayed by 0 ... 3000
    untouched steps, down to denormal and zero) and over special values (+-0, denormals, powers of two, tiny and huge
    parameters, inf, NaN, v = +inf)"""
    import torch
    from skrec import _hip
    L, st = _hip.lib(), _hip.stream
    rng = np.random.default_rng(7 * k + t0)
    rows = 6000
    age = rng.integers(0, 3000, rows)
    age[:200] = np.arange(200) * 5                                   # a dense sweep across the regime boundaries
    m = (rng.standard_normal((rows, 64)) * 1e-3 * np.exp(np.log(0.9) * age)[:, None]).astype(np.float32)
    v = (rng.random((rows, 64)) * 1e-6 * np.exp(np.log(0.999) * age)[:, None]).astype(np.float32)
    p = (rng.standard_normal((rows, 64)) * 0.05).astype(np.float32)
    sp = rows - 400                                                   # special rows at the end
    m[sp:sp + 40] = 0.0
    v[sp:sp + 40] = 0.0                                               # never touched
    m[sp + 40:sp + 60] = np.float32(1e-44) * rng.integers(-6, 7, (20, 64)).astype(np.float32)   # stalled denormal moments
    m[sp + 60:sp + 70] = -0.0
    p[sp + 70:sp + 80] = -0.0
    p[sp + 80:sp + 90] = 0.0
    p[sp + 90:sp + 120] = (2.0 ** rng.integers(-70, 3, (30, 64))).astype(np.float32) * rng.choice([-1, 1], (30, 64))
    p[sp + 120:sp + 140] *= np.float32(1e-30)
    p[sp + 140:sp + 150] = np.float32(1e-42)
    p[sp + 150:sp + 160] *= np.float32(1e30)
    p[sp + 160:sp + 165] = np.inf
    p[sp + 165:sp + 170] = np.nan
    v[sp + 170:sp + 175] = np.inf
    v[sp + 175:sp + 180] = np.nan
    m[sp + 180:sp + 185] = np.inf
    m[sp + 185:sp + 190] = np.nan
    v[sp + 190:sp + 200] = np.float32(1e-44) * rng.integers(0, 9, (10, 64)).astype(np.float32)
    v[sp + 200:sp + 220] *= np.float32(1e-25)
    m[sp + 220:sp + 240, ::7] *= np.float32(1e20)                     # one lively lane keeps a row off the rest path
    v[sp + 240:sp + 260] = -0.0
    m[sp + 260:sp + 300] *= (10.0 ** rng.integers(-30, 30, (40, 64))).astype(np.float32)
    v[sp + 300:sp + 340] *= (10.0 ** rng.integers(-30, 30, (40, 64))).astype(np.float32)
    n = rows * 64 - 37                                                # a ragged tail
    flat = lambda x: torch.from_numpy(x.reshape(-1)[:n].copy()).cuda()   # noqa: E731
    p0, m0, v0 = flat(p), flat(m), flat(v)
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    g = torch.zeros(n, device="cuda")
    for s in range(k):
        _hip.check(L.skr_adam_step(_hip.ptr(pa), _hip.ptr(g), _hip.ptr(ma), _hip.ptr(va), n, lr, 0.9, 0.999, eps, t0 + 1 + s, 0,
                                   None, st()))
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    tag = torch.zeros((n + 63) // 64, dtype=torch.int32, device="cuda")
    tag[17] = 1                                                       # one hot block: must be left alone
    _hip.check(L.skr_adam_block_cold(_hip.ptr(pb), _hip.ptr(mb), _hip.ptr(vb), n, lr, 0.9, 0.999, eps, t0, k, _hip.ptr(tag), 1, st()))
    torch.cuda.synchronize()
    bits = lambda t: t.view(torch.int32)                              # noqa: E731  (sign-of-zero-exact)
    hot = slice(17 * 64, 18 * 64)
    for x0, xa, xb in ((p0, pa, pb), (m0, ma, mb), (v0, va, vb)):
        assert torch.equal(bits(xb)[hot], bits(x0)[hot])
        xa2 = xa.clone()
        xa2[hot] = x0[hot]
        differ = (bits(xa2) != bits(xb)) & ~(torch.isnan(xa2) & torch.isnan(xb))    # any NaN equals any NaN: which sign /
        assert int(differ.sum()) == 0                                               # payload survives is the compiler's choice
    # the premise of the rest path, on the reference results themselves: old rows no longer move
    if eps > 0:
        old = torch.from_numpy(np.repeat(age[:sp] >= 400, 64)).cuda()
        assert torch.equal(bits(pa)[:sp * 64][old], bits(p0)[:sp * 64][old])


@pytest.mark.gpu
def test_cold_pass_ordinary_math_matches_compiler_forms():
    """the scaling-free square root over every float of its range and the scaling-free division over 2^32 hashed
    operand pairs of its range == sqrtf / the fp32 division as compiled for the ordinary update"""
    import ctypes as C
    from skrec import _hip
    _hip.require_gpu()
    bad = (C.c_uint64 * 4)()
    _hip.check(_hip.lib().skr_selftest_cold_math(1 << 32, bad, _hip.stream()))
    assert bad[3] == 0x7f7fffff - 0x0f800000 + 1          # every float of [2^-96, FLT_MAX] was tried
    assert bad[2] > 1000                                   # control: the raw hardware root is NOT sqrtf
    assert bad[0] == 0 and bad[1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,cap,n_items", [(2, 64, 500), (8, 2048, 100000), (3, 300, 200), (16, 128, 90), (5, 1, 10)])
def test_unpack_grad_rows_sorted_equals_per_rank_launches(n_ranks, cap, n_items):
    """one-launch ordered sum of the ranks' packed gradient rows == the n_ranks launches in rank order, bit for bit
    (dense rows, bias words, touch bytes); packs built the way the exchange builds them (unique_padded_rows)"""
    import torch
    from skrec import _hip
    from skrec.parallel import unique_padded_rows
    L, st = _hip.lib(), _hip.stream
    g = torch.Generator(device="cuda").manual_seed(n_ranks * 1000 + cap)
    raw = torch.randint(0, n_items, (n_ranks, cap), generator=g, device="cuda", dtype=torch.int32)
    raw[:, cap // 2:] = torch.where(torch.rand((n_ranks, cap - cap // 2), generator=g, device="cuda") < 0.3,
                                    torch.full_like(raw[:, cap // 2:], -1), raw[:, cap // 2:])     # some empty slots
    ids = unique_padded_rows(raw)
    assert bool(((ids[:, 1:] > ids[:, :-1]) | (ids[:, 1:] < 0)).all())      # ascending, then -1
    packs = torch.randn((n_ranks, cap, 66), generator=g, device="cuda")
    packs[:, :, 0] = ids.view(torch.float32)
    gV0 = torch.randn((n_items, 64), generator=g, device="cuda")
    gb0 = torch.randn(n_items, generator=g, device="cuda")
    flat = lambda: torch.cat([gV0.reshape(-1), gb0]).clone()           # noqa: E731  ([V | b] as in the optimiser's buffer)
    outs = []
    for fn in (L.skr_unpack_grad_rows, L.skr_unpack_grad_rows_sorted):
        buf = flat()
        gV, gb = buf[:n_items * 64].view(n_items, 64), buf[n_items * 64:]
        touch = torch.zeros((buf.numel() + 63) // 64, dtype=torch.uint8, device="cuda")
        _hip.check(fn(_hip.ptr(packs), cap, n_ranks, _hip.ptr(gV), _hip.ptr(gb), 64, _hip.ptr(touch), _hip.ptr(buf), st()))
        torch.cuda.synchronize()
        outs.append((buf, touch))
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], flat())                             # something was added


@pytest.mark.gpu
def test_blocked_adam_lists_may_hold_empty_slots():
    """negative ids in the lists of skr_adam_block_mark / skr_adam_block_hot are skipped (de-duplicated lists carry
    -1 in their empty slots); argument checks of the blocked-Adam entry points"""
    import torch
    from skrec import _hip
    L, st = _hip.lib(), _hip.stream
    n = 64 * 50
    g0 = torch.Generator(device="cuda").manual_seed(5)
    p0, m0, v0 = (torch.randn(n, generator=g0, device="cuda") * 0.1 for _ in range(3))
    v0 = v0.abs() * 1e-4
    grad0 = torch.randn(n, generator=g0, device="cuda") * 1e-2

    def run(ids):
        p, m, v, g = p0.clone(), m0.clone(), v0.clone(), grad0.clone()
        tag = torch.zeros(50, dtype=torch.int32, device="cuda")
        claim = torch.zeros(50, dtype=torch.int32, device="cuda")
        d = torch.tensor(ids, dtype=torch.int32, device="cuda")
        _hip.check(L.skr_adam_block_mark(_hip.ptr(d), d.numel(), 0, 64, _hip.ptr(tag), 7, _hip.ptr(claim), 10, st()))
        _hip.check(L.skr_adam_block_hot(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 10, 11,
                                        _hip.ptr(d), d.numel(), 0, 64, _hip.ptr(claim), st()))
        torch.cuda.synchronize()
        return p, m, v, g, tag
    a = run([3, 17, 17, 41])
    b = run([-1, 3, -1, 17, 41, -1, 17, -5])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert a[4].nonzero().flatten().tolist() == [3, 17, 41]
    assert float(a[3][3 * 64:4 * 64].abs().max()) == 0.0 and float(a[3][0:64].abs().max()) > 0.0   # consumed / untouched
    # argument checks
    d = torch.tensor([1], dtype=torch.int32, device="cuda")
    claim = torch.zeros(50, dtype=torch.int32, device="cuda")
    p, m, v, g = p0.clone(), m0.clone(), v0.clone(), grad0.clone()
    args = (_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8)
    assert L.skr_adam_block_hot(*args, 10, 10, _hip.ptr(d), 1, 0, 64, _hip.ptr(claim), st()) == -1      # step_t must exceed step_t0
    assert L.skr_adam_block_hot(*args, 10, 75, _hip.ptr(d), 1, 0, 64, _hip.ptr(claim), st()) == -1      # more than 64 steps
    assert L.skr_adam_block_hot(*args, 10, 11, None, 1, 0, 64, _hip.ptr(claim), st()) == -1
    tag = torch.zeros(50, dtype=torch.int32, device="cuda")
    assert L.skr_adam_block_cold(_hip.ptr(p), _hip.ptr(m), _hip.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0, 65, _hip.ptr(tag), 1, st()) == -1
    assert L.skr_adam_block_cold(_hip.ptr(p), _hip.ptr(m), _hip.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0, 0, _hip.ptr(tag), 1, st()) == -1
    # the one-launch step: k and s ranges, workspace capacity (slot numbers have 20 bits), NULL pointers
    w = torch.zeros(9 * 64 * 64, device="cuda")
    i32 = torch.zeros(64, dtype=torch.int32, device="cuda")
    loss = torch.zeros(64, device="cuda")
    fs = lambda cap, k, s_, meta=i32: L.skr_bpr_fused_step(_hip.ptr(p), _hip.ptr(m), _hip.ptr(v), n, _hip.ptr(w), cap, _hip.ptr(i32),  # noqa: E731
                                                          _hip.ptr(i32), _hip.ptr(i32), _hip.ptr(meta) if meta is not None else None, 1,
                                                          0, 10, 40, 1e-3, 0.9, 0.999, 1e-8, 0, k, s_, 1e-3, _hip.ptr(loss), st())
    assert fs(64, 65, 0) == -1 and fs(64, 4, 4) == -1 and fs(64, 4, -1) == -1 and fs(0, 4, 0) == -1 and fs((1 << 20) + 1, 4, 0) == -1
    assert fs(64, 4, 0, None) == -1
    scratch = torch.zeros(28 * 50 // 8 + 1, dtype=torch.int64, device="cuda")
    fp = lambda b_, k: L.skr_bpr_fused_plan(_hip.ptr(i32), _hip.ptr(i32), _hip.ptr(i32), b_, k, 0, 10, 40, 50, _hip.ptr(scratch),  # noqa: E731
                                            _hip.ptr(i32), _hip.ptr(i32), _hip.ptr(i32), _hip.ptr(i32), st())
    assert fp(1, 65) == -1 and fp(0, 1) == -1 and fp(1 << 18, 1) == -1          # k * 5 * n_batch beyond 2^20
    assert L.skr_bpr_fused_end(_hip.ptr(p), _hip.ptr(m), _hip.ptr(v), n, _hip.ptr(w), 64, _hip.ptr(i32), _hip.ptr(i32), None, 1e-3, 0.9,
                               0.999, 1e-8, 0, 4, None, 0, 0, st()) == -1
    assert L.skr_bpr_fused_end(_hip.ptr(p), _hip.ptr(m), _hip.ptr(v), n, _hip.ptr(w), 64, _hip.ptr(i32), _hip.ptr(i32), _hip.ptr(i32), 1e-3,
                               0.9, 0.999, 1e-8, 0, 4, None, 0, 1, st()) == -1          # which = 1 needs the next block's tags
    torch.cuda.synchronize()
