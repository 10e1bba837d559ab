"""GPU parity, SURVEY 8f-4 (GRU4RecPlus): the HIP kernels of csrc/gru.hip against oracle/gru4rec.py -- a
torch-CPU restatement of the reference's TensorFlow-1.14 graph with autograd supplying the gradients the
kernels derive by hand.  PARITY UNPINNED against the reference itself (TensorFlow absent, no recorded
output in the reference).  Tolerance: 1e-5 relative (fp32 summation order) unless noted."""
import os

import numpy as np
import pytest
import torch

from oracle import gru4rec as G
from gpu_utils import to_dev
from skrec import _hip

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _cell(rng, i_d, h):
    lim_g, lim_c = np.sqrt(6.0 / (i_d + 3 * h)), np.sqrt(6.0 / (i_d + 2 * h))
    return (rng.uniform(-lim_g, lim_g, (i_d + h, 2 * h)).astype(np.float32), (1 + 0.1 * rng.standard_normal(2 * h)).astype(np.float32),
            rng.uniform(-lim_c, lim_c, (i_d + h, h)).astype(np.float32), (0.1 * rng.standard_normal(h)).astype(np.float32))


def _close(got, want, rtol=1e-5, atol=1e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(np.abs(want).max(), 1e-30)
    assert np.abs(got - want).max() <= rtol * scale + atol, (np.abs(got - want).max(), scale)


@pytest.mark.parametrize("i_d,h,B,act", [(64, 64, 128, "tanh"), (128, 128, 37, "tanh"), (64, 128, 16, "relu"), (48, 32, 5, "tanh")])
def test_gru_cell_fwd_bwd(i_d, h, B, act):
    rng = np.random.default_rng(i_d + h + B)
    n_rows = 300
    table = (0.5 * rng.standard_normal((n_rows, i_d))).astype(np.float32)
    idx = rng.integers(0, n_rows, B).astype(np.int32)
    hprev = (0.5 * rng.standard_normal((B, h))).astype(np.float32)
    Wg, bg, Wc, bc = _cell(rng, i_d, h)
    G_out = rng.standard_normal((B, h)).astype(np.float32)
    # oracle
    tt = [torch.tensor(a, requires_grad=True) for a in (table, Wg, bg, Wc, bc)]
    hn = G.gru_cell(tt[0][torch.as_tensor(idx, dtype=torch.long)], torch.tensor(hprev), tt[1], tt[2], tt[3], tt[4], act)
    (hn * torch.tensor(G_out)).sum().backward()
    # device
    L, st = _hip.lib(), _hip.stream()
    d = {k: to_dev(v) for k, v in dict(table=table, idx=idx, h=hprev, Wg=Wg, bg=bg, Wc=Wc, bc=bc, dh=G_out).items()}
    r, u, c, out = (torch.empty((B, h), device="cuda") for _ in range(4))
    kind = {"tanh": 0, "relu": 1}[act]
    _hip.check(L.skr_gru_cell_fwd(_hip.ptr(d["table"]), _hip.ptr(d["idx"]), _hip.ptr(d["h"]), None, B, i_d, h, _hip.ptr(d["Wg"]),
                                  _hip.ptr(d["bg"]), _hip.ptr(d["Wc"]), _hip.ptr(d["bc"]), kind, _hip.ptr(r), _hip.ptr(u),
                                  _hip.ptr(c), _hip.ptr(out), st))
    _close(out.cpu().numpy(), hn.detach().numpy())
    gWg, gbg, gWc, gbc = (torch.zeros_like(d[k]) for k in ("Wg", "bg", "Wc", "bc"))
    dx = torch.empty((B, i_d), device="cuda")
    work = torch.empty(3 * B * h, device="cuda")
    _hip.check(L.skr_gru_cell_bwd(_hip.ptr(d["table"]), _hip.ptr(d["idx"]), _hip.ptr(d["h"]), B, i_d, h, _hip.ptr(d["Wg"]),
                                  _hip.ptr(d["Wc"]), kind, _hip.ptr(r), _hip.ptr(u), _hip.ptr(c), _hip.ptr(d["dh"]),
                                  _hip.ptr(gWg), _hip.ptr(gbg), _hip.ptr(gWc), _hip.ptr(gbc), _hip.ptr(dx), _hip.ptr(work), st))
    _close(gWg.cpu().numpy(), tt[1].grad.numpy(), 2e-5)
    _close(gbg.cpu().numpy(), tt[2].grad.numpy(), 2e-5)
    _close(gWc.cpu().numpy(), tt[3].grad.numpy(), 2e-5)
    _close(gbc.cpu().numpy(), tt[4].grad.numpy(), 2e-5)
    want_dx = np.zeros((n_rows, i_d), np.float32)       # autograd gives the scattered table gradient
    gtab = torch.zeros((n_rows, i_d), device="cuda")
    _hip.check(L.skr_scatter_add_rows(_hip.ptr(dx), _hip.ptr(d["idx"]), B, i_d, None, 0.0, _hip.ptr(gtab), None, None, st))
    _close(gtab.cpu().numpy(), tt[0].grad.numpy(), 2e-5)
    # the one-call form of the first layer (skr_gru_cell_bwd_scatter): the scatter rides in the weight-gradient launch -- the
    # weights' gradients and dx bit for bit (same threads, same order), the atomically scattered table gradient to its
    # tolerance, here with the l2 term of the looked-up rows (repeated inputs count each time)
    g2 = [torch.zeros_like(d[k]) for k in ("Wg", "bg", "Wc", "bc")]
    dx2 = torch.empty((B, i_d), device="cuda")
    gtab2 = torch.zeros((n_rows, i_d), device="cuda")
    reg = 0.03
    _hip.check(L.skr_gru_cell_bwd_scatter(_hip.ptr(d["table"]), _hip.ptr(d["idx"]), _hip.ptr(d["h"]), B, i_d, h, _hip.ptr(d["Wg"]),
                                          _hip.ptr(d["Wc"]), kind, _hip.ptr(r), _hip.ptr(u), _hip.ptr(c), _hip.ptr(d["dh"]),
                                          _hip.ptr(g2[0]), _hip.ptr(g2[1]), _hip.ptr(g2[2]), _hip.ptr(g2[3]), _hip.ptr(dx2),
                                          _hip.ptr(work), reg, _hip.ptr(gtab2), None, None, st))
    torch.cuda.synchronize()
    for a, b in zip(g2, (gWg, gbg, gWc, gbc)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert np.array_equal(dx2.cpu().numpy(), dx.cpu().numpy())
    counts = np.bincount(idx, minlength=n_rows).astype(np.float32)
    _close(gtab2.cpu().numpy(), tt[0].grad.numpy() + reg * counts[:, None] * table, 2e-5)
    # the row mask of the inference sweep: inactive rows keep their state
    active = (rng.random(B) < 0.5).astype(np.uint8)
    out2 = torch.empty((B, h), device="cuda")
    d_act = to_dev(active)
    _hip.check(L.skr_gru_cell_fwd(_hip.ptr(d["table"]), _hip.ptr(d["idx"]), _hip.ptr(d["h"]), _hip.ptr(d_act), B, i_d, h,
                                  _hip.ptr(d["Wg"]), _hip.ptr(d["bg"]), _hip.ptr(d["Wc"]), _hip.ptr(d["bc"]), kind, None, None,
                                  None, _hip.ptr(out2), st))
    want = np.where(active[:, None] == 1, out.cpu().numpy(), hprev)
    assert np.array_equal(out2.cpu().numpy(), want)


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("i_d,h,B,act", [(128, 128, 5000, "tanh"), (64, 64, 2049, "tanh"), (100, 32, 3001, "relu"), (37, 64, 4100, "tanh"),
                                         (64, 128, 2100, "tanh")])
def test_gru_cell_forward_for_many_sessions(i_d, h, B, act, mfma, monkeypatch):
    """the inference sweep's forward (more than 2048 sessions per call): the matrix-core kernel (v_mfma_f32_32x32x2_f32, 32
    sessions per workgroup) and the vector kernel it replaces (SKR_GRU_MFMA=0, 16 sessions per workgroup) against the
    torch-CPU cell; r, u, c outputs; the row mask that carries finished histories' states; a ragged last workgroup; an odd
    input size (the k-pairs' zero tail).  The switch is read once per process, so the vector kernel runs in a child process."""
    import subprocess
    import sys
    if mfma == "0":
        code = ("import os, sys; os.environ['SKR_GRU_MFMA'] = '0'; sys.path[:0] = [%r, %r]; import test_gpu_gru as t; "
                "t._many_sessions_case(%d, %d, %d, %r)" % (os.path.dirname(os.path.abspath(__file__)), os.path.join(
                    os.path.dirname(os.path.abspath(__file__)), "..", "scikit-recommender_amd"), i_d, h, B, act))
        subprocess.run([sys.executable, "-c", code], check=True, cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        return
    _many_sessions_case(i_d, h, B, act)


def _many_sessions_case(i_d, h, B, act):
    rng = np.random.default_rng(i_d + h + B)
    n_rows = 700
    table = (0.5 * rng.standard_normal((n_rows, i_d))).astype(np.float32)
    idx = rng.integers(0, n_rows, B).astype(np.int32)
    hprev = (0.5 * rng.standard_normal((B, h))).astype(np.float32)
    Wg, bg, Wc, bc = _cell(rng, i_d, h)
    hn = G.gru_cell(torch.tensor(table)[torch.as_tensor(idx, dtype=torch.long)], torch.tensor(hprev), torch.tensor(Wg), torch.tensor(bg),
                    torch.tensor(Wc), torch.tensor(bc), act).numpy()
    L, st = _hip.lib(), _hip.stream()
    d = {k: to_dev(v) for k, v in dict(table=table, idx=idx, h=hprev, Wg=Wg, bg=bg, Wc=Wc, bc=bc).items()}
    r, u, c, out = (torch.empty((B, h), device="cuda") for _ in range(4))
    kind = {"tanh": 0, "relu": 1}[act]
    _hip.check(L.skr_gru_cell_fwd(_hip.ptr(d["table"]), _hip.ptr(d["idx"]), _hip.ptr(d["h"]), None, B, i_d, h, _hip.ptr(d["Wg"]),
                                  _hip.ptr(d["bg"]), _hip.ptr(d["Wc"]), _hip.ptr(d["bc"]), kind, _hip.ptr(r), _hip.ptr(u),
                                  _hip.ptr(c), _hip.ptr(out), st))
    _close(out.cpu().numpy(), hn, 2e-5, 2e-6)
    # the saved gates are consistent with the state: h' = u h + (1 - u) c
    rr, uu, cc = (t.cpu().numpy() for t in (r, u, c))
    _close(uu * hprev + (1.0 - uu) * cc, hn, 2e-5, 2e-6)
    assert (rr > 0).all() and (rr < 1).all() and (uu > 0).all() and (uu < 1).all()
    # dense input rows (no index), no saved gates, the row mask
    active = (rng.random(B) < 0.5).astype(np.uint8)
    x_dense = to_dev(table[idx])
    out2 = torch.empty((B, h), device="cuda")
    _hip.check(L.skr_gru_cell_fwd(_hip.ptr(x_dense), None, _hip.ptr(d["h"]), _hip.ptr(to_dev(active)), B, i_d, h, _hip.ptr(d["Wg"]),
                                  _hip.ptr(d["bg"]), _hip.ptr(d["Wc"]), _hip.ptr(d["bc"]), kind, None, None, None, _hip.ptr(out2), st))
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), np.where(active[:, None] == 1, out.cpu().numpy(), hprev))


@pytest.mark.parametrize("loss", ["bpr_max", "top1_max"])
@pytest.mark.parametrize("fact", ["linear", "relu", "leaky_relu"])
@pytest.mark.parametrize("B,S,h", [(128, 2048, 64), (7, 0, 32), (33, 100, 128)])
def test_session_loss_and_grads(loss, fact, B, S, h):
    rng = np.random.default_rng(B + S + h)
    n_items, n_y = 500, B + S
    E = (0.3 * rng.standard_normal((n_items, h))).astype(np.float32)
    bias = (0.1 * rng.standard_normal(n_items)).astype(np.float32)
    out = (0.5 * rng.standard_normal((B, h))).astype(np.float32)
    Y = rng.integers(0, n_items, n_y).astype(np.int32)          # repeated targets on purpose
    reg, bpr_reg = 0.01, 0.7
    tE, tb, to = (torch.tensor(a, requires_grad=True) for a in (E, bias, out))
    Yl = torch.as_tensor(Y, dtype=torch.long)
    items, bs = tE[Yl], tb[Yl]
    logits = G.final_act(to @ items.t() + bs, fact)
    main = G.bpr_max_loss(logits, bpr_reg) if loss == "bpr_max" else G.top1_max_loss(logits)
    (main + reg * 0.5 * (items.pow(2).sum() + bs.pow(2).sum())).backward()
    L, st = _hip.lib(), _hip.stream()
    dE, db, do, dY = to_dev(E), to_dev(bias), to_dev(out), to_dev(Y)
    dlog = torch.empty((B, n_y), device="cuda")
    dout = torch.empty((B, h), device="cuda")
    lossbuf = torch.zeros(1, device="cuda")
    fk, lk = {"linear": 0, "relu": 1, "leaky_relu": 2}[fact], {"bpr_max": 0, "top1_max": 1}[loss]
    _hip.check(L.skr_session_loss(_hip.ptr(do), B, h, _hip.ptr(dE), _hip.ptr(db), _hip.ptr(dY), n_y, fk, lk, bpr_reg,
                                  _hip.ptr(dlog), _hip.ptr(dout), _hip.ptr(lossbuf), st))
    assert abs(float(lossbuf) - float(main.detach())) <= 2e-5 * abs(float(main.detach())) + 1e-7
    _close(dout.cpu().numpy(), to.grad.numpy(), 5e-5, 1e-7)
    gE, gb = torch.zeros_like(dE), torch.zeros_like(db)
    _hip.check(L.skr_session_out_grads(_hip.ptr(dlog), _hip.ptr(do), B, h, _hip.ptr(dY), n_y, _hip.ptr(dE), _hip.ptr(db), reg,
                                       _hip.ptr(gE), _hip.ptr(gb), None, None, st))
    _close(gE.cpu().numpy(), tE.grad.numpy(), 5e-5, 1e-7)
    _close(gb.cpu().numpy(), tb.grad.numpy(), 5e-5, 1e-7)
    # the one-call form (dL/dout and the output-side gradients in ONE launch, loss word and dout cleared by the call itself):
    # same dlogits bit for bit, the atomically summed outputs to their tolerance -- also from dirty output buffers
    dlog2 = torch.full((B, n_y), 3.0, device="cuda")
    dout2 = torch.full((B, h), 5.0, device="cuda")
    loss2 = torch.full((1,), 7.0, device="cuda")
    gE2, gb2 = torch.zeros_like(dE), torch.zeros_like(db)
    _hip.check(L.skr_session_loss_grads(_hip.ptr(do), B, h, _hip.ptr(dE), _hip.ptr(db), _hip.ptr(dY), n_y, fk, lk, bpr_reg,
                                        _hip.ptr(dlog2), _hip.ptr(dout2), _hip.ptr(loss2), 0, B, reg, _hip.ptr(gE2), _hip.ptr(gb2),
                                        None, None, st))
    torch.cuda.synchronize()
    assert np.array_equal(dlog2.cpu().numpy(), dlog.cpu().numpy())
    assert abs(float(loss2) - float(lossbuf)) <= 1e-6 * abs(float(lossbuf)) + 1e-9
    _close(dout2.cpu().numpy(), dout.cpu().numpy(), 2e-6, 1e-8)
    _close(gE2.cpu().numpy(), gE.cpu().numpy(), 2e-6, 1e-8)
    _close(gb2.cpu().numpy(), gb.cpu().numpy(), 2e-6, 1e-8)


@pytest.mark.parametrize("layers,loss,fact,act,reg", [([64], "bpr_max", "linear", "tanh", 0.0),
                                                       ([128], "bpr_max", "leaky_relu", "tanh", 1e-4),
                                                       ([64, 32], "top1_max", "linear", "relu", 1e-3)])
def test_training_trajectory_matches_oracle(layers, loss, fact, act, reg):
    """20 session-parallel steps with carried states and occasional resets: per-step loss and the final
    parameters against the oracle's autograd + TF-style Adam"""
    from skrec.recommender.GRU4RecPlus import SessionGRU
    rng = np.random.default_rng(len(layers) * 7 + layers[0])
    n_items, b, S = 400, 32, 200
    E_in = rng.normal(0, 0.1, (n_items, layers[0])).astype(np.float32)
    E_out = rng.normal(0, 0.1, (n_items, layers[-1])).astype(np.float32)
    b_out = rng.normal(0, 0.05, n_items).astype(np.float32)
    cells, i_d = [], layers[0]
    for h in layers:
        cells.append(_cell(rng, i_d, h))
        i_d = h
    kw = dict(hidden_act=act, loss=loss, bpr_reg=0.8, reg=reg, lr=3e-3)
    o = G.GRU4RecOracle(E_in, cells, E_out, b_out, final=fact, **kw)
    net = SessionGRU(E_in, cells, E_out, b_out, final_act=fact, **kw)
    st_o = [torch.zeros(b, h) for h in layers]
    st_d = net.zero_states(b)
    for step in range(20):
        X = rng.integers(0, n_items, b).astype(np.int32)
        Y = np.concatenate([rng.integers(0, n_items, b), rng.integers(0, n_items, S)]).astype(np.int32)
        lo, st_o = o.train_step(X, Y, st_o)
        st_d = net.train_step(to_dev(X), to_dev(Y), st_d)
        assert abs(float(net.loss) - lo) <= 5e-5 * abs(lo) + 1e-6, (step, float(net.loss), lo)
        for a, w in zip(st_d, st_o):
            _close(a.cpu().numpy(), w.numpy(), 1e-4, 1e-6)
        if step % 7 == 6:                    # sessions ending: their state rows are cleared (GRU4RecPlus.py:244-246)
            mask = rng.choice(b, 5, replace=False)
            st_o = [s.clone() for s in st_o]
            for s in st_o:
                s[mask] = 0
            st_d = [s.index_fill(0, to_dev(mask.astype(np.int64)), 0.0) for s in st_d]
    _close(net.E_in.cpu().numpy(), o.E_in.detach().numpy(), 2e-4, 2e-6)
    _close(net.E_out.cpu().numpy(), o.E_out.detach().numpy(), 2e-4, 2e-6)
    _close(net.b_out.cpu().numpy(), o.b_out.detach().numpy(), 2e-4, 2e-6)
    for (Wg, bg, Wc, bc), oc in zip(net.cells, o.cells):
        for a, w in zip((Wg, bg, Wc, bc), oc):
            _close(a.cpu().numpy(), w.detach().numpy(), 2e-4, 2e-6)


@pytest.mark.parametrize("layers,in_dim", [([128], 128), ([64, 32], 32)])
def test_blocked_adam_steps_are_bit_identical_to_dense_steps(layers, in_dim):
    """SessionGRU.begin_block: the TF-arithmetic dense Adam blocked over k steps (cold rows in one pass on a side stream,
    the step's rows by the hot launch) == one dense skr_adam_step_tf per step, BIT FOR BIT -- every parameter, both
    moments, the recurrent states, the losses; blocks of 5, 32 and 3 steps with state resets in between; item rows that
    repeat across the steps of a block and rows shared by input and target lists"""
    from skrec.recommender.GRU4RecPlus import SessionGRU
    rng = np.random.default_rng(77)
    n_items, b, n_s = 3000, 16, 40
    E_in = rng.normal(0, 0.1, (n_items, in_dim)).astype(np.float32)
    E_out = rng.normal(0, 0.1, (n_items, layers[-1])).astype(np.float32)
    cells, i_d = [], in_dim
    for h in layers:
        cells.append(_cell(rng, i_d, h))
        i_d = h
    args = (E_in, cells, E_out, np.zeros(n_items, np.float32), "tanh", "linear", "bpr_max", 1.0, 1e-4, 1e-2)
    dense, blocked = SessionGRU(*args), SessionGRU(*args)
    steps = 40
    # a small id range: rows repeat ACROSS steps; distinct within a step, so that no float atomic meets another and the
    # gradients are deterministic (the comparison isolates the optimiser)
    X = np.stack([rng.permutation(200)[:b] for _ in range(steps)]).astype(np.int32)
    Y = np.stack([np.concatenate([rng.permutation(200)[:b], 200 + rng.permutation(n_items - 200)[:n_s]]) for _ in range(steps)]).astype(np.int32)
    dX, dY = to_dev(X), to_dev(Y)
    sd, sb = dense.zero_states(b), blocked.zero_states(b)
    ld, lb = [], []
    for s_ in range(steps):
        if s_ == 7:
            sd = [t.index_fill(0, torch.tensor([2, 9], device="cuda"), 0.0) for t in sd]
        sd = dense.train_step(dX[s_], dY[s_], sd)
        ld.append(float(dense.loss.cpu()))
    s_ = 0
    for k in (5, 32, 3):
        blocked.begin_block(dX[s_:s_ + k], dY[s_:s_ + k])
        for _ in range(k):
            if s_ == 7:
                sb = [t.index_fill(0, torch.tensor([2, 9], device="cuda"), 0.0) for t in sb]
            sb = blocked.train_step(dX[s_], dY[s_], sb)
            lb.append(float(blocked.loss.cpu()))
            s_ += 1
    blocked.end_blocks()
    torch.cuda.synchronize()
    assert blocked.opt.t == dense.opt.t == steps
    np.testing.assert_allclose(lb, ld, rtol=2e-6)        # (the step's loss is a sum of float atomics over the sessions)
    assert torch.equal(blocked.flat, dense.flat) and torch.equal(blocked.opt.m, dense.opt.m) and torch.equal(blocked.opt.v, dense.opt.v)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(sd, sb))
    assert float(blocked.opt.grad.abs().max()) == 0.0


def test_user_embeddings_sweep_matches_oracle():
    from skrec.recommender.GRU4RecPlus import SessionGRU
    rng = np.random.default_rng(11)
    n_items, n_users, layers = 300, 57, [64, 64]
    E_in = rng.normal(0, 0.3, (n_items, 64)).astype(np.float32)
    E_out = rng.normal(0, 0.3, (n_items, 64)).astype(np.float32)
    cells = [_cell(rng, 64, 64), _cell(rng, 64, 64)]
    lens = rng.integers(0, 9, n_users)
    lens[3] = 0
    rowptr = np.zeros(n_users + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    hist = rng.integers(0, n_items, int(rowptr[-1])).astype(np.int32)
    o = G.GRU4RecOracle(E_in, cells, E_out, np.zeros(n_items, np.float32))
    net = SessionGRU(E_in, cells, E_out, np.zeros(n_items, np.float32))
    got = net.user_embeddings(to_dev(rowptr), to_dev(hist), int(lens.max())).cpu().numpy()
    want = o.user_embeddings(rowptr, hist)
    _close(got, want, 2e-5, 1e-6)
    assert np.all(got[3] == 0)


def test_gru4recplus_fit_through_the_api(tiny_dir, monkeypatch, tmp_path):
    """the drop-in class: constructor, fit (session-parallel loop), evaluate, predict; losses fall, the
    evaluator's fused and generic paths agree on the report"""
    import random
    from skrec import RunConfig
    from skrec.recommender.GRU4RecPlus import GRU4RecPlus
    monkeypatch.chdir(tmp_path)
    np.random.seed(2021); random.seed(2021); torch.manual_seed(2021)
    rc = RunConfig(recommender="GRU4RecPlus", data_dir=tiny_dir, file_column="UIRT", sep="\t",
                   metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20), test_batch_size=16, seed=2021)
    m = GRU4RecPlus(rc, dict(lr=0.01, layers=[64], batch_size=16, n_sample=64, epochs=3, early_stop=10))
    first = None
    losses = []
    te = m.train_epoch

    def train_epoch():
        te()
        losses.append(float(np.mean(m.step_losses)))
    m.train_epoch = train_epoch
    best = m.fit()
    assert len(losses) == 3 and losses[-1] < losses[0] and np.isfinite(losses).all()
    assert set(best.metrics()) == {f"{n}@{k}" for n in ("Precision", "Recall", "MAP", "NDCG", "MRR") for k in (5, 10, 20)}
    fused = m.evaluate()
    monkeypatch.setattr(GRU4RecPlus, "predict_factors", lambda self: None)
    generic = m.evaluate()
    np.testing.assert_allclose(np.array(list(fused.values())), np.array(list(generic.values())), rtol=1e-5, atol=2e-4)
    p = m.predict([0, 5, 9])
    assert p.shape == (3, m.items_num) and p.dtype == np.float32
    ue = m.cur_user_embeddings.cpu().numpy()
    want = ue[[0, 5, 9]] @ m.net.E_out.cpu().numpy().T + m.net.b_out.cpu().numpy()
    np.testing.assert_allclose(p, want, rtol=1e-4, atol=1e-5)


def _gru_fit(tiny_dir, workdir, epochs=2):
    import random
    from skrec import RunConfig
    from skrec.recommender.GRU4RecPlus import GRU4RecPlus
    os.chdir(workdir)
    np.random.seed(2021); random.seed(2021); torch.manual_seed(2021)
    rc = RunConfig(recommender="GRU4RecPlus", data_dir=tiny_dir, file_column="UIRT", sep="\t",
                   metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20), test_batch_size=16, seed=2021)
    m = GRU4RecPlus(rc, dict(lr=0.01, layers=[64], batch_size=16, n_sample=64, epochs=epochs, early_stop=10))
    losses, reports = [], []
    te, ev = m.train_epoch, m.evaluate

    def train_epoch():
        te()
        losses.append(m.step_losses.copy())

    def evaluate(test_users=None):
        r = ev(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        return r
    m.train_epoch, m.evaluate = train_epoch, evaluate
    m.fit()
    return m, np.concatenate(losses), np.stack(reports)


def test_fit_with_the_blocked_optimiser_equals_a_dense_launch_per_step(tiny_dir, tmp_path, monkeypatch):
    """GRU4RecPlus.fit(): steps prepared 32 at a time with the TF-Adam blocked over them (the default) == SKR_ADAM_BLOCK=1
    (one step at a time, one dense launch per step): the same sessions in the same order with the same negatives (numpy's
    stream is consumed identically), per-step losses, reports"""
    monkeypatch.setenv("SKR_ADAM_BLOCK", "1")
    m1, l1, r1 = _gru_fit(tiny_dir, str(tmp_path))
    monkeypatch.setenv("SKR_ADAM_BLOCK", "32")
    m2, l2, r2 = _gru_fit(tiny_dir, str(tmp_path))
    assert len(l1) == len(l2) and len(l1) > 40
    # (float atomics inside a step -- an item drawn twice -- may add in another order from run to run)
    np.testing.assert_allclose(l2, l1, rtol=2e-5)
    np.testing.assert_allclose(r2, r1, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(m2.net.flat.cpu().numpy(), m1.net.flat.cpu().numpy(), rtol=0, atol=2e-5)


def _gru_api_worker(rank, world, port, tiny_dir, workdir, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      SKR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    m, losses, reports = _gru_fit(tiny_dir, workdir)
    ret[rank] = dict(losses=losses, reports=reports, flat=m.net.flat.cpu().numpy())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_gru4recplus_fit_on_two_ranks(tiny_dir, tmp_path):
    """`torchrun ... run_skrec.py --recommender GRU4RecPlus` on two ranks: the parallel sessions split over the ranks, the
    optimiser blocked on every rank alike, the inference sweep and the evaluation sharded (users u % 2; fp64 metric sums
    all-reduced) == the single-process fit; replicas bit-identical"""
    import torch.multiprocessing as mp
    from test_gpu_dist import _free_port
    _, l1, r1 = _gru_fit(tiny_dir, str(tmp_path))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gru_api_worker, args=(2, _free_port(), tiny_dir, str(tmp_path), ret), nprocs=2, join=True)
        res = {k: ret[k] for k in range(2)}
    for r in res.values():
        np.testing.assert_allclose(r["losses"], l1, rtol=5e-5)
        np.testing.assert_allclose(r["reports"], r1, rtol=1e-4, atol=2e-4)
    assert np.array_equal(res[0]["flat"], res[1]["flat"]) and np.array_equal(res[0]["reports"], res[1]["reports"])


def test_pop_sampler_is_numpys_searchsorted():
    """skr_pop_sample (GRU4RecPlus.py:198-200): with the host's uniforms == np.searchsorted(pop_cumsum, u), including
    u exactly on a boundary and u = 0; with the device generator: in range and following the popularity law"""
    from skrec import _hip
    L = _hip.lib()
    rng = np.random.default_rng(3)
    n_items = 5000
    pop = np.power(rng.integers(1, 1000, n_items).astype(np.float64), 0.75)
    cs = np.cumsum(pop)
    cs = cs / cs[-1]
    u = rng.random(4096)
    u[:5] = [0.0, cs[0], cs[17], cs[-2], np.nextafter(1.0, 0.0)]
    d_cs, d_u = to_dev(cs), to_dev(u)
    out = torch.empty(len(u), dtype=torch.int32, device="cuda")
    _hip.check(L.skr_pop_sample(_hip.ptr(d_cs), n_items, _hip.ptr(d_u), 0, len(u), _hip.ptr(out), _hip.stream()))
    assert np.array_equal(out.cpu().numpy(), np.searchsorted(cs, u))
    big = torch.empty(400_000, dtype=torch.int32, device="cuda")
    _hip.check(L.skr_pop_sample(_hip.ptr(d_cs), n_items, None, 99, big.numel(), _hip.ptr(big), _hip.stream()))
    got = big.cpu().numpy()
    assert got.min() >= 0 and got.max() < n_items
    freq = np.bincount(got, minlength=n_items) / len(got)
    want = np.diff(np.concatenate([[0.0], cs]))
    assert np.abs(freq - want).max() < 5 * np.sqrt(want.max() / len(got)) + 1e-4


def _sharded_gru_worker(rank, world, port, ret, rccl_one=False):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from skrec.parallel import DistContext
    from skrec.recommender.GRU4RecPlus import SessionGRU, ShardedSessionGRU
    if world > 1 and rccl_one == "rccl":      # one process per GPU on RCCL
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        rccl_one = False
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if rccl_one:      # a group of one rank on RCCL, forced through the sharded engine's collectives
        os.environ["SKR_DIST_FORCE_ACTIVE"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        world = 2     # the branches below: "sharded"
    rng = np.random.default_rng(21)
    n_items, b, n_s, steps = 400, 24, 40, 12
    E_in = rng.normal(0, 0.1, (n_items, 64)).astype(np.float32)
    E_out = rng.normal(0, 0.1, (n_items, 128)).astype(np.float32)
    cells = [_cell(rng, 64, 64), _cell(rng, 64, 128)]
    args = (E_in, cells, E_out, np.zeros(n_items, np.float32), "tanh", "linear", "bpr_max", 1.0, 1e-3, 1e-2)
    net = ShardedSessionGRU(DistContext(rank, 1 if rccl_one else world), *args) if world > 1 else SessionGRU(*args)
    lo, hi = net.slots(b) if world > 1 else (0, b)
    states = net.zero_states(hi - lo)
    losses = []
    for s in range(steps):
        x = rng.integers(0, n_items, b).astype(np.int32)
        y = np.concatenate([rng.integers(0, n_items, b), rng.integers(0, n_items, n_s)]).astype(np.int32)
        y[3] = y[b + 1]                                   # a sampled negative equal to a positive
        if s == 5:                                        # sessions 2 and 20 end: their states are reset
            for g_ in (2, 20):
                if lo <= g_ < hi:
                    states = [st.index_fill(0, torch.tensor([g_ - lo], device="cuda"), 0.0) for st in states]
        states = net.train_step(to_dev(x), to_dev(y), states)
        losses.append(float(net.loss.cpu()))
    torch.cuda.synchronize()
    ret[rank] = dict(losses=np.array(losses), flat=net.flat.cpu().numpy(), lo=lo, hi=hi, state=states[-1].cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_session_sharded_engine_on_a_single_rank_rccl_group():
    """the sharded engine's exchange (all_gather_into_tensor of the compact block, rank-ordered sum) on backend "nccl" with a
    group of one rank == the single-process engine"""
    import torch.multiprocessing as mp
    from test_gpu_dist import _free_port
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sharded_gru_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        one = ret[0]
        ret2 = mgr.dict()
        mp.spawn(_sharded_gru_worker, args=(1, _free_port(), ret2, True), nprocs=1, join=True)
        r = ret2[0]
    np.testing.assert_allclose(r["losses"], one["losses"], rtol=2e-5)
    np.testing.assert_allclose(r["flat"], one["flat"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(r["state"], one["state"], rtol=0, atol=3e-5)


@pytest.mark.parametrize("world", [2, 3])
def test_session_sharded_steps_equal_the_single_process_steps(world):
    """ShardedSessionGRU (row f-4, BASELINE configs[4]): the b parallel sessions split over the ranks, ONE compact exchange
    per step (target rows of both output tables' gradients, input rows, GRU gradients, loss; all-gathered and added in rank
    order) == the single-process steps: per-step losses, every parameter, the recurrent states; replicas bit-identical"""
    import torch.multiprocessing as mp
    from test_gpu_dist import _free_port
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sharded_gru_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        one = ret[0]
        ret2 = mgr.dict()
        mp.spawn(_sharded_gru_worker, args=(world, _free_port(), ret2), nprocs=world, join=True)
        many = {k: ret2[k] for k in range(world)}
    for r in range(world):
        np.testing.assert_allclose(many[r]["losses"], one["losses"], rtol=2e-5)
        # (the ranks' gradient blocks are added in another order than one process adds them, and Adam at lr = 1e-2 turns a
        #  last-bit difference of a tiny gradient into up to ~1e-5 of a parameter)
        np.testing.assert_allclose(many[r]["flat"], one["flat"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(many[r]["state"], one["state"][many[r]["lo"]:many[r]["hi"]], rtol=0, atol=3e-5)
        assert np.array_equal(many[r]["flat"], many[0]["flat"])          # replicas: the same bits


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL branch needs two GPUs (one process per GPU); the test boxes of "
                    "this environment have one -- it runs as soon as a multi-GPU box executes the suite")
def test_session_sharded_steps_on_rccl():
    """ShardedSessionGRU on two RCCL ranks (all_gather_into_tensor of the compact block between two GPUs) == the
    single-process steps; replicas bit-identical"""
    import torch.multiprocessing as mp
    from test_gpu_dist import _free_port
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sharded_gru_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        one = ret[0]
        ret2 = mgr.dict()
        mp.spawn(_sharded_gru_worker, args=(2, _free_port(), ret2, "rccl"), nprocs=2, join=True)
        many = {k: ret2[k] for k in range(2)}
    for r in range(2):
        np.testing.assert_allclose(many[r]["losses"], one["losses"], rtol=2e-5)
        np.testing.assert_allclose(many[r]["flat"], one["flat"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(many[r]["state"], one["state"][many[r]["lo"]:many[r]["hi"]], rtol=0, atol=3e-5)
    assert np.array_equal(many[0]["flat"], many[1]["flat"])


def test_config4_shape_steps_match_oracle():
    """BASELINE configs[4]'s shape -- d = 128, 100 000 items, 128 parallel sessions walking 50-event histories, 2 048
    popularity^0.75 negatives per step (skr_pop_sample on numpy's uniforms), bpr_max -- for the steps around a session
    boundary (positions 47, 48 of one group of sessions, then 0, 1 of the next with fresh states): per-step loss, the
    recurrent state and every touched parameter row against the torch-CPU restatement (parity unpinned, see the header)"""
    from skrec.recommender.GRU4RecPlus import SessionGRU
    L = _hip.lib()
    rng = np.random.default_rng(4)
    n_items, d, b, n_s, T = 100_000, 128, 128, 2048, 50
    E_in = np.clip(rng.normal(0, 0.01, (n_items, d)), -0.02, 0.02).astype(np.float32)
    E_out = np.clip(rng.normal(0, 0.01, (n_items, d)), -0.02, 0.02).astype(np.float32)
    cells = [_cell(rng, d, d)]
    o = G.GRU4RecOracle(E_in, cells, E_out, np.zeros(n_items, np.float32), loss="bpr_max", bpr_reg=1.0, reg=1e-5, lr=1e-3)
    net = SessionGRU(E_in, cells, E_out, np.zeros(n_items, np.float32), loss="bpr_max", bpr_reg=1.0, reg=1e-5, lr=1e-3)
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.9
    cs = np.cumsum(pop ** 0.75)
    cs /= cs[-1]
    d_cs = to_dev(cs)
    sessions = rng.choice(n_items, (2, b, T), p=pop / pop.sum()).astype(np.int32)
    st_o, st_d = [torch.zeros(b, d)], net.zero_states(b)
    touched = set()
    for blk, t in ((0, 46), (0, 47), (0, 48), (1, 0), (1, 1)):
        if t == 0:
            st_o, st_d = [torch.zeros(b, d)], net.zero_states(b)
        u = rng.random(n_s)
        neg = torch.empty(n_s, dtype=torch.int32, device="cuda")
        _hip.check(L.skr_pop_sample(_hip.ptr(d_cs), n_items, _hip.ptr(to_dev(u)), 0, n_s, _hip.ptr(neg), _hip.stream()))
        neg_h = neg.cpu().numpy()
        assert np.array_equal(neg_h, np.searchsorted(cs, u))
        X, Y = sessions[blk, :, t], np.concatenate([sessions[blk, :, t + 1], neg_h]).astype(np.int32)
        lo, st_o = o.train_step(X, Y, st_o)
        st_d = net.train_step(to_dev(X), to_dev(Y), st_d)
        assert abs(float(net.loss) - lo) <= 5e-5 * abs(lo) + 1e-6, (blk, t, float(net.loss), lo)
        _close(st_d[0].cpu().numpy(), st_o[0].numpy(), 1e-4, 1e-6)
        touched.update(X.tolist())
        touched.update(Y.tolist())
    rows = np.array(sorted(touched))
    _close(net.E_in.cpu().numpy()[rows], o.E_in.detach().numpy()[rows], 2e-4, 2e-6)
    _close(net.E_out.cpu().numpy()[rows], o.E_out.detach().numpy()[rows], 2e-4, 2e-6)
    _close(net.b_out.cpu().numpy()[rows], o.b_out.detach().numpy()[rows], 2e-4, 2e-6)
    for a, w in zip(net.cells[0], o.cells[0]):
        _close(a.cpu().numpy(), w.detach().numpy(), 2e-4, 2e-6)
