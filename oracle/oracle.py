"""oracle/oracle.py -- TEST INFRASTRUCTURE (parity oracle), never on the product path.

ctypes bindings for
  * ``oracle/liboracle.so``          -- our plain-C restatement (``oracle/skr_oracle.c``), and
  * ``oracle/_ref/libskrec_ref.so``  -- the reference's own C++ headers compiled where they lie
                                        (``oracle/ref_wrap.cpp``), present only when built here,
plus numpy/torch-CPU restatements of the *Python* half of the reference's hot path (iterator
layouts, evaluator host logic, BPRMF / LightGCN / LayerGCN step maths).  Each function cites the
reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    ref = os.path.join(_HERE, "_ref", "libskrec_ref.so")
    need = force or not os.path.exists(so) or \
        os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "skr_oracle.c")) or \
        (os.path.isdir("/root/reference") and not os.path.exists(ref))
    if need:
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_sampler_new.restype = C.c_void_p
        L.orc_sampler_new.argtypes = [C.c_uint32]
        L.orc_sampler_free.argtypes = [C.c_void_p]
        L.orc_sampler_reseed.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_sampler_draws.restype = C.c_uint64
        L.orc_sampler_draws.argtypes = [C.c_void_p]
        L.orc_sampler_next_u32.restype = C.c_uint32
        L.orc_sampler_next_u32.argtypes = [C.c_void_p]
        L.orc_sampler_get_state.argtypes = [C.c_void_p, _u32p, C.POINTER(C.c_int)]
        L.orc_sampler_set_state.argtypes = [C.c_void_p, _u32p, C.c_int]
        L.orc_randint_choice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, _i32p]
        L.orc_sample_epoch.argtypes = [C.c_void_p, C.c_int, C.c_int, _i64p, _i32p, C.c_int, _i32p]
        L.orc_partial_sort_ids.argtypes = [_f32p, C.c_int, C.c_int, _i32p]
        L.orc_topk_ids_lowid.argtypes = [_f32p, C.c_int, C.c_int, _i32p]
        L.orc_evaluate_matrix.argtypes = [_f32p, C.c_int, C.c_int, _i64p, _i32p, _i32p, C.c_int,
                                          C.c_int, _f32p, C.c_void_p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libskrec_ref.so"))


def ref():
    """The reference's own C++ (compiled where it lies).  Raises if it was never built."""
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(os.path.join(_HERE, "_ref", "libskrec_ref.so"))
        R.ref_reseed.argtypes = [C.c_uint]
        R.ref_randint_choice.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, _i32p]
        R.ref_batch_randint_choice.argtypes = [C.c_int, _i32p, C.c_int, C.c_int, C.c_void_p, _i64p, _i32p,
                                               C.c_int, C.c_int, _i32p]
        R.ref_sample_epoch.argtypes = [C.c_int, C.c_int, _i64p, _i32p, C.c_int, _i32p]
        R.ref_evaluate_matrix.argtypes = [_f32p, C.c_int, C.c_int, _i64p, _i32p, _i32p, C.c_int, C.c_int,
                                          C.c_int, _f32p]
        _ref = R
    return _ref


# ------------------------------------------------------------------------------------------------
# S: sampler
# ------------------------------------------------------------------------------------------------
class Sampler:
    """One MT19937(2020) stream = the reference's process-global ``_gen`` (randint.h:20)."""

    def __init__(self, seed=2020):
        self._h = lib().orc_sampler_new(seed)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_sampler_free(self._h)
            self._h = None

    def reseed(self, seed=2020):
        lib().orc_sampler_reseed(self._h, seed)

    @property
    def draws(self):
        return int(lib().orc_sampler_draws(self._h))

    def next_u32(self):
        return int(lib().orc_sampler_next_u32(self._h))

    def get_state(self):
        w = np.zeros(624, np.uint32)
        p = C.c_int(0)
        lib().orc_sampler_get_state(self._h, w, C.byref(p))
        return w, p.value

    def set_state(self, words, pos):
        lib().orc_sampler_set_state(self._h, np.ascontiguousarray(words, np.uint32), int(pos))

    def randint_choice(self, high, size=1, replace=True, p=None, exclusion=None):
        """pyx_randint_choice (pyx_random.pyx:20-76) incl. its checks and scalar return."""
        if high <= 1:
            raise ValueError("'high' must be larger than 1.")
        if size <= 0:
            raise ValueError("'size' must be a positive integer.")
        if not isinstance(replace, bool):
            raise TypeError("'replace' must be bool.")
        pp = None
        if p is not None:
            pp = np.array(p, dtype=np.float32)
            if pp.ndim != 1:
                raise ValueError("'p' must be a 1-dim array_like")
            if len(pp) != high:
                raise ValueError("The length of 'p' must be equal with 'high'.")
        if exclusion is not None and not isinstance(exclusion, (int, np.integer)) and len(exclusion) >= high:
            raise ValueError("The length of 'exclusion' must be smaller than 'high'.")
        if isinstance(exclusion, (int, np.integer)):
            exclusion = [exclusion]
        n_ex = len(exclusion) if exclusion is not None else 0
        if replace is False and (high - n_ex <= size):
            raise ValueError("There is not enough integers to be sampled.")
        ex = np.ascontiguousarray(exclusion, np.int32) if exclusion is not None else None
        out = np.zeros(size, np.int32)
        lib().orc_randint_choice(self._h, high, size, int(replace),
                                 pp.ctypes.data if pp is not None else None,
                                 ex.ctypes.data if ex is not None and n_ex else None,
                                 n_ex, int(exclusion is not None), out)
        return out[0] if size == 1 else out

    def sample_epoch(self, num_items, rowptr, pos_items, num_neg=1):
        """_sampling_negative_items (data_iterator.py:81-94) -> int32 [E] or [E, num_neg]."""
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        pos_items = np.ascontiguousarray(pos_items, np.int32)
        out = np.zeros(len(pos_items) * num_neg, np.int32)
        lib().orc_sample_epoch(self._h, num_items, len(rowptr) - 1, rowptr, pos_items, num_neg, out)
        return out if num_neg == 1 else out.reshape(-1, num_neg)


def ref_randint_choice(high, size, replace=True, p=None, exclusion=None):
    pp = np.ascontiguousarray(p, np.float32) if p is not None else None
    ex = np.ascontiguousarray(exclusion, np.int32) if exclusion is not None else None
    out = np.zeros(size, np.int32)
    ref().ref_randint_choice(high, size, int(replace), pp.ctypes.data if pp is not None else None,
                             ex.ctypes.data if ex is not None and len(ex) else None,
                             len(ex) if ex is not None else 0, int(exclusion is not None), out)
    return out


def ref_sample_epoch(num_items, rowptr, pos_items, num_neg=1):
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    pos_items = np.ascontiguousarray(pos_items, np.int32)
    out = np.zeros(len(pos_items) * num_neg, np.int32)
    ref().ref_sample_epoch(num_items, len(rowptr) - 1, rowptr, pos_items, num_neg, out)
    return out if num_neg == 1 else out.reshape(-1, num_neg)


def pairwise_epoch(users_ary, pos_items, neg_items, batch_size, perm=None, drop_last=False):
    """PairwiseIterator.__iter__ batching (data_iterator.py:226-234; batch_iterator.py:98-106):
    consecutive slices of one permutation (identity when shuffle=False)."""
    n = len(users_ary)
    idx = np.arange(n) if perm is None else np.asarray(perm)
    out = []
    for s in range(0, n, batch_size):
        sl = idx[s:s + batch_size]
        if len(sl) < batch_size and drop_last:
            break
        out.append((users_ary[sl], pos_items[sl], neg_items[sl]))
    return out


def pointwise_layout(users_ary, pos_items, neg_items, num_neg):
    """PointwiseIterator (data_iterator.py:159-166, 175-181): users tiled num_neg+1 times,
    items = positives ++ negatives in negative-slot-major order, labels 1.0 then 0.0."""
    all_users = np.tile(users_ary, num_neg + 1)
    neg = np.asarray(neg_items).reshape(len(pos_items), -1) if num_neg > 1 else np.asarray(neg_items)
    neg_flat = neg.transpose().reshape([-1])
    all_items = np.concatenate([pos_items, neg_flat], axis=0)
    labels = np.concatenate([np.ones(len(pos_items), np.float32),
                             np.zeros(len(pos_items) * num_neg, np.float32)])
    return all_users, all_items, labels


# ------------------------------------------------------------------------------------------------
# E: evaluation
# ------------------------------------------------------------------------------------------------
METRIC2ID = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}  # evaluator.py:57


def _csr_from_lists(lists):
    rowptr = np.zeros(len(lists) + 1, np.int64)
    for i, l in enumerate(lists):
        rowptr[i + 1] = rowptr[i] + len(l)
    items = np.concatenate([np.asarray(l, np.int32) for l in lists]) if len(lists) and rowptr[-1] else \
        np.zeros(0, np.int32)
    return rowptr, np.ascontiguousarray(items, np.int32)


def eval_score_matrix(scores, test_items, metric_ids, top_k, return_ids=False):
    """eval_score_matrix (pyx_eval_matrix.pyx:22-37) via our C restatement."""
    scores = np.ascontiguousarray(scores, np.float32)
    rowptr, items = _csr_from_lists(test_items)
    m = np.ascontiguousarray(metric_ids, np.int32)
    out = np.zeros((scores.shape[0], len(m) * top_k), np.float32)
    ids = np.zeros((scores.shape[0], top_k), np.int32) if return_ids else None
    if len(items) == 0:
        items = np.zeros(1, np.int32)
    rc = lib().orc_evaluate_matrix(scores, scores.shape[0], scores.shape[1], rowptr, items, m, len(m), top_k,
                                   out, ids.ctypes.data if ids is not None else None)
    if rc != 0:
        raise ValueError("bad metric id or top_k > n_items")
    return (out, ids) if return_ids else out


def ref_eval_score_matrix(scores, test_items, metric_ids, top_k, thread_num=1):
    scores = np.ascontiguousarray(scores, np.float32).copy()
    rowptr, items = _csr_from_lists(test_items)
    if len(items) == 0:
        items = np.zeros(1, np.int32)
    m = np.ascontiguousarray(metric_ids, np.int32)
    out = np.zeros((scores.shape[0], len(m) * top_k), np.float32)
    ref().ref_evaluate_matrix(scores, scores.shape[0], scores.shape[1], rowptr, items, m, len(m), top_k,
                              thread_num, out)
    return out


def topk_ids_heap(row, top_k):
    """arg-top-K in the reference's heap order (evaluate.h:27-45)."""
    row = np.ascontiguousarray(row, np.float32)
    ids = np.zeros(min(2 * top_k, len(row)), np.int32)
    lib().orc_partial_sort_ids(row, len(row), top_k, ids)
    return ids[:top_k]


def topk_ids_lowid(row, top_k):
    """arg-top-K under the HIP path's documented tie rule (score desc, then lower id)."""
    row = np.ascontiguousarray(row, np.float32)
    ids = np.zeros(top_k, np.int32)
    lib().orc_topk_ids_lowid(row, len(row), top_k, ids)
    return ids


def ranking_evaluate(predict, user_train, user_test, metric=None, top_k=50, batch_size=256, test_users=None):
    """RankingEvaluator.evaluate (evaluator.py:163-214).  ``predict(list_of_users) -> fp32 [B, I]``.
    user_train / user_test: dict user -> int array.  Returns (names, float32 values, per-user rows)."""
    if metric is None:
        metric = ["Precision", "Recall", "MAP", "NDCG", "MRR"]
    elif isinstance(metric, str):
        metric = [metric]
    ids = [METRIC2ID[m] for m in metric]
    if isinstance(top_k, int):
        max_top, top_show = top_k, np.arange(top_k) + 1
    else:
        max_top, top_show = max(top_k), np.sort(top_k)
    users = list(user_test.keys()) if test_users is None else [u for u in test_users if u in user_test]
    rows = []
    for s in range(0, len(users), batch_size):
        bu = users[s:s + batch_size]
        sc = np.array(predict(bu), dtype=np.float32, copy=True)
        for i, u in enumerate(bu):
            if u in user_train and len(user_train[u]) > 0:
                sc[i][user_train[u]] = -np.inf
        rows.append(eval_score_matrix(sc, [user_test[u] for u in bu], ids, max_top))
    allr = np.concatenate(rows, axis=0)
    final = np.mean(allr, axis=0)  # float32 mean (evaluator.py:208)
    final = final.reshape(len(ids), max_top)[:, top_show - 1].reshape(-1)
    id2m = {v: k for k, v in METRIC2ID.items()}
    names = [f"{id2m[i]}@{k}" for i in ids for k in top_show]
    return names, final, allr


# ------------------------------------------------------------------------------------------------
# T: model maths (numpy fp32, explicit gradients -- no autograd), pinned by tests/golden/*.npz
# ------------------------------------------------------------------------------------------------
def _softplus_neg(x):
    """-logsigmoid(x) the way torch evaluates it: -(min(0,x) - log1p(exp(-|x|)))  (utils/torch.py:63)"""
    x = x.astype(np.float32)
    return -(np.minimum(np.float32(0), x) - np.log1p(np.exp(-np.abs(x)))).astype(np.float32)


def _sigmoid_neg(x):
    z = np.exp(-np.abs(x)).astype(np.float32)
    return np.where(x >= 0, z / (1 + z), 1 / (1 + z)).astype(np.float32)


class Adam:
    """torch.optim.Adam, single-tensor path (dense; every element every step) -- BPRMF.py:99,127."""

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.p = params
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0

    def step(self, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = np.float32(self.lr / bc1)
        bc2s = np.float32(bc2 ** 0.5)
        f = np.float32
        for p, g, m, v in zip(self.p, grads, self.m, self.v):
            m += f(1.0 - self.b1) * (g - m)
            v *= f(self.b2)
            v += (f(1.0 - self.b2) * g) * g
            denom = np.sqrt(v) / bc2s + f(self.eps)
            p += (-step_size * m) / denom


def bpr_batch(P, Q, bias, RP, RQ, u, i, j, loss_scale, reg, reg_scale):
    """One BPR batch: loss parts and dense gradients.  Scores from (P, Q, bias), regulariser rows from
    (RP, RQ) -- BPRMF.py:114-124 (same tables) / LightGCN.py:187-196 (propagated vs ego tables).
    Returns (bpr_loss*loss_scale, l2, gP, gQ, gb, gRP, gRQ); gRP/gRQ alias gP/gQ when tables alias."""
    f = np.float32
    pu, qi, qj = P[u], Q[i], Q[j]
    xi = (pu * qi).sum(1, dtype=np.float32)
    xj = (pu * qj).sum(1, dtype=np.float32)
    if bias is not None:
        xi = xi + bias[i]
        xj = xj + bias[j]
    x = (xi - xj).astype(np.float32)
    loss = f(_softplus_neg(x).sum(dtype=np.float64)) * f(loss_scale)
    c = (-_sigmoid_neg(x) * f(loss_scale)).astype(np.float32)
    gP, gQ = np.zeros_like(P), np.zeros_like(Q)
    gb = np.zeros_like(bias) if bias is not None else None
    same = RP is P
    gRP = gP if same else np.zeros_like(RP)
    gRQ = gQ if same else np.zeros_like(RQ)
    np.add.at(gP, u, c[:, None] * (qi - qj))
    np.add.at(gQ, i, c[:, None] * pu)
    np.add.at(gQ, j, -c[:, None] * pu)
    ru, ri, rj = RP[u], RQ[i], RQ[j]
    l2 = 0.5 * ((ru.astype(np.float64) ** 2).sum() + (ri.astype(np.float64) ** 2).sum() + (rj.astype(np.float64) ** 2).sum())
    rs = f(reg * reg_scale)
    np.add.at(gRP, u, rs * ru)
    np.add.at(gRQ, i, rs * ri)
    np.add.at(gRQ, j, rs * rj)
    if bias is not None:
        l2 += 0.5 * ((bias[i].astype(np.float64) ** 2).sum() + (bias[j].astype(np.float64) ** 2).sum())
        np.add.at(gb, i, c + rs * bias[i])
        np.add.at(gb, j, -c + rs * bias[j])
    return loss, f(l2), gP, gQ, gb, gRP, gRQ


def lightgcn_propagate(A, E0, n_layers):
    """_forward_gcn (LightGCN.py:89-100): mean of E0, A E0, ..., A^K E0 (A: scipy CSR fp32)."""
    layers = [E0]
    x = E0
    for _ in range(n_layers):
        x = (A @ x).astype(np.float32)
        layers.append(x)
    return np.stack(layers, axis=1).mean(axis=1, dtype=np.float32)


def lightgcn_step(A, E0, n_users, u, i, j, n_layers, reg, batch_size_cfg):
    """loss + dense gradient wrt E0 of one LightGCN batch (LightGCN.py:180-199)."""
    f = np.float32
    Ebar = lightgcn_propagate(A, E0, n_layers)
    n = len(u)
    loss, l2, gPu, gQi, _, gRu, gRi = bpr_batch(Ebar[:n_users], Ebar[n_users:], None, E0[:n_users], E0[n_users:],
                                                u, i, j, 1.0 / n, reg, 1.0 / batch_size_cfg)
    H = (np.concatenate([gPu, gQi], 0) * f(1.0 / (n_layers + 1))).astype(np.float32)
    G = H
    At = A.T.tocsr()
    for _ in range(n_layers):
        G = (At @ G).astype(np.float32) + H
    return loss, l2, (G + np.concatenate([gRu, gRi], 0)).astype(np.float32)


def _cos_rows(Y, E, eps=1e-8):
    ny = np.maximum(np.sqrt((Y * Y).sum(1, dtype=np.float32)), f32(eps))
    ne = np.maximum(np.sqrt((E * E).sum(1, dtype=np.float32)), f32(eps))
    return ((Y / ny[:, None]) * (E / ne[:, None])).sum(1, dtype=np.float32), ny, ne


f32 = np.float32


def layergcn_forward(A, E0, n_layers):
    """_LayerGCN.forward (LayerGCN.py:207-220): returns (sum of refined layers, [Y_k], [w_k])."""
    X, out, Ys, Ws = E0, np.zeros_like(E0), [], []
    for _ in range(n_layers):
        Y = (A @ X).astype(np.float32)
        w, _, _ = _cos_rows(Y, E0)
        X = (w[:, None] * Y).astype(np.float32)
        out = out + X
        Ys.append(Y)
        Ws.append(w)
    return out.astype(np.float32), Ys, Ws


def layergcn_step(A, E0, n_users, u, i, j, n_layers, reg):
    """loss + dense gradient wrt E0 of one LayerGCN batch (calculate_loss, LayerGCN.py:245-253)."""
    out, Ys, Ws = layergcn_forward(A, E0, n_layers)
    loss, l2, gPu, gQi, _, gRu, gRi = bpr_batch(out[:n_users], out[n_users:], None, E0[:n_users], E0[n_users:],
                                                u, i, j, 1.0, reg, 1.0)
    gO = np.concatenate([gPu, gQi], 0)
    gE = np.concatenate([gRu, gRi], 0).astype(np.float32)
    dZ = gO
    for k in range(n_layers - 1, -1, -1):
        Y, w = Ys[k], Ws[k]
        _, ny, ne = _cos_rows(Y, E0)
        yh, eh = Y / ny[:, None], E0 / ne[:, None]
        dw = (dZ * Y).sum(1, dtype=np.float32)
        dY = w[:, None] * dZ + dw[:, None] * (eh - w[:, None] * yh) / ny[:, None]
        gE = gE + dw[:, None] * (yh - w[:, None] * eh) / ne[:, None]
        back = (A.T @ dY.astype(np.float32)).astype(np.float32)
        if k > 0:
            dZ = gO + back
        else:
            gE = gE + back
    return loss, l2, gE.astype(np.float32)
