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
