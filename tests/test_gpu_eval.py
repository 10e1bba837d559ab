"""GPU parity, E rows: skr_eval_scores (drop-in for cpp_evaluate_matrix), skr_rank_metrics,
skr_eval_fused_topk and the RankingEvaluator host mirror, against the oracle and the golden vectors."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from helpers import lists_from_csr, random_csr

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
ALL = [1, 2, 3, 4, 5]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_scores_golden_cases(golden):
    """the reference's own eval_score_matrix outputs, bit for bit (tie-free rows; -inf masked columns)"""
    from gpu_utils import eval_scores
    e = golden("golden_eval")
    for c in range(int(e["n_cases"])):
        tests = lists_from_csr(e[f"c{c}_test_rowptr"], e[f"c{c}_test_items"])
        K = int(e[f"c{c}_K"])
        rows, ids, sums = eval_scores(e[f"c{c}_scores"], tests, e[f"c{c}_mids"], K)
        assert np.array_equal(rows.view(np.uint32), e[f"c{c}_rows"].view(np.uint32)), c
        np.testing.assert_allclose(sums, e[f"c{c}_rows"].astype(np.float64).sum(0), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("B,I,K", [(3, 5, 5), (4, 1000, 1), (5, 1024, 10), (2, 4097, 128), (7, 20000, 50),
                                    (64, 100000, 10), (1, 2048, 100), (3, 5000, 300), (2, 100000, 512), (2, 600, 512), (3, 1500, 129)])
def test_eval_scores_vs_oracle(B, I, K):
    from gpu_utils import eval_scores
    rng = np.random.default_rng(B * 1000 + I + K)
    sc = np.stack([rng.permutation(I).astype(np.float32) for _ in range(B)]) * np.float32(0.37) - np.float32(11)
    for b in range(B):  # masked (train) columns
        sc[b, rng.choice(I, min(I - K, int(rng.integers(0, 60))), replace=False)] = -np.inf
    tests = [rng.choice(I, int(rng.integers(0, min(I, 12) + 1)), replace=False) for _ in range(B)]
    want, want_ids = O.eval_score_matrix(sc, tests, ALL, K, return_ids=True)
    rows, ids, _ = eval_scores(sc, tests, ALL, K)
    assert np.array_equal(ids, want_ids)
    assert np.array_equal(rows.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("K,I", [(25, 3000), (1, 77), (10, 15), (128, 5000), (50, 100), (64, 64), (400, 5000), (512, 700)])
def test_eval_scores_ties_follow_the_reference_heap_order(K, I):
    """equal scores: the ids AND the metric rows are those of the reference's partial_sort_copy
    (libstdc++ heap order, evaluate.h:39-45) -- quantised scores, all-equal rows, ascending and descending
    rows (every element enters / none enters the heap), -inf masked prefixes, signed zeros"""
    from gpu_utils import eval_scores
    rng = np.random.default_rng(3 + K)
    sc = np.round(rng.standard_normal((12, I)).astype(np.float32) * 4) / 4
    sc[0, :] = 1.0                                        # an all-equal row
    sc[1, :] = np.arange(I, dtype=np.float32) // 3        # ascending with runs of three
    sc[2, :] = -(np.arange(I, dtype=np.float32) // 5)     # descending with runs of five
    sc[3, :] = 0.0
    sc[3, ::2] = -0.0                                     # signed zeros compare equal
    sc[4, : I // 2] = -np.inf                             # masked prefix
    sc[5, :] = rng.integers(0, 3, I)                      # three distinct values only
    sc[6, :] = rng.standard_normal(I).astype(np.float32)  # tie-free control row
    truth = [rng.choice(I, size=min(I, 7), replace=False) for _ in range(12)]
    rows, ids, _ = eval_scores(sc, truth, [1, 2, 3, 4, 5], K)
    for b in range(12):
        assert np.array_equal(ids[b], O.topk_ids_heap(sc[b], K)), b
    assert np.array_equal(rows, O.eval_score_matrix(sc, truth, [1, 2, 3, 4, 5], K))


def test_eval_scores_argument_errors():
    from gpu_utils import eval_scores
    rng = np.random.default_rng(3)
    sc = rng.standard_normal((6, 3000)).astype(np.float32)
    with pytest.raises(ValueError):
        eval_scores(sc, [[] for _ in range(6)], [6], 5)        # unknown metric id
    with pytest.raises(ValueError):
        eval_scores(sc[:, :4], [[] for _ in range(6)], [1], 5)  # top_k > n_items
    with pytest.raises(ValueError):
        eval_scores(sc, [[] for _ in range(6)], [1], 513)         # beyond SKR_MAX_TOPK_SCORES


def test_rank_metrics_vs_oracle():
    import torch
    from gpu_utils import to_dev, dev
    from skrec import _hip
    rng = np.random.default_rng(8)
    B, I, K = 500, 300, 37
    ids = np.stack([rng.permutation(I)[:K] for _ in range(B)]).astype(np.int32)
    rowptr, items = random_csr(rng, B, I, 0, 50)
    # reference semantics through the oracle: a score row whose descending order is `ids`
    sc = np.full((B, I), -1e9, np.float32)
    for b in range(B):
        sc[b, ids[b]] = np.arange(K, 0, -1, dtype=np.float32)
    want = O.eval_score_matrix(sc, lists_from_csr(rowptr, items), ALL, K)
    rows = torch.zeros((B, 5 * K), dtype=torch.float32, device=dev())
    sums = torch.zeros(5 * K, dtype=torch.float64, device=dev())
    d_ids, d_ptr, d_items = to_dev(ids), to_dev(rowptr), to_dev(items)
    _hip.check(_hip.lib().skr_rank_metrics(_hip.ptr(d_ids), B, K, None, _hip.ptr(d_ptr), _hip.ptr(d_items),
                                           _hip.metric_array(ALL), 5, _hip.ptr(rows), _hip.ptr(sums), _hip.stream()))
    torch.cuda.synchronize()
    assert np.array_equal(rows.cpu().numpy().view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(sums.cpu().numpy(), want.astype(np.float64).sum(0), rtol=1e-12)


def _fused_reference(Ut, users, It, bias, rowptr, items, K):
    """fp64 scores, ranks 1..K+1 and the gaps between consecutive ranks.  fp32 summation order may
    swap two items whose fp64 scores are closer than the fp32 dot-product noise, so ids are compared
    only where the neighbouring gaps are clear of it (top-K ties)."""
    sc = Ut[users].astype(np.float64) @ It.astype(np.float64).T
    if bias is not None:
        sc = sc + bias.astype(np.float64)
    for r, u in enumerate(users):
        sc[r, items[rowptr[u]:rowptr[u + 1]]] = -np.inf
    order = np.argsort(-sc, axis=1, kind="stable")[:, :K + 1]
    top = np.take_along_axis(sc, order, 1)
    gaps = -np.diff(top, axis=1)                    # gaps[:, p] = score(rank p) - score(rank p+1)
    return order[:, :K], top[:, :K], gaps


NOISE = 2e-5  # >> 64 * 2^-24 * sum|a*b| for the factor scales used here


@pytest.mark.parametrize("B,I,K,with_bias,with_mask", [(1, 40, 5, True, True), (64, 32, 10, False, True),
                                                       (65, 1000, 20, True, True), (200, 5000, 50, True, False),
                                                       (130, 33, 3, False, True), (300, 20011, 128, True, True),
                                                       (1000, 3000, 10, False, True),
                                                       # catalogues that end inside the first 16-item group, on its edge, one item
                                                       # into the second, and on a tile's first group (fused_topk_kernel_v6's steps)
                                                       (2, 5, 4, True, True), (3, 16, 4, False, False), (70, 17, 9, True, True),
                                                       (64, 48, 10, True, False), (5, 49, 7, False, True)])
def test_fused_topk_vs_fp64(B, I, K, with_bias, with_mask, fused_mode):
    from gpu_utils import fused_topk
    rng = np.random.default_rng(B + I + K)
    nU = B + 17
    Ut = (rng.standard_normal((nU, 64)) * 0.3).astype(np.float32)
    It = (rng.standard_normal((I, 64)) * 0.3).astype(np.float32)
    bias = (rng.standard_normal(I) * 0.1).astype(np.float32) if with_bias else None
    users = rng.permutation(nU)[:B].astype(np.int32)
    max_tr = max(0, min(I - K, 120))
    rowptr, items = random_csr(rng, nU, I, 0, max_tr) if with_mask else (None, np.zeros(0, np.int32))
    ids, sc = fused_topk(Ut, users, It, bias, rowptr, items, K)
    rp = rowptr if with_mask else np.zeros(nU + 1, np.int64)
    want_ids, want_sc, gaps = _fused_reference(Ut, users, It, bias, rp, items, K)
    clear = gaps > NOISE
    # rank p is comparable id-for-id when the gaps above and below it are clear
    pos_ok = clear[:, 1:] if K == 1 else np.concatenate([clear[:, :1], clear[:, :-1] & clear[:, 1:]], axis=1)[:, :K]
    pos_ok[:, 0] = clear[:, 0]
    assert pos_ok.mean() > 0.6
    assert np.array_equal(ids[pos_ok], want_ids[pos_ok])
    set_ok = clear[:, K - 1]                        # the K / K+1 boundary decides the id SET
    assert set_ok.mean() > 0.6
    for r in np.flatnonzero(set_ok):
        assert set(ids[r]) == set(want_ids[r]), r
    np.testing.assert_allclose(np.sort(sc, axis=1), np.sort(want_sc, axis=1), rtol=2e-5, atol=2e-5)
    # every row: a valid ranking of distinct unmasked items, scores descending
    for r in range(B):
        assert len(set(ids[r])) == K and ids[r].min() >= 0 and ids[r].max() < I
        assert np.all(np.diff(sc[r]) <= 0)
        if with_mask:
            u = users[r]
            assert not np.isin(ids[r], items[rp[u]:rp[u + 1]]).any()


@pytest.mark.parametrize("B,I,K", [(96, 777, 12), (70, 6000, 100), (130, 3000, 64), (65, 9000, 128)])
def test_fused_exact_on_integer_scores(B, I, K, fused_mode):
    """integer-valued factors make every fp32 dot product exact, so ids AND scores must match the
    oracle's ranking bit for bit, ties included (lower id first) -- with thousands of items and a few
    dozen distinct scores the K-th place is almost always inside a run of equal scores, which is what the
    id-word phase of the mid-sweep selection and the final sort have to get right"""
    from gpu_utils import fused_topk
    rng = np.random.default_rng(4 + K)
    Ut = rng.integers(-3, 4, (B, 64)).astype(np.float32)
    It = rng.integers(-3, 4, (I, 64)).astype(np.float32)
    bias = rng.integers(-5, 6, I).astype(np.float32)
    rowptr, items = random_csr(rng, B, I, 0, 100)
    ids, sc = fused_topk(Ut, np.arange(B, dtype=np.int32), It, bias, rowptr, items, K)
    full = Ut @ It.T + bias
    for r in range(B):
        full[r, items[rowptr[r]:rowptr[r + 1]]] = -np.inf
        want = O.topk_ids_lowid(full[r], K)
        assert np.array_equal(ids[r], want)
        assert np.array_equal(sc[r], full[r, want])


def test_ranking_evaluator_end_to_end(golden):
    """RankingEvaluator (generic predict() contract) == the reference's MetricReport values"""
    from skrec.utils.py import RankingEvaluator
    e = golden("golden_eval")
    d = golden("tiny_dataset")
    trd, ted = {}, {}
    for u, i, _ in d["train"]:
        trd.setdefault(int(u), []).append(int(i))
    for u, i, _ in d["test"]:
        ted.setdefault(int(u), []).append(int(i))
    trd = {u: np.int32(v) for u, v in sorted(trd.items())}
    ted = {u: np.int32(v) for u, v in sorted(ted.items())}
    table = e["e2e_table"]

    class Fixed:
        def predict(self, users):
            return table[np.asarray(users)].copy()
    for tag, metric, top_k, bs in (("a", None, (5, 10, 20), 16), ("b", ["Recall", "NDCG"], 7, 64), ("c", "MRR", [3], 5)):
        ev = RankingEvaluator(trd, ted, metric=metric, top_k=top_k, batch_size=bs, num_thread=2)
        rep = ev.evaluate(Fixed())
        assert list(rep.metrics()) == list(e[f"e2e_{tag}_names"])
        got = np.array(list(rep.values()), np.float32)
        assert np.array_equal(got.view(np.uint32), e[f"e2e_{tag}_values"].view(np.uint32)), tag
        rep2 = ev.evaluate(Fixed(), test_users=list(e["e2e_sub_users"]))
        got2 = np.array(list(rep2.values()), np.float32)
        assert np.array_equal(got2.view(np.uint32), e[f"e2e_{tag}_sub_values"].view(np.uint32))
    with pytest.raises(AssertionError):
        ev.evaluate(object())


def test_fused_evaluator_reranks_structural_ties_in_reference_order(fused_mode):
    """RankingEvaluator's fused path on factors with structural ties -- cold users (all-zero rows) and
    duplicated items -- returns the reference's MetricReport: tied rows are re-ranked from their dense score
    row in libstdc++'s heap order (oracle = the reference's evaluator restated, pinned on its own build)"""
    import torch
    from skrec.utils.py import RankingEvaluator
    rng = np.random.default_rng(8)
    nU, nI = 150, 400
    Ut = rng.integers(-2, 3, (nU, 64)).astype(np.float32)      # integer factors: exact scores on every path
    It = rng.integers(-2, 3, (nI, 64)).astype(np.float32)
    Ut[::7] = 0.0                                              # cold users
    It[10:20] = It[30:40]                                      # duplicated items
    train = {u: np.sort(rng.choice(nI, rng.integers(1, 30), replace=False)) for u in range(nU) if u % 11}
    test = {u: rng.choice(nI, rng.integers(1, 6), replace=False) for u in range(nU) if u % 5}
    dU, dI = torch.from_numpy(Ut).cuda(), torch.from_numpy(It).cuda()

    class M(object):
        def predict_factors(self):
            return dU, dI, None

        def predict(self, users):
            return Ut[np.asarray(users)] @ It.T

    ev = RankingEvaluator(train, test, metric=["Precision", "Recall", "MAP", "NDCG", "MRR"], top_k=[5, 10, 20], batch_size=32)
    got = ev.evaluate(M())
    _, want, _ = O.ranking_evaluate(M().predict, train, test, metric=["Precision", "Recall", "MAP", "NDCG", "MRR"],
                                    top_k=[5, 10, 20], batch_size=32)
    np.testing.assert_allclose(np.array(list(got.values()), np.float32), want, rtol=1e-6, atol=0)   # fp32 mean order


def test_evaluator_ranks_deeper_than_the_fused_kernel_through_the_score_matrix():
    """top_k is a free integer in the reference (run_config.py:16).  Lists deeper than the fused kernel's 128, and factor
    widths other than 64, go through the score-matrix path (predict() -> skr_mask_train -> skr_eval_scores): same report
    as the oracle's evaluator loop"""
    import torch
    from skrec import _hip
    from skrec.utils.py import RankingEvaluator
    rng = np.random.default_rng(5)
    nU, nI = 90, 700
    train = {u: np.sort(rng.choice(nI, rng.integers(1, 40), replace=False)) for u in range(nU)}
    test = {u: rng.choice(nI, rng.integers(1, 9), replace=False) for u in range(nU) if u % 3}
    for width, top_k in ((64, (10, 200)), (128, (5, 50)), (128, 300)):
        Ut = rng.integers(-3, 4, (nU, width)).astype(np.float32)
        It = rng.integers(-3, 4, (nI, width)).astype(np.float32) + rng.permutation(nI).astype(np.float32)[:, None] / 1024   # no ties
        dU, dI = torch.from_numpy(Ut).cuda(), torch.from_numpy(It).cuda()

        class M(object):
            def predict_factors(self):
                return dU, dI, None

            def predict(self, users):       # the models' predict(): skr_score_matrix at the factors' width
                return _hip.score_matrix(dU, users, dI, None).cpu().numpy()
        ev = RankingEvaluator(train, test, metric=["Precision", "Recall", "MAP", "NDCG", "MRR"], top_k=top_k, batch_size=32)
        got = ev.evaluate(M())
        ref = lambda us: (Ut[np.asarray(us)].astype(np.float64) @ It.T.astype(np.float64)).astype(np.float32)   # noqa: E731
        _, want, _ = O.ranking_evaluate(ref, train, test, metric=["Precision", "Recall", "MAP", "NDCG", "MRR"], top_k=top_k,
                                        batch_size=32)
        np.testing.assert_allclose(np.array(list(got.values()), np.float32), want, rtol=1e-6, atol=0)


def test_fused_evaluator_sees_a_tie_at_the_k_boundary(fused_mode):
    """a tie that exists ONLY between the K-th and the (K+1)-th score (evaluate.h:39-45 keeps the one the heap order
    prefers, not the lower id): the fused path asks for K + 1 entries, sees the tie and re-ranks the user from the dense
    row -- per-user metric rows equal the oracle's for every user, bit for bit"""
    import torch
    from skrec.utils.py import RankingEvaluator
    rng = np.random.default_rng(12)
    nU, nI, K = 96, 300, 10
    # scores are exact integers on every path: one-hot user rows select a column of small-integer item factors
    It = np.zeros((nI, 64), np.float32)
    Ut = np.zeros((nU, 64), np.float32)
    for u in range(nU):
        Ut[u, u % 64] = 1.0
    for d in range(64):
        col = rng.permutation(nI).astype(np.float32)             # all distinct: no tie anywhere ...
        srt = np.argsort(-col)
        if d % 2 == 0:
            col[srt[K]] = col[srt[K - 1]]                         # ... except exactly at the K / K+1 boundary
        It[:, d] = col
    train = {u: np.sort(rng.choice(nI, 3, replace=False)) for u in range(nU)}
    test = {u: rng.choice(nI, 4, replace=False) for u in range(nU)}
    dU, dI = torch.from_numpy(Ut).cuda(), torch.from_numpy(It).cuda()

    class M(object):
        def predict_factors(self):
            return dU, dI, None

        def predict(self, users):
            return Ut[np.asarray(users)] @ It.T
    ev = RankingEvaluator(train, test, metric=["Precision", "Recall", "MAP", "NDCG", "MRR"], top_k=K, batch_size=32)
    rows, _, _ = ev.per_user_rows(M(), list(test.keys()))
    _, _, want_rows = O.ranking_evaluate(M().predict, train, test, metric=["Precision", "Recall", "MAP", "NDCG", "MRR"], top_k=K,
                                         batch_size=32)
    assert np.array_equal(rows.view(np.uint32), want_rows.view(np.uint32))


def test_eval_scores_is_deterministic_on_long_rows():
    """Regression: the decision to compact the LDS candidate buffer used to be taken from a counter that other
    threads were already incrementing, so threads could disagree and a candidate was lost in about one of 1e5
    rows at top-100 (different rows on different runs).  Same input, repeated launches -> identical lists, and
    the lists are the oracle's."""
    import torch
    from skrec import _hip
    L = _hip.lib()
    rng = np.random.default_rng(21)
    B, I, K = 3072, 100_000, 100
    sc = torch.from_numpy((rng.standard_normal((B, I)) * 0.1).astype(np.float32)).cuda()
    outs = []
    for _ in range(5):
        ids = torch.empty((B, K), dtype=torch.int32, device="cuda")
        _hip.check(L.skr_eval_scores(_hip.ptr(sc), B, I, I, None, None, None, 0, K, None, _hip.ptr(ids), None, _hip.stream()))
        torch.cuda.synchronize()
        outs.append(ids)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    host = sc[:64].cpu().numpy()
    for r in range(64):
        assert np.array_equal(outs[0][r].cpu().numpy(), O.topk_ids_heap(host[r], K))


def _rejected():
    import ctypes
    from skrec import _hip
    n = ctypes.c_int32(-1)
    _hip.check(_hip.lib().skr_eval_fused_rejected(ctypes.byref(n), _hip.stream()))
    return n.value


def test_f16x2_guard_accepts_tables_of_one_magnitude_and_is_as_accurate_as_the_fp32_chain(monkeypatch):
    """SKR_FUSED_MODE=f16x2 (fused_topk_kernel_v7): on factors of one magnitude -- any common scale, it is divided out by a
    power of two -- the guard accepts every user, and the scores are as close to float64 as the FP32-MFMA kernel's"""
    from gpu_utils import fused_topk
    rng = np.random.default_rng(7)
    B, I, K = 192, 6000, 20
    for scale in (1e-3, 0.2, 30.0):
        U = (rng.standard_normal((B, 64)) * scale).astype(np.float32)
        V = (rng.standard_normal((I, 64)) * scale).astype(np.float32)
        b = (rng.standard_normal(I) * scale * scale).astype(np.float32)
        err = {}
        for mode in ("f16x2", "fp32"):
            monkeypatch.setenv("SKR_FUSED_MODE", mode)
            ids, sc = fused_topk(U, np.arange(B, dtype=np.int32), V, b, None, np.zeros(0, np.int32), K)
            if mode == "f16x2":
                assert _rejected() == 0
            exact = np.einsum("bkd,bd->bk", V.astype(np.float64)[ids], U.astype(np.float64)) + b.astype(np.float64)[ids]
            denom = np.einsum("bkd,bd->bk", np.abs(V.astype(np.float64))[ids], np.abs(U.astype(np.float64))) + np.abs(b.astype(np.float64))[ids]
            err[mode] = (np.abs(sc - exact) / denom).max()
        assert err["f16x2"] < 4e-7 and err["f16x2"] < 2.0 * err["fp32"], err


def test_f16x2_guard_hands_rows_it_cannot_vouch_for_to_bf16x3(monkeypatch):
    """users (and items) far smaller than their table's largest element lose the low fp16 piece to denormals: the guard must
    catch every such user -- its smallest returned score lies under the floor -- and the bf16x3 kernel recompute it inside the
    same call; the result is then as good as bf16x3's for EVERY user.  Also: a table holding an inf is rejected whole."""
    from gpu_utils import fused_topk
    monkeypatch.setenv("SKR_FUSED_MODE", "f16x2")
    rng = np.random.default_rng(11)
    B, I, K = 160, 3000, 10
    U = (rng.standard_normal((B, 64)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((I, 64)) * 0.3).astype(np.float32)
    small = rng.permutation(B)[:37]
    U[small] *= np.float32(2.0 ** -22)                    # rows 4 million times smaller than the table's largest
    V[rng.permutation(I)[:500]] *= np.float32(2.0 ** -20)
    ids, sc = fused_topk(U, np.arange(B, dtype=np.int32), V, None, None, np.zeros(0, np.int32), K)
    n_rej = _rejected()
    assert 37 <= n_rej < B
    exact = np.einsum("bkd,bd->bk", V.astype(np.float64)[ids], U.astype(np.float64))
    denom = np.einsum("bkd,bd->bk", np.abs(V.astype(np.float64))[ids], np.abs(U.astype(np.float64)))
    assert (np.abs(sc - exact) / denom).max() < 4e-7      # the small users included: relative to THEIR scores
    full = U.astype(np.float64) @ V.astype(np.float64).T
    order = np.argsort(-full, axis=1, kind="stable")
    want = order[:, :K]
    gap = -np.diff(np.take_along_axis(full, order[:, :K + 1], axis=1), axis=1)
    clear = (gap > 1e-5 * np.abs(np.take_along_axis(full, order[:, :1], axis=1))).all(axis=1)
    assert clear.mean() > 0.5
    assert np.array_equal(ids[clear], want[clear])
    # an inf in the item table: nothing can be vouched for, every user goes to the bf16x3 kernel (non-finite scores are outside
    # what either kernel promises -- skr_common.h: rank_key -- the point is that f16x2 does not pretend)
    V2 = V.copy()
    V2[5, 3] = np.inf
    fused_topk(np.abs(U), np.arange(B, dtype=np.int32), V2, None, None, np.zeros(0, np.int32), K)
    assert _rejected() == B


def test_fused_default_arithmetic_on_random_cases():
    """tools/f16x2_stress.py, 30 cases: random shapes, top_k 1..128, scales over seven decades, rows and elements of mixed magnitude,
    bias, train masks -- in the evaluator's DEFAULT arithmetic (SKR_FUSED_MODE unset): scores within 1e-6 of float64's relative to
    sum |u v| + |bias|, no masked item returned, id lists equal to float64's wherever every rank gap is clear of the noise"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("f16x2_stress", os.path.join(REPO, "tools", "f16x2_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n_rows, n_id_rows, worst, n_rej = mod.run(30, seed=7, verbose=False)
    assert n_id_rows > 0.6 * n_rows and worst < 1e-6
    assert 0 < n_rej < n_rows          # both the guarded kernel and its fall-back were exercised
