"""CPU suite, part 1: the oracle is pinned -- against the golden vectors produced by running the
reference (tests/golden/make_golden.py) and, when it was built in this container, against the
reference's own C++ compiled where it lies (oracle/_ref/libskrec_ref.so)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O
from helpers import tiny_arrays, lists_from_csr

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref was not built (reference not mounted)")


def test_sampler_known_answers_and_iterators(golden):
    g = golden("golden_sampler")
    U, I, rowptr, pos, srt, uary = tiny_arrays(golden)
    assert np.array_equal(rowptr, g["train_rowptr"]) and np.array_equal(pos, g["train_items_fileorder"])
    s = O.Sampler(2020)
    assert np.array_equal(s.randint_choice(1682, 10, exclusion=[1, 2, 3]), g["ka1"])
    assert s.randint_choice(1682, 1, exclusion=[5]) == g["ka2"]
    assert np.array_equal(s.randint_choice(50, 20, replace=False, exclusion=[0, 1, 2, 3]), g["ka3"])
    assert np.array_equal(s.randint_choice(30, 15, p=g["ka4_p"]), g["ka4"])
    ka5 = np.concatenate([np.atleast_1d(s.randint_choice(40, n, exclusion=e))
                          for n, e in zip([3, 5, 2], [[1, 2], [3], [4, 5, 6]])])
    assert np.array_equal(ka5, g["ka5"])
    assert np.array_equal(s.randint_choice(7, 40), g["ka6"])
    # PairwiseIterator(shuffle=False), two epochs: the stream continues across epochs
    assert np.array_equal(uary, g["pw_e1_users"]) and np.array_equal(pos, g["pw_e1_pos"])
    assert np.array_equal(s.sample_epoch(I, rowptr, pos, 1), g["pw_e1_neg"])
    assert np.array_equal(s.sample_epoch(I, rowptr, pos, 1), g["pw_e2_neg"])
    assert list(g["pw_e1_lens"]) == [128] * 5 + [len(pos) - 640] and int(g["pw_len"]) == 6
    n3 = s.sample_epoch(I, rowptr, pos, 3)
    m = len(g["pw3_neg"])  # drop_last=True
    assert m == 700 and int(g["pw3_len"]) == 7 and np.array_equal(n3[:m], g["pw3_neg"])
    n2 = s.sample_epoch(I, rowptr, pos, 2)
    au, ai, al = O.pointwise_layout(uary, pos, n2, 2)
    assert np.array_equal(au, g["pt_users"]) and np.array_equal(ai, g["pt_items"]) and np.array_equal(al, g["pt_labels"])
    np.random.seed(7)
    neg = s.sample_epoch(I, rowptr, pos, 1)
    perm = np.random.permutation(len(pos))
    assert np.array_equal(uary[perm], g["pws_users"]) and np.array_equal(pos[perm], g["pws_pos"])
    assert np.array_equal(neg[perm], g["pws_neg"])


def test_eval_golden_rows_and_reports(golden):
    e = golden("golden_eval")
    for c in range(int(e["n_cases"])):
        tests = lists_from_csr(e[f"c{c}_test_rowptr"], e[f"c{c}_test_items"])
        rows = O.eval_score_matrix(e[f"c{c}_scores"], tests, e[f"c{c}_mids"], int(e[f"c{c}_K"]))
        assert np.array_equal(rows.view(np.uint32), e[f"c{c}_rows"].view(np.uint32)), c
    d = golden("tiny_dataset")
    tr, te = d["train"], d["test"]
    trd, ted = {}, {}
    for u, i, _ in tr:
        trd.setdefault(int(u), []).append(int(i))
    for u, i, _ in te:
        ted.setdefault(int(u), []).append(int(i))
    trd = {u: np.int32(v) for u, v in sorted(trd.items())}
    ted = {u: np.int32(v) for u, v in sorted(ted.items())}
    table = e["e2e_table"]
    predict = lambda users: table[np.asarray(users)]  # noqa: E731
    for tag, metric, top_k, bs in (("a", None, (5, 10, 20), 16), ("b", ["Recall", "NDCG"], 7, 64), ("c", "MRR", [3], 5)):
        names, vals, _ = O.ranking_evaluate(predict, trd, ted, metric, top_k, bs)
        assert names == list(e[f"e2e_{tag}_names"])
        assert np.array_equal(vals.view(np.uint32), e[f"e2e_{tag}_values"].view(np.uint32))
        _, vals2, _ = O.ranking_evaluate(predict, trd, ted, metric, top_k, bs, test_users=list(e["e2e_sub_users"]))
        assert np.array_equal(vals2.view(np.uint32), e[f"e2e_{tag}_sub_values"].view(np.uint32))


@needs_ref
def test_sampler_vs_compiled_reference():
    rng = np.random.default_rng(0)
    O.ref().ref_reseed(2020)
    s = O.Sampler(2020)
    for t in range(150):
        high = int(rng.integers(2, 3000))
        ex = rng.choice(high, int(rng.integers(0, high // 2 + 1)), replace=False).astype(np.int32) \
            if rng.random() < 0.8 else None
        nex = 0 if ex is None else len(ex)
        rep = bool(rng.random() < 0.6)
        size = int(rng.integers(1, max(2, (high - nex) // 2)))
        if not rep and high - nex <= size:
            continue
        p = rng.random(high).astype(np.float32) if rng.random() < 0.3 else None
        assert np.array_equal(O.ref_randint_choice(high, size, rep, p, ex),
                              np.atleast_1d(s.randint_choice(high, size, rep, p, ex))), t
    U, I = 400, 250
    lens = rng.integers(0, 40, U)
    rowptr = np.zeros(U + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    pos = np.concatenate([rng.choice(I, l, replace=False) for l in lens]).astype(np.int32)
    for nn in (1, 3):
        for _ in range(2):
            assert np.array_equal(O.ref_sample_epoch(I, rowptr, pos, nn), s.sample_epoch(I, rowptr, pos, nn))


@needs_ref
def test_eval_vs_compiled_reference_including_ties():
    rng = np.random.default_rng(1)
    for t in range(80):
        B, I = int(rng.integers(1, 6)), int(rng.integers(3, 400))
        K = min(int(rng.integers(1, I + 1)), 60)
        sc = rng.standard_normal((B, I)).astype(np.float32)
        if t % 3 == 0:
            sc = np.round(sc * 2) / 2  # many exact ties: the heap order of partial_sort_copy matters
        if t % 5 == 0:
            sc[:, rng.integers(0, I, I // 3)] = -np.inf
        tests = [rng.choice(I, int(rng.integers(0, min(I, 8) + 1)), replace=False).astype(np.int32) for _ in range(B)]
        r = O.ref_eval_score_matrix(sc, tests, [1, 2, 3, 4, 5], K, thread_num=int(rng.integers(1, 4)))
        o = O.eval_score_matrix(sc, tests, [1, 2, 3, 4, 5], K)
        assert np.array_equal(r.view(np.uint32), o.view(np.uint32)), t


def test_tie_rules_agree_on_tie_free_rows():
    rng = np.random.default_rng(2)
    row = rng.permutation(500).astype(np.float32)
    assert np.array_equal(O.topk_ids_heap(row, 20), O.topk_ids_lowid(row, 20))
    assert np.array_equal(O.topk_ids_lowid(row, 20), np.argsort(-row, kind="stable")[:20])


def _golden_batches(golden, epochs, bs):
    U, I, rowptr, pos, srt, uary = tiny_arrays(golden)
    s = O.Sampler(2020)
    np.random.seed(2021)
    for _ in range(epochs):
        neg = s.sample_epoch(I, rowptr, pos, 1)
        perm = np.random.permutation(len(pos))
        for st in range(0, len(pos), bs):
            idx = perm[st:st + bs]
            yield uary[idx], pos[idx], neg[idx]


def test_bprmf_trajectory_matches_reference(golden):
    g = golden("golden_bprmf")
    Um, V, b = g["U0"].copy(), g["V0"].copy(), g["b0"].reshape(-1).copy()
    opt = O.Adam([Um, V, b], 1e-3)
    for k, (u, i, j) in enumerate(_golden_batches(golden, 3, 256)):
        loss, l2, gU, gV, gb, _, _ = O.bpr_batch(Um, V, b, Um, V, u, i, j, 1.0, 1e-3, 1.0)
        assert abs(loss - g["bpr_sum"][k]) <= 1e-5 * abs(g["bpr_sum"][k])
        assert abs(l2 - g["l2"][k]) <= 1e-5 * abs(g["l2"][k])
        opt.step([gU, gV, gb])
    assert k + 1 == len(g["bpr_sum"])
    np.testing.assert_allclose(Um, g["U1"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(V, g["V1"], rtol=0, atol=2e-6)


def _adj(g, n):
    return sp.csr_matrix((g["adj_val"], (g["adj_idx"][0], g["adj_idx"][1])), shape=(n, n)).astype(np.float32)


def test_lightgcn_trajectory_matches_reference(golden):
    g = golden("golden_lightgcn")
    U, I = g["U0"].shape[0], g["V0"].shape[0]
    A = _adj(g, U + I)
    E0 = np.concatenate([g["U0"], g["V0"]], 0).copy()
    opt = O.Adam([E0], 1e-3)
    for k, (u, i, j) in enumerate(_golden_batches(golden, 2, 256)):
        loss, l2, gE = O.lightgcn_step(A, E0, U, u, i, j, 3, 1e-3, 256)
        assert abs(loss - g["bpr_mean"][k]) <= 1e-5 * abs(g["bpr_mean"][k])
        assert abs(l2 - g["l2"][k]) <= 1e-5 * abs(g["l2"][k])
        opt.step([gE])
    np.testing.assert_allclose(E0[:U], g["U1"], rtol=0, atol=2e-6)
    Ebar = O.lightgcn_propagate(A, E0, 3)
    np.testing.assert_allclose(Ebar[:U], g["Uf"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(Ebar[U:], g["Vf"], rtol=0, atol=2e-6)


def test_layergcn_trajectory_matches_reference(golden):
    g = golden("golden_layergcn")
    U, I = g["U0"].shape[0], g["V0"].shape[0]
    A = _adj(g, U + I)
    E0 = np.concatenate([g["U0"], g["V0"]], 0).copy()
    opt = O.Adam([E0], 1e-3)
    for k, (u, i, j) in enumerate(_golden_batches(golden, 2, 256)):
        loss, l2, gE = O.layergcn_step(A, E0, U, u, i, j, 4, 1e-2)
        tot = loss + np.float32(1e-2) * l2
        assert abs(tot - g["loss"][k]) <= 1e-5 * abs(g["loss"][k])
        opt.step([gE])
    np.testing.assert_allclose(E0[:U], g["U1"], rtol=0, atol=2e-6)
    out, _, _ = O.layergcn_forward(A, E0, 4)
    np.testing.assert_allclose(out[:U], g["Uf"], rtol=0, atol=5e-6)


def test_oracle_stream_reproduces_sequential_and_kg_iterators(golden):
    """The restated sampler, driven the way the reference's loops drive it (one randint_choice per user / head
    with that group's draw count and exclusion set), reproduces the negatives of the reference's
    sequential and knowledge-graph iterators (golden_iterators.npz) -- pins the oracle for SURVEY 8f-3."""
    from oracle import oracle as O
    g = golden("golden_iterators")
    tiny = golden("tiny_dataset")
    tr = tiny["train"]
    order = np.lexsort((np.arange(len(tr)), tr[:, 2], tr[:, 0]))      # columns (user, item, time): by user, then time
    users, items = tr[order, 0].astype(np.int64), tr[order, 1].astype(np.int32)
    hist = {int(u): items[users == u] for u in np.unique(users)}
    n_items = int(tiny["num_items"])
    s = O.Sampler(2020)

    def epoch(counts, excl, high, k):
        out = []
        for key, n in counts:
            r = np.atleast_1d(s.randint_choice(high, n * k, exclusion=excl[key]))
            out.append(r.reshape(n, k) if k > 1 else r)
        return np.concatenate(out)

    def counts(stop):
        return [(u, len(h) - stop) for u, h in hist.items() if len(h) > stop]

    # the constructions of make_golden.make_iterators, in order
    assert np.array_equal(epoch(counts(3), hist, n_items, 1), g["spw_e0_c3"])
    assert np.array_equal(epoch(counts(3), hist, n_items, 1), g["spw_e1_c3"])
    assert np.array_equal(epoch(counts(2), hist, n_items, 2)[:600], g["spwpad_e0_c3"])
    assert np.array_equal(epoch(counts(1), hist, n_items, 1), g["spw11_e0_c3"])
    neg = epoch(counts(2), hist, n_items, 2)                             # spt: num_neg=2, num_next=1
    n = len(neg)
    assert np.array_equal(np.concatenate([neg[:, 0], neg[:, 1]]), g["spt_e0_c2"][n:])
    neg = epoch(counts(2), hist, n_items, 4)                             # sptpad: num_neg=2, num_next=2
    assert np.array_equal(np.concatenate([neg[:, :2], neg[:, 2:]], 0), g["sptpad_e0_c2"][len(neg):])
    tri = g["kg_triplets"]
    heads = np.unique(tri[:, 0])
    tails = {int(h): tri[tri[:, 0] == h][:, 2] for h in heads}
    kc = [(int(h), len(tails[int(h)])) for h in heads]
    assert np.array_equal(epoch(kc, tails, 60, 1), g["kg_e0_c3"])
    assert np.array_equal(epoch(kc, tails, 60, 3), g["kg3_e0_c3"])


def test_gru4rec_oracle_losses_against_a_float64_restatement():
    """oracle/gru4rec.py's bpr_max / top1_max (torch fp32) against the same formulas of GRU4RecPlus.py:137-166
    written out element by element in float64 numpy.  (PARITY UNPINNED against the reference itself: TensorFlow
    is absent and the reference holds no recorded output for this model.)"""
    import torch
    from oracle import gru4rec as G
    rng = np.random.default_rng(0)
    b, n = 4, 9
    lg = rng.normal(0, 1.5, (b, n))
    hm = 1.0 - np.eye(b, n)
    masked = lg * hm
    e = np.exp(masked - masked.max(1, keepdims=True)) * hm
    s = e / e.sum(1, keepdims=True)
    pos = np.diag(lg)[:, None]
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))  # noqa: E731
    bpr = (-np.log((sig(pos - lg) * s).sum(1) + 1e-24) + 0.6 * ((lg ** 2) * s).sum(1)).mean()
    top1 = (((sig(-pos + lg) + sig(lg ** 2)) * s).sum(1)).mean()
    t = torch.tensor(lg, dtype=torch.float32)
    assert abs(float(G.bpr_max_loss(t, 0.6)) - bpr) < 1e-5 * abs(bpr)
    assert abs(float(G.top1_max_loss(t)) - top1) < 1e-5 * abs(top1)
    np.testing.assert_allclose(G.softmax_neg(t).numpy(), s, rtol=1e-5, atol=1e-7)
    # GRUCell restatement: gate bias 1 and zero kernels give r = u = sigmoid(1), c = act(0) = 0 -> h' = u h
    h = torch.tensor(rng.normal(0, 1, (3, 8)), dtype=torch.float32)
    x = torch.zeros(3, 5)
    out = G.gru_cell(x, h, torch.zeros(13, 16), torch.ones(16), torch.zeros(13, 8), torch.zeros(8), "tanh")
    np.testing.assert_allclose(out.numpy(), sig(1.0) * h.numpy(), rtol=1e-6)
