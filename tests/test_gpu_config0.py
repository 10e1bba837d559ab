"""GPU parity at the shape of BASELINE configs[0]: BPRMF d=64 on an ml-100k-SHAPED set (943 users x 1 349 items,
~99 k ratings, ratio split by time 0.7 / 0.3 -- the real file is not in the container: no network), batch 1024,
PairwiseSampler + RankingEvaluator, driven through the reference's API (RunConfig -> BPRMF -> fit()) and compared with
the ORACLE's replay of the same run: exact-stream negatives (oracle/skr_oracle.c, pinned to randint.h), numpy's epoch
permutations, explicit-gradient BPR + dense Adam in numpy (oracle.bpr_batch / oracle.Adam, pinned by the reference's
recorded fit() trajectories in tests/golden), and the reference's evaluator loop (oracle.ranking_evaluate on the
C restatement of evaluate.h / metric.h).

Epochs here have 69 steps, so -- unlike the 64-user golden set -- they contain whole 32-step blocks of the temporally
blocked Adam (the shipped default), the block that does not divide the epoch, and hot rows caught up across blocks.
Reference: skrec/recommender/BPRMF.py:99-127, io/data_iterator.py:81-94,226-234, utils/py/evaluator.py:163-214."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
SEED = 2021
N_USERS, N_ITEMS = 943, 1349


def _ml100k_shaped(rng):
    """(train rows, test rows) of (user, item, time): log-normal user activity >= 20, Zipf item popularity, per-user
    items without replacement, first 70 % of a user's ratings by time -> train (tutorial.ipynb:243-252)"""
    act = np.exp(rng.standard_normal(N_USERS) * 0.9)
    act = np.clip(np.round(act / act.sum() * 99_287), 20, 700).astype(np.int64)
    pop = 1.0 / np.arange(1, N_ITEMS + 1) ** 0.9
    pop = pop[rng.permutation(N_ITEMS)]
    pop /= pop.sum()
    train, test, t = [], [], 0
    for u in range(N_USERS):
        items = rng.choice(N_ITEMS, size=act[u], replace=False, p=pop)
        cut = int(np.ceil(len(items) * 0.7))
        for k, it in enumerate(items):
            (train if k < cut else test).append((u, int(it), t))
            t += 1
    train, test = np.array(train, np.int64), np.array(test, np.int64)
    # every id must occur so that num_users / num_items are the intended ones
    train[0, 1] = N_ITEMS - 1 if (N_ITEMS - 1) not in train[train[:, 0] == 0][:, 1] else train[0, 1]
    assert train[:, 0].max() == N_USERS - 1 and max(train[:, 1].max(), test[:, 1].max()) == N_ITEMS - 1
    return train, test


@pytest.mark.parametrize("n_dim", [64, 32, 128, 50])
def test_bprmf_config0_shape_fit_replays_the_oracle(n_dim, tmp_path, monkeypatch):
    """n_dim is a free integer in the reference (BPRMF.py:27,51): 64 takes the one-launch step and the fused evaluator, 32 and 50
    live in zero-padded 64-float rows (same kernels), 128 in 128-float rows (skr_bpr_step_dim + a dense Adam launch per step,
    evaluation through the score-matrix path)"""
    import random
    import torch
    from oracle import oracle as O
    from skrec import RunConfig
    from skrec.recommender.BPRMF import BPRMF
    from skrec.utils.py.random import reset_global_sampler
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("SKR_ADAM_BLOCK", raising=False)          # the shipped default (32 steps per block)
    train, test = _ml100k_shaped(np.random.default_rng(100))
    root = tmp_path / "ml100k_shaped"
    root.mkdir()
    for name, rows in (("train", train), ("test", test)):
        with open(root / f"ml100k_shaped.{name}", "w") as f:
            for u, i, t in rows:
                f.write(f"{u}\t{i}\t1.0\t{t}\n")
    np.random.seed(SEED)
    random.seed(SEED)
    torch.manual_seed(SEED)
    reset_global_sampler(2020)
    metric, top_k = ("Precision", "Recall", "MAP", "NDCG", "MRR"), (10, 20, 30, 40, 50, 100)
    rc = RunConfig(recommender="BPRMF", data_dir=str(root), file_column="UIRT", sep="\t", hyperopt=False, gpu_id=0,
                   metric=metric, top_k=top_k, test_batch_size=64, test_thread=4, seed=SEED)
    lr, reg, bsz, epochs = 1e-3, 1e-3, 1024, 3
    m = BPRMF(rc, dict(lr=lr, reg=reg, n_dim=n_dim, batch_size=bsz, epochs=epochs))
    assert (m.num_users, m.num_items) == (N_USERS, N_ITEMS)
    assert m.adam_block == (32 if n_dim <= 64 else 1) and m.user_embeddings.shape == (N_USERS, n_dim)
    U, V, b = (t.cpu().numpy().copy() for t in (m.user_embeddings, m.item_embeddings, m.item_biases))
    np_state = np.random.get_state()                              # fit() draws one permutation per epoch from here on

    reports, losses, snaps = [], [], []
    ev, te = m.evaluate, m.train_epoch
    test_users_all = np.fromiter(m.evaluator.user_pos_test.keys(), dtype=np.int32)

    def evaluate(test_users=None):
        r = ev(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        rows, _, _ = m.evaluator.per_user_rows(m, test_users_all)           # the per-user metric rows behind the report
        snaps.append((rows.copy(),) + tuple(t.cpu().numpy().copy() for t in (m.user_embeddings, m.item_embeddings, m.item_biases)))
        return r

    def train_epoch(it):
        # fit() runs on a stream of the library's own (base.on_compute_stream), ordered against the caller's
        assert torch.cuda.current_stream() != torch.cuda.default_stream()
        te(it)
        losses.append(m.step_losses.cpu().numpy().copy())
    m.evaluate, m.train_epoch = evaluate, train_epoch
    m.fit()
    assert torch.cuda.current_stream() == torch.cuda.default_stream()
    got_losses = np.concatenate(losses, 0)

    # ---- the oracle's replay -----------------------------------------------------------------------------------
    # S1 arrays (data_iterator.py:65-78): users ascending, a user's positives in file order
    order = np.lexsort((train[:, 2], train[:, 0]))
    users_ary, pos_items = train[order, 0].astype(np.int32), train[order, 1].astype(np.int32)
    rowptr = np.zeros(N_USERS + 1, np.int64)
    rowptr[1:] = np.cumsum(np.bincount(users_ary, minlength=N_USERS))
    user_train = {u: pos_items[rowptr[u]:rowptr[u + 1]] for u in range(N_USERS)}
    test_order = np.lexsort((test[:, 2], test[:, 0]))
    tu, ti = test[test_order, 0], test[test_order, 1].astype(np.int32)
    user_test = {int(u): ti[tu == u] for u in np.unique(tu)}
    assert list(user_test.keys()) == [int(u) for u in test_users_all]
    sampler = O.Sampler(2020)
    opt = O.Adam([U, V, b], lr=lr)
    np.random.set_state(np_state)
    want_losses, want_reports = [], []
    n_steps = -(-len(users_ary) // bsz)
    assert n_steps > 2 * 32 and n_steps % 32 != 0                # whole blocks AND a ragged one
    K = max(top_k)
    for ep in range(epochs):
        neg = sampler.sample_epoch(N_ITEMS, rowptr, pos_items, 1)
        perm = np.random.permutation(len(users_ary))
        for bu, bi, bj in O.pairwise_epoch(users_ary, pos_items, neg, bsz, perm):
            loss, l2, gU, gV, gb, _, _ = O.bpr_batch(U, V, b, U, V, bu, bi, bj, 1.0, reg, 1.0)
            want_losses.append((loss, l2))
            opt.step([gU, gV, gb])
        # (1) the trajectory: the model's tables at this point are the oracle's
        rows_gpu, Ug, Vg, bg = snaps[ep]
        for got, want in ((Ug, U), (Vg, V), (bg, b)):
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)
        # (2) the evaluator, on the model's OWN tables (so that training noise is not in the comparison): the reference's
        # loop -- fp32 host GEMM, -inf masking, evaluate.h's partial sort, metric.h, float32 mean.  Per-user rows must be
        # BIT-equal except for users where two of the best K+1 scores are closer than fp32 summation noise (the order of
        # a dot product's 64 additions is the GEMM library's choice in the reference too); those users are proven to be
        # such near-ties in float64 and are few.
        predict = lambda us: (Ug[us] @ Vg.T + bg[None, :]).astype(np.float32)   # noqa: E731
        names, vals, rows_cpu = O.ranking_evaluate(predict, user_train, user_test, metric=list(metric), top_k=top_k, batch_size=64)
        assert rows_gpu.shape == rows_cpu.shape
        differ = np.nonzero((rows_gpu != rows_cpu).any(1))[0]
        assert len(differ) <= max(1, len(test_users_all) // 100), f"{len(differ)} users ranked differently"
        for r in differ:
            u = int(test_users_all[r])
            sc = Ug[u].astype(np.float64) @ Vg.astype(np.float64).T + bg
            sc[user_train[u]] = -np.inf
            top = np.sort(sc)[::-1][:K + 1]
            assert np.min(top[:-1] - top[1:]) < 2e-5 * np.abs(top).max(), f"user {u} differs without a near-tie"
        same = np.ones(len(test_users_all), bool)
        same[differ] = False
        assert np.array_equal(rows_gpu[same], rows_cpu[same])
        # the report itself: the float32 mean over the same rows (evaluator.py:208) -> identical when no user differs
        sel = (np.sort(top_k) - 1)
        rep_from_rows = np.mean(rows_gpu, axis=0).reshape(len(metric), K)[:, sel].reshape(-1)
        np.testing.assert_array_equal(reports[ep], rep_from_rows)
        # the north-star bar on the metric means: 1e-5 relative; k proven near-tie users can move a mean by at most k / n
        np.testing.assert_allclose(reports[ep], vals, rtol=1e-5, atol=len(differ) / len(test_users_all))
        if len(differ) == 0:
            np.testing.assert_array_equal(reports[ep], vals)
    want_losses = np.array(want_losses, np.float32)
    assert got_losses.shape == want_losses.shape == (epochs * n_steps, 2)
    np.testing.assert_allclose(got_losses[:, 0], want_losses[:, 0], rtol=1e-5)
    np.testing.assert_allclose(got_losses[:, 1], want_losses[:, 1], rtol=1e-5)
    assert list(m.evaluator.metrics_list) == names


@pytest.mark.parametrize("model_name,dim", [("LightGCN", 64), ("LayerGCN", 64), ("LightGCN", 32), ("LightGCN", 128), ("LayerGCN", 32),
                                            ("LayerGCN", 128)])
def test_graph_models_config0_shape_fit_replays_the_oracle(model_name, dim, tmp_path, monkeypatch):
    """LightGCN (3 layers, `pre` adjacency) and LayerGCN (4 layers) on the same ml-100k-shaped set through the API, against the
    oracle's replay (exact-stream negatives, numpy permutations, oracle.lightgcn_step / layergcn_step -- float32 scipy
    propagation with explicit backward -- and oracle.Adam): per-step losses 1e-5, the ego table after every epoch 3e-6.
    The graph has 139 k non-zeros, so the propagation goes through skr_spmm_plan_* with the step's row / column masks live
    (the 64-user golden set needs SKR_SPMM_PLAN=1 for that).  embed_size / embed_dim are free integers in the reference
    (LightGCN.py:34, LayerGCN.py:28): 32 lives in zero-padded 64-float rows, 128 is propagated in two 64-column slices
    (the row stride of skr_spmm_epilogue) with LayerGCN's refinements as launches of their own over the whole rows."""
    import random
    import scipy.sparse as sp
    import torch
    from oracle import oracle as O
    from skrec import RunConfig
    from skrec.utils.py.random import reset_global_sampler
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("SKR_SPMM_PLAN", raising=False)
    train, test = _ml100k_shaped(np.random.default_rng(100))
    root = tmp_path / "ml100k_shaped"
    root.mkdir()
    for name, rows in (("train", train), ("test", test)):
        with open(root / f"ml100k_shaped.{name}", "w") as f:
            for u, i, t in rows:
                f.write(f"{u}\t{i}\t1.0\t{t}\n")
    np.random.seed(SEED)
    random.seed(SEED)
    torch.manual_seed(SEED)
    reset_global_sampler(2020)
    rc = RunConfig(recommender=model_name, data_dir=str(root), file_column="UIRT", sep="\t", hyperopt=False, gpu_id=0,
                   metric=("Recall", "NDCG"), top_k=(10, 20), test_batch_size=64, test_thread=4, seed=SEED)
    lr, bsz, epochs = 1e-3, 1024, 2
    if model_name == "LightGCN":
        from skrec.recommender.LightGCN import LightGCN as Model
        reg, n_layers = 1e-3, 3
        m = Model(rc, dict(lr=lr, reg=reg, embed_size=dim, n_layers=n_layers, adj_type="pre", batch_size=bsz, epochs=epochs))
    else:
        from skrec.recommender.LayerGCN import LayerGCN as Model
        reg, n_layers = 1e-2, 4
        m = Model(rc, dict(lr=lr, reg=reg, embed_dim=dim, n_layers=n_layers, dropout=0.0, batch_size=bsz, epochs=epochs))
    assert (m.num_users, m.num_items) == (N_USERS, N_ITEMS) and m.engine is None and m.user_embeddings.shape == (N_USERS, dim)
    E0 = np.concatenate([m.user_embeddings.cpu().numpy(), m.item_embeddings.cpu().numpy()], 0).copy()
    np_state = np.random.get_state()
    losses, snaps = [], []
    te = m.train_epoch

    def train_epoch(it):
        te(it)
        losses.append(m.step_losses.cpu().numpy().copy())
        snaps.append(np.concatenate([m.user_embeddings.cpu().numpy(), m.item_embeddings.cpu().numpy()], 0).copy())
    m.train_epoch = train_epoch
    m.fit()
    got_losses = np.concatenate(losses, 0)
    # the oracle's replay
    order = np.lexsort((train[:, 2], train[:, 0]))
    users_ary, pos_items = train[order, 0].astype(np.int32), train[order, 1].astype(np.int32)
    rowptr = np.zeros(N_USERS + 1, np.int64)
    rowptr[1:] = np.cumsum(np.bincount(users_ary, minlength=N_USERS))
    N = N_USERS + N_ITEMS
    R = sp.csr_matrix((np.ones(len(users_ary), np.float32), (users_ary, pos_items + N_USERS)), shape=(N, N))
    A = (R + R.T).tocsr()
    deg = np.asarray(A.sum(1)).reshape(-1).astype(np.float64)
    if model_name == "LightGCN":       # D^-1/2 A D^-1/2, zero degree -> 0 (utils/common.py:27-40)
        with np.errstate(divide="ignore"):
            dinv = np.where(deg > 0, np.power(deg, -0.5), 0.0)
    else:                              # +1e-7 on the degrees (LayerGCN.py:186)
        dinv = np.power(deg + 1e-7, -0.5)
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).astype(np.float32).tocsr()
    sampler = O.Sampler(2020)
    opt = O.Adam([E0], lr=lr)
    np.random.set_state(np_state)
    want = []
    for ep in range(epochs):
        neg = sampler.sample_epoch(N_ITEMS, rowptr, pos_items, 1)
        perm = np.random.permutation(len(users_ary))
        for bu, bi, bj in O.pairwise_epoch(users_ary, pos_items, neg, bsz, perm):
            if model_name == "LightGCN":
                loss, l2, G = O.lightgcn_step(A, E0, N_USERS, bu, bi, bj, n_layers, reg, bsz)
                want.append((loss, l2))
            else:
                loss, l2, G = O.layergcn_step(A, E0, N_USERS, bu, bi, bj, n_layers, reg)
                want.append((loss, l2))
            opt.step([G])
        diff = np.abs(snaps[ep] - E0)
        if model_name == "LightGCN":
            assert diff.max() <= 3e-6
        else:
            # LayerGCN divides by row norms, and Adam turns the SIGN of a gradient element into a step of lr whatever its size:
            # where a gradient element is at the level of fp32 summation noise (rows far from the batch), two correct
            # evaluations can step in opposite directions.  Nearly all elements agree to 3e-6; the rest stay within a few lr.
            assert (diff <= 3e-6).mean() >= 0.99 and diff.max() <= 5e-4, ((diff <= 3e-6).mean(), diff.max())
    want = np.array(want, np.float32)
    assert got_losses.shape == want.shape
    np.testing.assert_allclose(got_losses[:, 0], want[:, 0], rtol=1e-5)
    # (the regulariser's value is one fp32 sum of 3 * batch * dim squares, added in wavefront order here and pairwise by numpy:
    #  at dim = 128 the two roundings of the same sum are up to 1.2e-5 apart)
    np.testing.assert_allclose(got_losses[:, 1], want[:, 1], rtol=1e-5 if dim <= 64 else 2e-5)
