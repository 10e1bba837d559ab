"""GPU parity: edge cases the reference's code handles implicitly (ragged / empty inputs, tiny and
awkward sizes), the predict() score kernel, and the bench.py output contract."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from helpers import random_csr

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def test_fused_tiny_catalogues_and_duplicate_users(fused_mode):
    from gpu_utils import fused_topk
    rng = np.random.default_rng(0)
    for I, K, B in ((10, 10, 3), (31, 7, 70), (64, 64, 5), (65, 1, 129), (97, 33, 64)):
        Ut = rng.integers(-2, 3, (9, 64)).astype(np.float32)
        It = rng.integers(-2, 3, (I, 64)).astype(np.float32)
        users = rng.integers(0, 9, B).astype(np.int32)          # duplicates on purpose
        rowptr, items = random_csr(rng, 9, I, 0, max(0, min(I - K, 6)))
        ids, sc = fused_topk(Ut, users, It, None, rowptr, items, K)
        full = Ut[users] @ It.T
        for r, u in enumerate(users):
            full[r, items[rowptr[u]:rowptr[u + 1]]] = -np.inf
            want = O.topk_ids_lowid(full[r], K)
            assert np.array_equal(ids[r], want), (I, K, B, r)
            assert np.array_equal(sc[r], full[r, want])


def test_eval_scores_zero_users_and_single_item_rows():
    import torch
    from gpu_utils import eval_scores
    from skrec import _hip
    rows, ids, _ = eval_scores(np.float32([[0.5]]), [[0]], [1, 2, 3, 4, 5], 1)
    assert ids.tolist() == [[0]] and np.array_equal(rows, O.eval_score_matrix(np.float32([[0.5]]), [[0]], [1, 2, 3, 4, 5], 1))
    # n_users == 0 is a no-op, not an error
    _hip.check(_hip.lib().skr_eval_scores(_hip.ptr(torch.zeros(4, device="cuda")), 0, 4, 4, None, None, None, 0, 2, None,
                                          None, None, _hip.stream()))


def test_sampler_ragged_shapes():
    from gpu_utils import ExactSampler, fast_epoch
    from fast_sampler_twin import sample_fast
    # leading / trailing empty rows, one huge row, num_neg > 1
    lens = np.array([0, 0, 3, 0, 900, 1, 0, 0], np.int64)
    I = 1000
    rng = np.random.default_rng(3)
    rowptr = np.zeros(len(lens) + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    pos = np.concatenate([np.sort(rng.choice(I, l, replace=False)) for l in lens]).astype(np.int32)
    ref, gpu = O.Sampler(2020), ExactSampler(2020)
    for nn in (1, 4):
        assert np.array_equal(gpu.epoch(I, rowptr, pos, nn), ref.sample_epoch(I, rowptr, pos, nn).reshape(-1))
    assert np.array_equal(fast_epoch(5, 1, 0, I, rowptr, pos, 3), sample_fast(5, 1, 0, I, rowptr, pos, 3))
    # a row that covers the whole catalogue is rejected like pyx_random.pyx:49
    full = np.arange(4, dtype=np.int32)
    with pytest.raises(ValueError):
        gpu.epoch(4, np.array([0, 4], np.int64), full, 1)


def test_iterator_rejects_bad_arguments(tiny_dir):
    from skrec.io import RSDataset, PairwiseIterator, PointwiseIterator
    train = RSDataset(tiny_dir, "\t", "UIRT").train_data
    with pytest.raises(ValueError):
        PairwiseIterator(train, num_neg=0)
    with pytest.raises((AssertionError, ValueError)):
        PointwiseIterator(train, num_neg=0)
    with pytest.raises(ValueError):
        PairwiseIterator(train, sampler_mode="bogus")
    it = PairwiseIterator(train, batch_size=100, shuffle=False, sampler_mode="fast")
    u, i, j = next(iter(it))
    assert u.shape == i.shape == j.shape == (100,) and j.dtype == np.int32


def test_score_matrix_predict_surface():
    import torch
    from gpu_utils import to_dev
    from skrec import _hip
    rng = np.random.default_rng(6)
    U = rng.standard_normal((50, 64)).astype(np.float32)
    V = rng.standard_normal((333, 64)).astype(np.float32)
    b = rng.standard_normal(333).astype(np.float32)
    users = [3, 3, 49, 0, 17]
    got = _hip.score_matrix(to_dev(U), users, to_dev(V), to_dev(b)).cpu().numpy()
    want = torch.matmul(torch.from_numpy(U[users]), torch.from_numpy(V).T).numpy() + b
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    got2 = _hip.score_matrix(to_dev(U), users, to_dev(V), None).cpu().numpy()
    np.testing.assert_allclose(got2, want - b, rtol=1e-5, atol=1e-5)


def test_bench_contract_on_a_small_workload():
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "8", "--warmup", "2", "--users", "60000",
                        "--items", "5000", "--interactions", "1500000", "--eval-users", "4096"],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 20 * cb["value"]


def test_dataset_csr_gpu_sort_equals_numpy(monkeypatch):
    """ingest: the CSR views built with the GPU sorts (large tables) are identical to the numpy ones"""
    import pandas as pd
    from skrec.io import dataset as D
    rng = np.random.default_rng(5)
    n, nu, ni = 1_300_000, 40_000, 3_000
    df = pd.DataFrame({"user": rng.integers(0, nu, n), "item": rng.integers(0, ni, n), "rating": 1.0, "time": np.arange(n)})
    a = D.ImplicitFeedback(df.copy(), nu, ni).to_csr_arrays()
    monkeypatch.setattr(D, "_sort_device", lambda n_: None)
    b = D.ImplicitFeedback(df.copy(), nu, ni).to_csr_arrays()
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x, y)


@pytest.mark.parametrize("adj_type", ["plain", "norm", "gcmc", "pre", "mean_plus_eye"])
def test_lightgcn_adjacency_device_build_equals_scipy(adj_type):
    """large graphs skip scipy: the device-built CSR has the structure of LightGCN._create_adj_mat's matrix
    (LightGCN.py:142-169, duplicate pairs summed, zero-degree rows zero) and its values to 1 ulp"""
    import scipy.sparse as sp
    import torch
    from skrec.recommender.LightGCN import build_adjacency, build_adjacency_device
    rng = np.random.default_rng(9)
    nu, ni, n = 700, 300, 6000
    users, items = rng.integers(0, nu - 5, n), rng.integers(0, ni - 3, n)      # some zero-degree nodes, some duplicates
    want = sp.csr_matrix(build_adjacency(users, items, nu, ni, adj_type))
    want.sum_duplicates()
    want.sort_indices()
    adj, adj_t = build_adjacency_device(users, items, nu, ni, adj_type, torch.device("cuda", 0))
    for got, ref in ((adj, want), (adj_t, sp.csr_matrix(want.T))):
        ref.sort_indices()
        assert np.array_equal(got.rowptr.cpu().numpy(), ref.indptr)
        assert np.array_equal(got.col.cpu().numpy()[:got.nnz], ref.indices)
        np.testing.assert_allclose(got.val.cpu().numpy()[:got.nnz], ref.data, rtol=3e-7, atol=0)


def test_layergcn_device_adjacency_equals_scipy(tiny_dir, monkeypatch, tmp_path):
    """LayerGCN's large-graph path (device-built normalised adjacency) == get_norm_adj_mat restated with scipy"""
    import scipy.sparse as sp
    from skrec import RunConfig
    from skrec.recommender import LayerGCN as M
    monkeypatch.chdir(tmp_path)
    rc = RunConfig(recommender="LayerGCN", data_dir=tiny_dir, file_column="UIRT", sep="\t", metric=("Recall",), top_k=(5,))
    host = M.LayerGCN(rc, dict(epochs=1))
    monkeypatch.setattr(M, "DEVICE_ADJ_MIN_PAIRS", 1)
    dev = M.LayerGCN(rc, dict(epochs=1))
    for a, b in ((host.adj.rowptr, dev.adj.rowptr), (host.adj.col, dev.adj.col)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_allclose(dev.adj.val.cpu().numpy(), host.adj.val.cpu().numpy(), rtol=2e-7, atol=0)
