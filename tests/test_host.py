"""CPU suite, part 3: host-side mirror of the reference interface (no GPU): configs, registry,
dataset views, batch iterator contracts, evaluator bookkeeping."""
import os
import sys

import numpy as np
import pytest

import skrec
from skrec import RunConfig, ModelRegistry, merge_config_with_cmd_args
from skrec.io import RSDataset
from skrec.utils.py import BatchIterator, MetricReport, EarlyStopping, RankingEvaluator
from helpers import tiny_arrays


def test_public_api_surface():
    for name in ("RunConfig", "ModelRegistry", "merge_config_with_cmd_args", "PairwiseIterator", "PointwiseIterator",
                 "RankingEvaluator", "MetricReport", "EarlyStopping", "randint_choice", "batch_randint_choice",
                 "BatchIterator", "RSDataset", "ImplicitFeedback", "Config", "ModelConfig"):
        assert hasattr(skrec, name), name
    from skrec.io import PairwiseSampler, PointwiseSampler, PairwiseIterator, PointwiseIterator
    assert PairwiseSampler is PairwiseIterator and PointwiseSampler is PointwiseIterator
    from skrec.utils.hyperopt import HyperOpt  # noqa: F401  (importable without the hyperopt package)


def test_run_config_defaults_and_validation():
    rc = RunConfig()
    assert rc.test_batch_size == 64 and rc.test_thread == 4 and rc.seed == 2021 and rc.top_k[-1] == 100
    assert list(dict(rc.items()))[:3] == ["recommender", "data_dir", "file_column"]
    with pytest.raises(AssertionError):
        RunConfig(test_batch_size=0)
    assert RunConfig(foo=1).recommender == "BPRMF"  # unknown keys are swallowed like the reference


def test_cmd_arg_merge(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["x", "--lr", "1e-2", "--top_k", "[10,20]", "--recommender", "LightGCN",
                                      "--hyperopt", "false", "--data_dir", "dataset/a_b"])
    d = merge_config_with_cmd_args({"lr": 1e-3})
    assert d == {"lr": 0.01, "top_k": [10, 20], "recommender": "LightGCN", "hyperopt": False, "data_dir": "dataset/a_b"}
    monkeypatch.setattr(sys, "argv", ["x", "--lr"])
    with pytest.raises(SyntaxError):
        merge_config_with_cmd_args({})
    monkeypatch.setattr(sys, "argv", ["x", "lr", "1"])
    with pytest.raises(SyntaxError):
        merge_config_with_cmd_args({})


def test_registry_and_model_configs():
    reg = ModelRegistry()
    for name in ("BPRMF", "LightGCN", "LayerGCN"):
        assert reg.load_skrec_model(name)
        model, cfg = reg.get_model(name)
        assert model.__name__ == name and cfg.__name__ == name + "Config"
    assert not reg.load_skrec_model("NoSuchModel")
    from skrec.recommender.BPRMF import BPRMFConfig
    from skrec.recommender.LightGCN import LightGCNConfig
    from skrec.recommender.LayerGCN import LayerGCNConfig
    b, l, y = BPRMFConfig(), LightGCNConfig(), LayerGCNConfig()
    assert (b.lr, b.reg, b.n_dim, b.batch_size, b.epochs, b.early_stop) == (1e-3, 1e-3, 64, 1024, 1000, 200)
    assert (l.n_layers, l.adj_type, l.early_stop) == (3, "pre", 100)
    assert (y.reg, y.n_layers, y.batch_size, y.dropout) == (1e-2, 4, 2048, 0.0)
    assert BPRMFConfig.num_combos() == 20
    with pytest.raises(AssertionError):
        LightGCNConfig(adj_type="bogus")


def test_dataset_views(tiny_dir, golden):
    ds = RSDataset(tiny_dir, "\t", "UIRT")
    U, I, rowptr, pos, srt, uary = tiny_arrays(golden)
    assert (ds.num_users, ds.num_items) == (U, I) and ds.data_name == "tiny"
    rp, fo, so = ds.train_data.to_csr_arrays()
    assert np.array_equal(rp, rowptr) and np.array_equal(fo, pos) and np.array_equal(so, srt)
    ud = ds.train_data.to_user_dict()
    assert list(ud) == sorted(ud) and 63 not in ud and all(v.dtype == np.int32 for v in ud.values())
    assert np.array_equal(np.concatenate(list(ud.values())), pos)
    pairs = ds.train_data.to_user_item_pairs()
    assert pairs.dtype == np.int32 and pairs.shape == (len(pos), 2)
    assert ds.train_data.to_coo_matrix().shape == (U, I)
    assert 5 not in ds.test_data.to_user_dict() and 63 in ds.test_data.to_user_dict()
    with pytest.raises(FileNotFoundError):
        RSDataset(tiny_dir + "_missing", "\t", "UIRT").num_users


def test_batch_iterator_contract():
    users, items = list(range(10)), list(range(10, 20))
    out = list(BatchIterator(users, items, batch_size=4, shuffle=False))
    assert [len(b[0]) for b in out] == [4, 4, 2] and out[2][1] == [18, 19]
    assert len(BatchIterator(users, batch_size=4, drop_last=True)) == 2
    np.random.seed(3)
    shuffled = [x for b in BatchIterator(users, batch_size=3, shuffle=True) for x in b]
    np.random.seed(3)
    assert shuffled == np.random.permutation(10).tolist()
    with pytest.raises(ValueError):
        BatchIterator([1, 2], [1], batch_size=1)


def test_metric_report_and_early_stopping():
    r1 = MetricReport(["NDCG@10", "Recall@10"], np.float32([0.2, 0.1]))
    assert r1.values_str.split("\t")[0].strip() == "0.20000000"
    with pytest.raises(KeyError):
        r1["HR@10"]
    es = EarlyStopping("NDCG@10", patience=2)
    assert not es(r1)
    worse = MetricReport(["NDCG@10", "Recall@10"], np.float32([0.1, 0.3]))
    assert not es(worse) and es(worse) and es.best_result is r1
    never = EarlyStopping("NDCG@10", patience=0)
    assert not any(never(worse) for _ in range(5))


def test_evaluator_bookkeeping():
    ev = RankingEvaluator({0: np.int32([1])}, {0: np.int32([2])}, metric=["Recall", "NDCG"], top_k=[20, 10])
    assert ev.metrics_list == ["Recall@10", "Recall@20", "NDCG@10", "NDCG@20"] and ev.max_top == 20
    assert RankingEvaluator(None, {0: [1]}, metric="MRR", top_k=3).metrics_list == ["MRR@1", "MRR@2", "MRR@3"]
    with pytest.raises(AssertionError):
        RankingEvaluator(None, {0: [1]}, metric=["HR"])
    with pytest.raises(AssertionError):
        RankingEvaluator(None, {})
    with pytest.raises(NotImplementedError):
        RankingEvaluator(None, {0: [1]}, top_k=513)
    assert RankingEvaluator(None, {0: [1]}, top_k=500).max_top == 500      # beyond the fused kernel's 128: the score-matrix path


def test_hot_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from skrec import _hip
    with pytest.raises(_hip.HipError):
        skrec.randint_choice(10, 3)


def test_time_ordered_instances_match_reference(golden, tiny_dir):
    """_generative_time_order_positive_items restated (data_iterator.py:44-78) == the columns the reference's
    sequential iterators produced on the tiny dataset (golden_iterators.npz, shuffle=False)."""
    from skrec.io import RSDataset
    from skrec.io.data_iterator import _time_ordered_instances
    g = golden("golden_iterators")
    ds = RSDataset(tiny_dir, "\t", "UIRT")
    ud = ds.train_data.to_user_dict_by_time()
    for tag, (npv, nnx, pad), n_take in (("spw", (3, 1, None), None), ("spw11", (1, 1, None), None),
                                         ("spwpad", (4, 2, ds.num_items), 600)):
        n_pos, users, seqs, nxt = _time_ordered_instances(ud, npv, nnx, pad)
        assert sum(n_pos.values()) == len(users)
        sl = slice(0, n_take)
        assert np.array_equal(users[sl], g[tag + "_e0_c0"])
        assert np.array_equal(seqs.squeeze()[sl], g[tag + "_e0_c1"])
        assert np.array_equal(nxt.squeeze()[sl], g[tag + "_e0_c2"])


def test_knowledge_graph_head_dict_order():
    import pandas as pd
    from skrec.io.dataset import KnowledgeGraph
    tri = np.array([[3, 0, 5], [1, 1, 2], [3, 1, 5], [1, 0, 9], [0, 0, 1]])
    kg = KnowledgeGraph(pd.DataFrame(tri, columns=["head", "relation", "tail"]))
    d = kg.to_head_dict()
    assert list(d) == [0, 1, 3] and kg.num_entities == 10 and kg.num_relations == 2
    assert d[3]["tail"].tolist() == [5, 5] and d[3]["relation"].tolist() == [0, 1] and d[1]["tail"].tolist() == [2, 9]


def test_unique_padded_rows():
    import torch
    from skrec.parallel import unique_padded_rows
    ids = torch.tensor([[5, 3, 5, -1, 3, 9], [7, 7, 7, 7, 7, 7], [0, 1, 2, 3, 4, 5]], dtype=torch.int32)
    got = unique_padded_rows(ids)
    assert got.dtype == torch.int32
    for row, src in zip(got.tolist(), ids.tolist()):
        real = [x for x in row if x >= 0]
        assert sorted(real) == sorted(set(x for x in src if x >= 0)) and len(real) == len(set(real))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 1000, 65536, 65537, 300_001])
def test_native_host_permutation_is_numpys(n):
    """skr_host_permutation == np.random.permutation(n) from the same generator state, and leaves the generator in the
    same state (reference: batch_iterator.py:61-63 draws one per epoch)"""
    from skrec import _hip
    np.random.seed(5 + n)
    np.random.random(n % 700)                      # somewhere inside a 624-word block
    np.random.normal()                             # a cached gaussian must survive
    a = np.random.permutation(n)
    ra = np.random.normal(size=3)
    np.random.seed(5 + n)
    np.random.random(n % 700)
    np.random.normal()
    b = _hip.host_permutation(n)
    rb = np.random.normal(size=3)
    assert b.dtype == np.int32 and np.array_equal(a, b) and np.array_equal(ra, rb)


def test_large_tables_parse_like_the_c_parser(tmp_path):
    """files above 1 MB go through pandas' pyarrow parser: same frame (values and dtypes) as the C parser"""
    import pandas as pd
    from skrec.io.dataset import _read_table
    rng = np.random.default_rng(3)
    n = 80_000
    df = pd.DataFrame({"user": rng.integers(0, 50_000, n), "item": rng.integers(0, 9_000, n),
                       "rating": rng.integers(1, 6, n).astype(np.float64), "time": rng.integers(0, 10 ** 9, n)})
    p = str(tmp_path / "big.train")
    df.to_csv(p, sep="\t", header=False, index=False)
    assert os.path.getsize(p) > (1 << 20)
    names = ["user", "item", "rating", "time"]
    got = _read_table(p, "\t", names, lambda m: None)
    want = pd.read_csv(p, sep="\t", header=None, names=names)
    assert got.dtypes.tolist() == want.dtypes.tolist() and got.equals(want)


def test_native_host_permutation_argument_checks():
    import ctypes as C
    from skrec import _hip
    L = _hip.lib()
    key = np.zeros(624, np.uint32)
    pos = C.c_int(624)
    out = np.zeros(4, np.int32)
    assert L.skr_host_permutation(None, C.byref(pos), 4, out.ctypes.data) == -1           # NULL state
    assert L.skr_host_permutation(key.ctypes.data, C.byref(pos), -1, out.ctypes.data) == -1  # negative n
    assert L.skr_host_permutation(key.ctypes.data, C.byref(pos), 1 << 31, out.ctypes.data) == -1
    bad = C.c_int(700)
    assert L.skr_host_permutation(key.ctypes.data, C.byref(bad), 4, out.ctypes.data) == -1   # position beyond the state
    assert L.skr_host_permutation(key.ctypes.data, C.byref(pos), 0, None) == 0               # empty permutation
    assert pos.value == 624


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 65, 1000, 4097, 262_144, 300_001])
def test_device_shuffle_bijection_on_the_host(n):
    """skr_shuffle_permutation_host = the keyed bijection skr_shuffle_gather evaluates per output row:
    a permutation of [0, n) for every n and seed, different for different seeds, no structure a mini-batch could see"""
    from skrec import _hip
    L = _hip.lib()
    a, b = np.empty(n, np.int32), np.empty(n, np.int32)
    _hip.check(L.skr_shuffle_permutation_host(2021, n, n, a.ctypes.data))
    _hip.check(L.skr_shuffle_permutation_host(2022, n, n, b.ctypes.data))
    assert np.array_equal(np.sort(a), np.arange(n)) and np.array_equal(np.sort(b), np.arange(n))
    half = np.empty(n // 2, np.int32)
    _hip.check(L.skr_shuffle_permutation_host(2021, n, n // 2, half.ctypes.data))
    assert np.array_equal(half, a[:n // 2])                       # a prefix of the same permutation
    if n >= 1000:
        assert (a == b).mean() < 0.01 and (a == np.arange(n)).mean() < 0.01
        x = a.astype(np.float64)
        assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.1
        # where do the elements of each 1/16 of the source land?  every 1/16 of the output gets its share
        H = np.zeros((16, 16))
        np.add.at(H, (np.minimum(np.arange(n) * 16 // n, 15), np.minimum(a.astype(np.int64) * 16 // n, 15)), 1)
        exp = n / 256
        assert ((H - exp) ** 2 / exp).sum() / 255 < 2.0           # chi-square per degree of freedom ~ 1
    assert L.skr_shuffle_permutation_host(1, 5, 6, a.ctypes.data) == -1
    assert L.skr_shuffle_permutation_host(1, 1 << 31, 1, a.ctypes.data) == -1


def test_bench_roofline_arithmetic():
    """bench.py's roofline entry of the blocked cold pass is plain arithmetic on the HIP-event times it is given: the
    mean over EVERY launch, the mean number of optimiser steps the launches applied, 20 B per cold parameter, and the
    by-traffic figure from the recorded counter bytes -- the numbers a reader re-derives from profiles/"""
    import importlib
    bench = importlib.import_module("bench")
    cold = [(0.50, 32, "pre")] * 6 + [(0.40, 32, "warmup")] + [(0.30, 20, "timed")]
    n_par, hot = 70_500_000, 5000
    r = bench.cold_roofline(n_par, hot, 32, cold, 0.39, True, 0.9e9, "profiles/x.json (recorded; command: y)")
    ms = (6 * 0.50 + 0.40 + 0.30) / 8
    assert r["launches_averaged"] == 8 and abs(r["avg_launch_ms"] - ms) < 1e-12
    assert abs(r["optimizer_steps_per_launch"] - (7 * 32 + 20) / 8) < 1e-12
    cold_bytes = (n_par - 64 * hot) * 20.0
    assert r["algorithmic_bytes_per_launch"] == cold_bytes
    assert abs(r["achieved"] - cold_bytes / (ms * 1e-3) / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["achieved_by_traffic"] - 0.9e9 / (ms * 1e-3) / 1e9) < 1e-6 and r["traffic_source"].startswith("profiles/")
    assert r["launches_by_phase"] == {"pre": 6, "warmup": 1, "timed": 1} and r["full_k_launches"] == 7
    assert r["timed_region_launches"] == [{"ms": 0.30, "optimizer_steps": 20}]
    assert abs(r["alone"]["achieved"] - cold_bytes / 0.39e-3 / 1e9) < 1e-6
    r2 = bench.cold_roofline(n_par, hot, 32, cold, None, True, None, "x")
    assert r2["traffic"] is None and r2["achieved_by_traffic"] is None and r2["traffic_source"] is None and "alone" not in r2


def test_fused_step_words_on_the_host():
    """The words of the one-launch BPRMF step (include/skrec_hip.h, skr_bpr_fused_step) as skrec.recommender.fused.build_fused_meta
    derives them with sorts and scans (the GPU suite compares skr_bpr_fused_plan with it), against a literal walk over the
    steps of a small block: slot, earlier namings mod 6, previous naming step, one owner per (step, row), slot tables"""
    import torch
    from skrec.recommender.fused import build_fused_meta
    rng = np.random.default_rng(3)
    k, b, nU, nI = 9, 16, 12, 150
    u = rng.integers(0, nU, k * b).astype(np.int32)
    i = rng.integers(0, nI, k * b).astype(np.int32)
    j = rng.integers(0, nI, k * b).astype(np.int32)
    meta, slot_block, slot_fin, n_slots = build_fused_meta(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j), 1, k, b, 0, nU, nU + nI)
    meta, slot_block, slot_fin = meta[0].numpy().astype(np.int64), slot_block[0].numpy(), slot_fin[0].numpy()
    seen, last, first = {}, {}, {}                     # row -> namings so far / last naming step / first naming step
    owners = set()
    for s in range(k):
        sl = slice(s * b, (s + 1) * b)
        refs = np.stack([u[sl], nU + i[sl], nU + j[sl], nU + nI + (i[sl] >> 6), nU + nI + (j[sl] >> 6)])       # [5, b]
        named_now = set()
        for r in range(5):
            for c in range(b):
                row, w = int(refs[r, c]), int(meta[s, r, c])
                slot, n0, own, prev1 = w & 0xfffff, (w >> 20) & 7, (w >> 23) & 1, (w >> 24) & 0x7f
                assert slot_block[slot] == row
                assert n0 == seen.get(row, 0) % 6 and prev1 == (last[row] + 1 if row in last else 0)
                if own:
                    assert (s, row) not in owners
                    owners.add((s, row))
                named_now.add(row)
        assert {(s, r_) for r_ in named_now} <= owners          # every pair of this step has its owner
        for row in named_now:
            seen[row] = seen.get(row, 0) + 1
            last[row] = s
            first.setdefault(row, s)
    n = int(n_slots[0])
    assert n == len(seen) and sorted(slot_block[:n]) == sorted(seen) and (slot_block[n:] == -1).all()
    for slot in range(n):
        row = int(slot_block[slot])
        assert slot_fin[slot] == (seen[row] % 6) | (last[row] << 8) | (first[row] << 16)


def test_gru_session_schedule_is_the_references_loop():
    """GRU4RecPlus._schedule (the generator train_epoch prepares its 32-step blocks from) against a literal restatement of the
    reference's session-parallel loop (GRU4RecPlus.py:208-247): the same positions step after step, the same slots reset
    before the same steps, numpy's generator left in the same state (one permutation per epoch, drawn at the first step)"""
    from types import SimpleNamespace
    from skrec.recommender.GRU4RecPlus import GRU4RecPlus
    rng = np.random.default_rng(3)
    for n_sessions, b in ((40, 8), (9, 4), (200, 16), (5, 5)):
        lens = rng.integers(2, 12, n_sessions)
        lens[rng.integers(0, n_sessions, 3)] = 2                    # sessions of one step
        offset_idx = np.zeros(n_sessions + 1, np.int32)
        offset_idx[1:] = np.cumsum(lens)
        # the reference's loop, verbatim control flow, recording what each step reads and which slots were zeroed before it
        np.random.seed(11)
        want, pending = [], None
        user_idx = np.random.permutation(len(offset_idx) - 1)
        iters = np.arange(b, dtype=np.int32)
        maxiter = iters.max()
        start = offset_idx[user_idx[iters]]
        end = offset_idx[user_idx[iters] + 1]
        finished = False
        while not finished:
            min_len = (end - start).min()
            for i in range(min_len - 1):
                want.append(((start + i).astype(np.int64).copy(), pending))
                pending = None
            start = start + min_len - 1
            mask = np.arange(len(iters))[(end - start) <= 1]
            for idx in mask:
                maxiter += 1
                if maxiter >= len(offset_idx) - 1:
                    finished = True
                    break
                iters[idx] = maxiter
                start[idx] = offset_idx[user_idx[maxiter]]
                end[idx] = offset_idx[user_idx[maxiter] + 1]
            if len(mask):
                pending = mask if pending is None else np.union1d(pending, mask)
        state_ref = np.random.get_state()[1].copy()
        np.random.seed(11)
        stub = SimpleNamespace(offset_idx=offset_idx, config=SimpleNamespace(batch_size=b))
        got = list(GRU4RecPlus._schedule(stub))
        assert np.array_equal(np.random.get_state()[1], state_ref)
        assert len(got) == len(want) and len(got) > 0
        for (gp, gr), (wp, wr) in zip(got, want):
            assert np.array_equal(gp, wp)
            assert (gr is None) == (wr is None) and (gr is None or np.array_equal(gr, wr))


def test_padded_width_and_columns():
    import torch
    from skrec.recommender.LightGCN import pad_columns, padded_width
    assert [padded_width(d) for d in (1, 32, 50, 64, 65, 100, 128, 129, 256)] == [64, 64, 64, 64, 128, 128, 128, 192, 256]
    with pytest.raises(NotImplementedError):
        padded_width(257)
    t = torch.arange(6.0).view(2, 3)
    p = pad_columns(t, 64)
    assert p.shape == (2, 64) and torch.equal(p[:, :3], t) and float(p[:, 3:].abs().sum()) == 0.0 and pad_columns(p, 64) is p


def test_on_compute_stream_is_a_passthrough_without_a_gpu(monkeypatch):
    """fit() is wrapped by base.on_compute_stream; without a GPU (and with SKR_COMPUTE_STREAM=0) the wrapper only calls through"""
    from skrec.recommender.base import on_compute_stream

    class M:
        @on_compute_stream
        def fit(self, a, b=2):
            """doc"""
            return a + b
    assert M().fit(1, b=5) == 6 and M.fit.__doc__ == "doc"
    monkeypatch.setenv("SKR_COMPUTE_STREAM", "0")
    assert M().fit(1) == 3
