"""GPU parity, multi-process: the user-sharded LightGCN engine (skrec.parallel.ShardedLightGCN) on TWO
ranks replays the reference's recorded single-process trajectory -- same global batches, per-step
losses within 1e-5, final tables within fp32 noise.  Both ranks share the one GPU of the test box and
talk over gloo (the collective semantics are those of RCCL; the kernels are the real ones)."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches():
    """the reference run's global batches: oracle stream for the negatives, numpy permutation per epoch"""
    from oracle import oracle as O
    from helpers import csr_from_pairs
    d = np.load(os.path.join(GOLDEN, "tiny_dataset.npz"))
    U, I = int(d["num_users"]), int(d["num_items"])
    rowptr, pos, _, uary = csr_from_pairs(d["train"][:, 0], d["train"][:, 1], U)
    s = O.Sampler(2020)
    np.random.seed(2021)
    out = []
    for _ in range(2):
        neg = s.sample_epoch(I, rowptr, pos, 1)
        perm = np.random.permutation(len(pos))
        for st in range(0, len(pos), 256):
            idx = perm[st:st + 256]
            out.append((uary[idx], pos[idx], neg[idx]))
    return out


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from skrec.parallel import DistContext, ShardedLightGCN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))
    U, I = g["U0"].shape[0], g["V0"].shape[0]
    adj = sp.csr_matrix((g["adj_val"], (g["adj_idx"][0], g["adj_idx"][1])), shape=(U + I, U + I))
    eng = ShardedLightGCN(DistContext(rank, world), adj, g["U0"], g["V0"], n_layers=3, lr=1e-3, reg=1e-3, batch_size_cfg=256)
    dev = eng.device
    losses = []
    for u, i, j in _batches():
        eng.train_step(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(j).to(dev))
        losses.append(eng.loss.cpu().numpy().copy())
    try:                      # after a training step only the batch's rows of `final` are valid
        eng.whole_final()
        guarded = False
    except ValueError:
        guarded = True
    assert guarded
    eng.propagate()
    uf = torch.zeros((U, 64), device=dev)
    uf[torch.from_numpy(eng.mine).to(dev)] = eng.whole_final()[:eng.n_local]
    eng.ctx.all_reduce(uf)
    ret[rank] = dict(losses=np.stack(losses), U1=eng.gather_user_table().cpu().numpy(), V1=eng.item_rows.cpu().numpy(),
                     Uf=uf.cpu().numpy(), Vf=eng.whole_final()[eng.n_local:].cpu().numpy(), n_local=eng.n_local)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,plan", [(1, "auto"), (2, "auto"), (2, "1"), (3, "1")])
def test_sharded_lightgcn_replays_reference(world, plan, monkeypatch):
    """plan "1": the SpMM plan (row / column masks honoured, compact exchanges of exactly the batch's rows) although the
    graph is tiny; "auto": the plan-free kernel, which computes whole products"""
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)        # the spawned ranks inherit it
    g = np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        res = {k: ret[k] for k in range(world)}
    assert sum(r["n_local"] for r in res.values()) == g["U0"].shape[0]
    for r in res.values():
        np.testing.assert_allclose(r["losses"][:, 0], g["bpr_mean"], rtol=1e-5)
        np.testing.assert_allclose(r["losses"][:, 1], g["l2"], rtol=1e-5)
        np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["Uf"], g["Uf"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["Vf"], g["Vf"], rtol=0, atol=3e-6)
    if world > 1:   # replicas of the item table stay bit-identical
        assert np.array_equal(res[0]["V1"], res[1]["V1"])


def _api_worker(rank, world, port, data_dir, workdir, ret, backend="gloo"):
    """what `torchrun ... run_skrec.py --recommender LightGCN` does on every rank"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SKR_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(workdir)
    import random
    from skrec import RunConfig
    from skrec.recommender.LightGCN import LightGCN
    from skrec.utils.py.random import reset_global_sampler
    reset_global_sampler(2020)
    np.random.seed(2021); random.seed(2021); torch.manual_seed(2021)
    rc = RunConfig(recommender="LightGCN", data_dir=data_dir, file_column="UIRT", sep="\t",
                   metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20), test_batch_size=16, seed=2021)
    m = LightGCN(rc, dict(lr=1e-3, reg=1e-3, embed_size=64, n_layers=3, adj_type="pre", batch_size=256, epochs=2))
    assert m.dist.world == world and m.engine is not None
    reports, losses = [], []
    ev, te = m.evaluate, m.train_epoch

    def evaluate(test_users=None):
        r = ev(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        return r

    def train_epoch(it):
        te(it)
        losses.append(m.step_losses.cpu().numpy().copy())
    m.evaluate, m.train_epoch = evaluate, train_epoch
    m.fit()
    ret[rank] = dict(reports=np.stack(reports), losses=np.concatenate(losses, 0), U1=m.user_embeddings.cpu().numpy(),
                     V1=m.item_embeddings.cpu().numpy())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_lightgcn_fit_under_torchrun_contract(tiny_dir, tmp_path):
    """LightGCN.fit() through the drop-in API on two ranks == the reference's single-process run"""
    g = np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_api_worker, args=(world, _free_port(), tiny_dir, str(tmp_path), ret), nprocs=world, join=True)
        res = {k: ret[k] for k in range(world)}
    for r in res.values():
        np.testing.assert_allclose(r["losses"][:, 0], g["bpr_mean"], rtol=1e-5)
        np.testing.assert_allclose(r["losses"][:, 1], g["l2"], rtol=1e-5)
        np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)
    assert np.array_equal(res[0]["reports"], res[1]["reports"])


def _bprmf_worker(rank, world, port, data_dir, workdir, exchange, ret, backend="gloo"):
    exchange, _, adam_block = exchange.partition("/")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SKR_DIST_BACKEND=backend, SKR_EXCHANGE=exchange, SKR_ADAM_BLOCK=adam_block or "8",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(workdir)
    import random
    from skrec import RunConfig
    from skrec.recommender.BPRMF import BPRMF
    from skrec.utils.py.random import reset_global_sampler
    reset_global_sampler(2020)
    np.random.seed(2021); random.seed(2021); torch.manual_seed(2021)
    rc = RunConfig(recommender="BPRMF", data_dir=data_dir, file_column="UIRT", sep="\t",
                   metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20), test_batch_size=16, seed=2021)
    m = BPRMF(rc, dict(lr=1e-3, reg=1e-3, n_dim=64, batch_size=256, epochs=3))
    assert m.engine is not None and m.engine.n_local in (64, 32, 22, 21)
    reports, losses = [], []
    ev, te = m.evaluate, m.train_epoch

    def evaluate(test_users=None):
        r = ev(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        return r

    def train_epoch(it):
        te(it)
        losses.append(m.step_losses.cpu().numpy().copy())
    m.evaluate, m.train_epoch = evaluate, train_epoch
    m.fit()
    ret[rank] = dict(reports=np.stack(reports), losses=np.concatenate(losses, 0),
                     U1=m.engine.gather_user_table().cpu().numpy(), V1=m.engine.item_rows.cpu().numpy(),
                     b1=m.engine.item_bias.cpu().numpy(), pred=m.predict([0, 3, 9, 63]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "dense"), (2, "sparse"), (3, "sparse"), (3, "auto"), (2, "sparse/1"), (2, "dense/3")])
def test_bprmf_fit_under_torchrun_contract(world, exchange, tiny_dir, tmp_path):
    """BPRMF.fit() on N ranks (global batch 256 split by user ownership; item gradients summed either by a
    dense all-reduce or by the packed-row all-gather; Adam blocked over 8 / 3 batches or stepped per batch ("/1"))
    == the reference's single-process run"""
    g = np.load(os.path.join(GOLDEN, "golden_bprmf.npz"))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bprmf_worker, args=(world, _free_port(), tiny_dir, str(tmp_path), exchange, ret), nprocs=world, join=True)
        res = {k: ret[k] for k in range(world)}
    for r in res.values():
        np.testing.assert_allclose(r["losses"][:, 0], g["bpr_sum"], rtol=1e-5)
        np.testing.assert_allclose(r["losses"][:, 1], g["l2"], rtol=1e-5)
        np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(r["b1"], g["b1"].reshape(-1), rtol=0, atol=2e-6)
        np.testing.assert_allclose(r["pred"], g["pred"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(res[0]["V1"], res[1]["V1"]) and np.array_equal(res[0]["reports"], res[world - 1]["reports"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL branch needs two GPUs (one process per GPU); the test boxes of this "
                    "environment have one -- it runs as soon as a multi-GPU box executes the suite")
@pytest.mark.parametrize("exchange", ["sparse/32", "dense/1"])
def test_bprmf_fit_on_rccl(exchange, tiny_dir, tmp_path):
    """the same two-rank fit() with backend "nccl" (= RCCL): init_process_group(device_id=...), all_gather_into_tensor,
    the collectives beside the side-stream cold pass"""
    g = np.load(os.path.join(GOLDEN, "golden_bprmf.npz"))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bprmf_worker, args=(2, _free_port(), tiny_dir, str(tmp_path), exchange, ret, "nccl"), nprocs=2, join=True)
        res = {k: ret[k] for k in range(2)}
    for r in res.values():
        np.testing.assert_allclose(r["losses"][:, 0], g["bpr_sum"], rtol=1e-5)
        np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=2e-6)
    assert np.array_equal(res[0]["V1"], res[1]["V1"])


_TWO_GPUS = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL branch needs two GPUs (one process per GPU); the "
                               "test boxes of this environment have one -- it runs as soon as a multi-GPU box executes the suite")


@_TWO_GPUS
@pytest.mark.parametrize("plan", ["auto", "1"])
def test_lightgcn_fit_on_rccl(plan, tiny_dir, tmp_path, monkeypatch):
    """ShardedLightGCN through the drop-in API on two RCCL ranks (the asynchronous all-reduce of the item block beside the
    user-side product, the compact exchanges of the masked layers with the plan forced) == the reference's recorded run"""
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)
    g = np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_api_worker, args=(2, _free_port(), tiny_dir, str(tmp_path), ret, "nccl"), nprocs=2, join=True)
        res = {k: ret[k] for k in range(2)}
    for r in res.values():
        np.testing.assert_allclose(r["losses"][:, 0], g["bpr_mean"], rtol=1e-5)
        np.testing.assert_allclose(r["losses"][:, 1], g["l2"], rtol=1e-5)
        np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)
    assert np.array_equal(res[0]["V1"], res[1]["V1"]) and np.array_equal(res[0]["reports"], res[1]["reports"])


def _layergcn_worker(rank, world, port, data_dir, workdir, dropout, ret, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SKR_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(workdir)
    import random
    from skrec import RunConfig
    from skrec.recommender.LayerGCN import LayerGCN
    from skrec.utils.py.random import reset_global_sampler
    reset_global_sampler(2020)
    np.random.seed(2021); random.seed(2021); torch.manual_seed(2021)
    rc = RunConfig(recommender="LayerGCN", data_dir=data_dir, file_column="UIRT", sep="\t",
                   metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20), test_batch_size=16, seed=2021)
    m = LayerGCN(rc, dict(lr=1e-3, reg=1e-2, embed_dim=64, n_layers=4, dropout=dropout, batch_size=256, epochs=2))
    assert m.engine is not None and m.dist.world == world
    reports, losses, nnz = [], [], []
    ev, te = m.evaluate, m.train_epoch

    def evaluate(test_users=None):
        r = ev(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        return r

    def train_epoch(it):
        nnz.append(m.engine.train_blocks[0].nnz)
        te(it)
        losses.append(m.step_losses.cpu().numpy().copy())
    m.evaluate, m.train_epoch = evaluate, train_epoch
    m.fit()
    e = m.engine
    ret[rank] = dict(reports=np.stack(reports), losses=np.concatenate(losses, 0), U1=m.user_embeddings.cpu().numpy(),
                     V1=m.item_embeddings.cpu().numpy(), Uf=e.gather_user_rows(e.whole_out()[:e.n_local]).cpu().numpy(),
                     pred=m.predict([0, 3, 9, 63]), n_test=len(m.evaluator.user_pos_test), nnz=nnz,
                     full_nnz=e.full_blocks[0].nnz)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,plan", [(2, "auto"), (2, "1"), (3, "1")])
def test_layergcn_fit_under_torchrun_contract(world, plan, tiny_dir, tmp_path, monkeypatch):
    """LayerGCN.fit() on N ranks (user-sharded rows, replicated item rows, one exchange per layer and direction beside the
    user-side product, compact in the masked layer) == the reference's single-process run; plan "1" forces the SpMM plan
    (masks honoured) on the tiny graph"""
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)
    g = np.load(os.path.join(GOLDEN, "golden_layergcn.npz"))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_layergcn_worker, args=(world, _free_port(), tiny_dir, str(tmp_path), 0.0, ret), nprocs=world, join=True)
        res = {k: ret[k] for k in range(world)}
    for r in res.values():
        total = r["losses"][:, 0] + np.float32(1e-2) * r["losses"][:, 1]
        np.testing.assert_allclose(total, g["loss"], rtol=1e-5)
        np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)   # incl. the all-tied zero-degree user
        np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["Uf"], g["Uf"], rtol=0, atol=6e-6)
        np.testing.assert_allclose(r["pred"], g["pred"], rtol=1e-4, atol=2e-6)
    assert np.array_equal(res[0]["V1"], res[1]["V1"]) and np.array_equal(res[0]["reports"], res[world - 1]["reports"])


@_TWO_GPUS
@pytest.mark.parametrize("plan,dropout", [("auto", 0.0), ("1", 0.0), ("1", 0.25)])
def test_layergcn_fit_on_rccl(plan, dropout, tiny_dir, tmp_path, monkeypatch):
    """ShardedLayerGCN.fit() on two RCCL ranks == the reference's recorded run (dropout 0); with edge dropout the pruning draw
    of rank 0 is broadcast over RCCL and the replicas stay bit-identical"""
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)
    g = np.load(os.path.join(GOLDEN, "golden_layergcn.npz"))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_layergcn_worker, args=(2, _free_port(), tiny_dir, str(tmp_path), dropout, ret, "nccl"), nprocs=2, join=True)
        res = {k: ret[k] for k in range(2)}
    if dropout == 0.0:
        for r in res.values():
            total = r["losses"][:, 0] + np.float32(1e-2) * r["losses"][:, 1]
            np.testing.assert_allclose(total, g["loss"], rtol=1e-5)
            np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)
            np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)
    else:
        assert np.isfinite(res[0]["losses"]).all()
    assert np.array_equal(res[0]["V1"], res[1]["V1"]) and np.array_equal(res[0]["reports"], res[1]["reports"])


@pytest.mark.parametrize("plan", ["auto", "1"])
def test_layergcn_sharded_edge_dropout(plan, tiny_dir, tmp_path, monkeypatch):
    """dropout > 0 on two ranks: every rank prunes the SAME edges (rank 0's draw), so the replicas of the
    item table stay bit-identical and the kept-edge counts add up to int(E * (1 - dropout))"""
    world = 2
    monkeypatch.setenv("SKR_SPMM_PLAN", plan)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_layergcn_worker, args=(world, _free_port(), tiny_dir, str(tmp_path), 0.25, ret), nprocs=world, join=True)
        res = {k: ret[k] for k in range(world)}
    n_edges = sum(r["full_nnz"] for r in res.values())
    for ep in range(2):
        assert sum(r["nnz"][ep] for r in res.values()) == int(n_edges * 0.75)
    assert np.array_equal(res[0]["V1"], res[1]["V1"]) and np.array_equal(res[0]["reports"], res[1]["reports"])
    assert np.isfinite(res[0]["losses"]).all()


@pytest.mark.parametrize("model", ["bprmf-sparse", "bprmf-dense", "lightgcn", "layergcn"])
def test_fit_on_a_single_rank_rccl_group(model, tiny_dir, tmp_path, monkeypatch):
    """The boxes of this environment have one GPU, so the multi-rank tests above run on gloo.  This one runs the SAME sharded
    engines on backend "nccl" (= RCCL) with a group of ONE rank forced through every collective (SKR_DIST_FORCE_ACTIVE=1):
    init_process_group(device_id=...), all_gather_into_tensor, all_reduce(async_op=True) + wait beside the side-stream cold
    pass, broadcast, barrier -- the calls an N-GPU run makes, against the reference's recorded trajectories."""
    monkeypatch.setenv("SKR_DIST_FORCE_ACTIVE", "1")
    monkeypatch.setenv("SKR_SPMM_PLAN", "1")
    with mp.Manager() as mgr:
        ret = mgr.dict()
        if model.startswith("bprmf"):
            g = np.load(os.path.join(GOLDEN, "golden_bprmf.npz"))
            mp.spawn(_bprmf_worker, args=(1, _free_port(), tiny_dir, str(tmp_path), model.split("-")[1] + "/32", ret, "nccl"), nprocs=1, join=True)
            r = ret[0]
            np.testing.assert_allclose(r["losses"][:, 0], g["bpr_sum"], rtol=1e-5)
            np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(r["U1"], g["U1"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)
        elif model == "lightgcn":
            g = np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))
            mp.spawn(_api_worker, args=(1, _free_port(), tiny_dir, str(tmp_path), ret, "nccl"), nprocs=1, join=True)
            r = ret[0]
            np.testing.assert_allclose(r["losses"][:, 0], g["bpr_mean"], rtol=1e-5)
            np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)
            np.testing.assert_allclose(r["reports"], g["reports"], rtol=1e-5, atol=2e-4)
        else:
            g = np.load(os.path.join(GOLDEN, "golden_layergcn.npz"))
            mp.spawn(_layergcn_worker, args=(1, _free_port(), tiny_dir, str(tmp_path), 0.0, ret, "nccl"), nprocs=1, join=True)
            r = ret[0]
            total = r["losses"][:, 0] + np.float32(1e-2) * r["losses"][:, 1]
            np.testing.assert_allclose(total, g["loss"], rtol=1e-5)
            np.testing.assert_allclose(r["V1"], g["V1"], rtol=0, atol=3e-6)


def _order_worker(rank, world, port, model, ret):
    """a few steps of a sharded engine on batches whose rows are all distinct (every gradient row is written once, so nothing
    depends on the order of float atomics), then an unmasked propagation"""
    import torch.distributed as dist
    from skrec.parallel import DistContext, ShardedLayerGCN, ShardedLightGCN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))
    U, I = g["U0"].shape[0], g["V0"].shape[0]
    ctx = DistContext(rank, world)
    if model == "lightgcn":
        adj = sp.csr_matrix((g["adj_val"], (g["adj_idx"][0], g["adj_idx"][1])), shape=(U + I, U + I))
        eng = ShardedLightGCN(ctx, adj, g["U0"], g["V0"], n_layers=3, lr=1e-3, reg=1e-3, batch_size_cfg=32)
    else:
        d = np.load(os.path.join(GOLDEN, "tiny_dataset.npz"))
        pairs = np.unique(d["train"][:, :2].astype(np.int64), axis=0)
        eng = ShardedLayerGCN(ctx, torch.from_numpy(pairs[:, 0].copy()), torch.from_numpy(pairs[:, 1].copy()), U, I,
                              g["U0"], g["V0"], n_layers=4, lr=1e-3, reg=1e-2)
    dev = eng.device
    rng = np.random.RandomState(5)
    for _ in range(4):
        u = rng.permutation(U)[:32].astype(np.int32)
        i = rng.permutation(I // 2)[:32].astype(np.int32)
        j = (I // 2 + rng.permutation(I // 2)[:32]).astype(np.int32)
        eng.train_step(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(j).to(dev))
    whole = eng.propagate()
    ret[rank] = dict(ego=eng.ego.cpu().numpy(), whole=whole.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model", ["lightgcn", "layergcn"])
def test_pipelined_exchange_order_is_bit_identical(model, monkeypatch):
    """several ranks: the next item-side partial product queued BEFORE the wait for the current layer's exchange (the default,
    skrec.parallel._pipelined) runs the same launches on the same operands as the layer-by-layer order (SKR_DIST_PIPELINE=0):
    same bits, forward and backward, masked layers and compact exchanges included.  (The first hop's scatter adds with
    float atomics, whose order is not fixed: it is switched off here, and the batches name every row once, so that both
    runs are deterministic; a repeat of the first run checks that they are.)"""
    monkeypatch.setenv("SKR_SPMM_PLAN", "1")
    monkeypatch.setenv("SKR_FIRST_HOP_SCATTER", "0")
    runs = []
    for flag in ("1", "1", "0"):
        monkeypatch.setenv("SKR_DIST_PIPELINE", flag)
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_order_worker, args=(2, _free_port(), model, ret), nprocs=2, join=True)
            runs.append({k: ret[k] for k in range(2)})
    for rank in range(2):
        for key in ("ego", "whole"):
            assert np.array_equal(runs[0][rank][key], runs[1][rank][key]), ("not deterministic", model, rank, key)
            assert np.array_equal(runs[0][rank][key], runs[2][rank][key]), (model, rank, key)
    assert np.isfinite(runs[0][0]["ego"]).all() and not np.array_equal(runs[0][0]["ego"][-96:], np.load(os.path.join(GOLDEN, "golden_lightgcn.npz"))["V0"])


def _width_worker(rank, world, port, model, dim, data_dir, workdir, ret):
    """fit() through the drop-in API at an embedding width below 64, on one process (the one-GPU path) or on `world` gloo ranks"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SKR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(workdir)
    import random
    from skrec import RunConfig
    from skrec.utils.py.random import reset_global_sampler
    reset_global_sampler(2020)
    np.random.seed(2021); random.seed(2021); torch.manual_seed(2021)
    rc = RunConfig(recommender=model, data_dir=data_dir, file_column="UIRT", sep="\t",
                   metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20), test_batch_size=16, seed=2021)
    if model == "BPRMF":
        from skrec.recommender.BPRMF import BPRMF
        m = BPRMF(rc, dict(lr=1e-3, reg=1e-3, n_dim=dim, batch_size=256, epochs=2))
    elif model == "LightGCN":
        from skrec.recommender.LightGCN import LightGCN
        m = LightGCN(rc, dict(lr=1e-3, reg=1e-3, embed_size=dim, n_layers=3, adj_type="pre", batch_size=256, epochs=2))
    else:
        from skrec.recommender.LayerGCN import LayerGCN
        m = LayerGCN(rc, dict(lr=1e-3, reg=1e-2, embed_dim=dim, n_layers=4, dropout=0.0, batch_size=256, epochs=2))
    assert (m.engine is not None) == (world > 1)
    reports, losses = [], []
    ev, te = m.evaluate, m.train_epoch

    def evaluate(test_users=None):
        r = ev(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        return r

    def train_epoch(it):
        te(it)
        losses.append(m.step_losses.cpu().numpy().copy())
    m.evaluate, m.train_epoch = evaluate, train_epoch
    m.fit()
    if model == "BPRMF" and world > 1:
        U1, V1 = m.engine.gather_user_table()[:, :dim], m.engine.item_rows[:, :dim]
        assert float(m.engine.item_rows[:, dim:].abs().max()) == 0.0      # the padding stays zero
    else:
        U1, V1 = m.user_embeddings, m.item_embeddings
    assert U1.shape[1] == dim and V1.shape[1] == dim
    pred = m.predict([0, 3, 9, 63]) if model != "LightGCN" else np.zeros(1, np.float32)   # (LightGCN.predict needs eval mode)
    ret[rank] = dict(reports=np.stack(reports), losses=np.concatenate(losses, 0), U1=U1.cpu().numpy(), V1=V1.cpu().numpy(), pred=pred)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("model,dim", [("BPRMF", 32), ("LightGCN", 50), ("LayerGCN", 32)])
def test_sharded_fit_at_a_narrower_width_equals_the_one_gpu_fit(model, dim, tiny_dir, tmp_path, monkeypatch):
    """n_dim / embed_size / embed_dim below 64 on several ranks: zero-padded 64-float rows in the sharded engines, as on one GPU
    (whose fit at these widths replays the oracle: tests/test_gpu_config0.py) -- same losses, reports and tables"""
    monkeypatch.setenv("SKR_SPMM_PLAN", "1")
    res = {}
    for world in (1, 2):
        wd = tmp_path / str(world)
        wd.mkdir()
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_width_worker, args=(world, _free_port(), model, dim, tiny_dir, str(wd), ret), nprocs=world, join=True)
            res[world] = {k: ret[k] for k in range(world)}
    one = res[1][0]
    for r in res[2].values():
        np.testing.assert_allclose(r["losses"], one["losses"], rtol=1e-5)
        np.testing.assert_allclose(r["reports"], one["reports"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(r["U1"], one["U1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["V1"], one["V1"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(r["pred"], one["pred"], rtol=1e-4, atol=2e-6)
    assert np.array_equal(res[2][0]["V1"], res[2][1]["V1"])
