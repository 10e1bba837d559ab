"""CPU suite, part 4: the N > 1 path of bench.py on two, three and four gloo ranks -- user sharding u % N, the
re-indexing of the shard's CSR, and the one exchange step (all-reduce of the replicated item
table's gradient) reproduce the single-process gradients.  Kernels are replaced by the oracle's
explicit-gradient maths here (no GPU); the collective and the sharding code are the real ones."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy(seed=0, nU=40, nI=30):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 9, nU)
    rowptr = np.zeros(nU + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    items = np.concatenate([np.sort(rng.choice(nI, l, replace=False)) for l in lens]).astype(np.int32)
    users = np.repeat(np.arange(nU, dtype=np.int32), lens)
    return dict(rowptr=torch.from_numpy(rowptr), users=torch.from_numpy(users), items=torch.from_numpy(items),
                test_item=torch.arange(nU, dtype=torch.int32)), nU, nI


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full, nU, nI = _toy()
    ds, mine = bench.shard(full, rank, world, torch.device("cpu"))
    # shard bookkeeping: local row r is global user r*world + rank, rows keep their items
    assert torch.equal(mine.long(), torch.arange(rank, nU, world))
    g_rowptr, g_items = full["rowptr"].numpy(), full["items"].numpy()
    l_rowptr, l_items, l_users = ds["rowptr"].numpy(), ds["items"].numpy(), ds["users"].numpy()
    for r, u in enumerate(mine.tolist()):
        assert np.array_equal(l_items[l_rowptr[r]:l_rowptr[r + 1]], g_items[g_rowptr[u]:g_rowptr[u + 1]])
    assert np.array_equal(np.repeat(np.arange(len(mine)), np.diff(l_rowptr)), l_users)
    # one data-parallel step: local BPR gradients, all-reduce of the item part only
    rng = np.random.default_rng(1)
    U = (rng.standard_normal((nU, 64)) * 0.1).astype(np.float32)       # same on both ranks
    V = (rng.standard_normal((nI, 64)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(nI) * 0.1).astype(np.float32)
    Ul = U[mine.numpy()]
    neg = ((l_items + 7 + l_users) % nI).astype(np.int32)
    _, _, gU, gV, gb, _, _ = O.bpr_batch(Ul, V, b, Ul, V, l_users, l_items, neg, 1.0, 1e-3, 1.0)
    g_item = torch.from_numpy(np.concatenate([gV.reshape(-1), gb]))
    dist.all_reduce(g_item)                                             # the path's one exchange step
    out[rank] = (mine.numpy(), gU, g_item.numpy(), l_users, l_items, neg)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_step_equals_single_process(world):
    """two ranks, and -- users 40 do not divide by 3: ragged shards -- three and four"""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = {k: out[k] for k in range(world)}
    full, nU, nI = _toy()
    rng = np.random.default_rng(1)
    U = (rng.standard_normal((nU, 64)) * 0.1).astype(np.float32)
    V = (rng.standard_normal((nI, 64)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(nI) * 0.1).astype(np.float32)
    # the union of the two local batches, in global user ids
    gu = np.concatenate([res[r][0][res[r][3]] for r in range(world)]).astype(np.int32)
    gi = np.concatenate([res[r][4] for r in range(world)])
    gj = np.concatenate([res[r][5] for r in range(world)])
    _, _, gU, gV, gb, _, _ = O.bpr_batch(U, V, b, U, V, gu, gi, gj, 1.0, 1e-3, 1.0)
    want_item = np.concatenate([gV.reshape(-1), gb])
    for r in range(world):
        np.testing.assert_allclose(res[r][2], want_item, rtol=1e-5, atol=1e-7)      # replicas agree
        np.testing.assert_allclose(res[r][1], gU[res[r][0]], rtol=1e-5, atol=1e-7)  # user rows are local
    for r in range(1, world):
        assert np.array_equal(res[0][2], res[r][2])                                   # bit-identical replicas


def test_fast_sampler_twin_is_shard_invariant():
    """the slot-keyed sampler gives every (user, slot) the same negative however users are split"""
    from fast_sampler_twin import sample_fast
    from helpers import random_csr
    rng = np.random.default_rng(4)
    rowptr, pos = random_csr(rng, 60, 25, 0, 8)
    whole = sample_fast(9, 2, 0, 25, rowptr, pos, 2)
    cut_u = 23
    cut = int(rowptr[cut_u])
    lo = sample_fast(9, 2, 0, 25, rowptr[:cut_u + 1].copy(), pos[:cut].copy(), 2)
    hi = sample_fast(9, 2, cut * 2, 25, (rowptr[cut_u:] - cut).copy(), pos[cut:].copy(), 2)
    assert np.array_equal(np.concatenate([lo, hi]), whole)


def test_launch_command_is_the_drivers_form():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], 29411)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29411"
    assert cmd[-7] == os.path.abspath(bench.__file__) and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]


_STUB = """
import json, os, sys
import torch.distributed as dist
dist.init_process_group("gloo")          # env:// -- what torch.distributed.run exports
r, w = dist.get_rank(), dist.get_world_size()
import torch
t = torch.tensor([float(r + 1)])
dist.all_reduce(t)
if r == 0:
    print(json.dumps({"world": w, "sum": float(t), "argv": sys.argv[1:], "backend_env": os.environ.get("SKR_DIST_BACKEND"),
                      "master": os.environ["MASTER_ADDR"], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
dist.destroy_process_group()
sys.exit(int(os.environ.get("STUB_EXIT", "0")) if r == w - 1 else 0)
"""


def test_launcher_starts_fresh_ranks_and_relays_rank0(tmp_path, capfd):
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent starts two children under torch.distributed.run, relays rank
    0's line and returns the children's status (a stub script stands in for bench.py's body: no GPU here)."""
    import json
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SKR_DIST_BACKEND")}
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "3"], script=str(stub), env=env)
    out = capfd.readouterr().out
    assert rc == 0
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert line["world"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--gpus", "2", "--steps", "3"]
    assert line["master"] == "127.0.0.1" and line["ipc"] == "0"
    assert line["backend_env"] == "gloo"          # no GPU visible here: fewer devices than ranks => rehearsal backend
    env["STUB_EXIT"] = "3"
    assert bench.launch_ranks(2, [], script=str(stub), env=env) != 0      # a failing rank fails the launcher


def test_launcher_ends_ranks_that_never_finish(tmp_path, capfd):
    """a rank stuck in a collective must not hold the caller: past SKR_BENCH_RANKS_TIMEOUT the launcher ends the ranks' own
    process group and reports 124"""
    import time
    stub = tmp_path / "hang.py"
    stub.write_text("import time\ntime.sleep(600)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SKR_DIST_BACKEND")}
    env["SKR_BENCH_RANKS_TIMEOUT"] = "8"
    t0 = time.time()
    assert bench.launch_ranks(2, [], script=str(stub), env=env) == 124
    assert time.time() - t0 < 60
    assert "did not finish" in capfd.readouterr().err
