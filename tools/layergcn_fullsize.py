"""LayerGCN at the size of BASELINE configs[2]: ms per training step on one MI355X through
skrec.parallel.ShardedLayerGCN (world 1) -- 4 layers, batch 2048 (the reference's defaults, LayerGCN.py:25-33), full-graph
propagation with the cosine layer refinement forward and backward per mini-batch, dense Adam.
usage: python tools/layergcn_fullsize.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "scikit-recommender_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from skrec import _hip  # noqa: E402
from skrec.parallel import DistContext, ShardedLayerGCN  # noqa: E402
from skrec.utils.py.random import DeviceSampler  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = _hip.require_gpu()
nU, nI, nE, b, D = 1_000_000, 100_000, 50_000_000, 2048, 64
full = bench.synth_dataset(nU, nI, nE, 20260101, dev)
ctx = DistContext(0, 1)
g0 = torch.Generator().manual_seed(2021)
user0 = (torch.rand(nU, D, generator=g0) * 2 - 1) * (6.0 / (nU + D)) ** 0.5
item0 = (torch.rand(nI, D, generator=g0) * 2 - 1) * (6.0 / (nI + D)) ** 0.5
eng = ShardedLayerGCN(ctx, full["users"].long(), full["items"].long(), nU, nI, user0, item0, 4, 1e-3, 1e-2, device=dev)
need = (K + 3) * b
end_user = int(torch.searchsorted(full["rowptr"], torch.tensor(need, device=dev))) + 1
nnz = int(full["rowptr"][end_user])
neg = torch.empty(nnz, dtype=torch.int32, device=dev)
DeviceSampler(2020).sample_epoch_exact(nI, end_user, full["rowptr"][:end_user + 1].contiguous(), full["items"][:nnz].contiguous(), nnz, 1, neg)
uu, ii, jj = _hip.shuffle_gather([full["users"][:nnz].contiguous(), full["items"][:nnz].contiguous(), neg], None, seed=5, n_out=(K + 3) * b)
for s in range(3):
    eng.train_step(uu[s * b:(s + 1) * b], ii[s * b:(s + 1) * b], jj[s * b:(s + 1) * b])
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(3, K + 3):
    eng.train_step(uu[s * b:(s + 1) * b], ii[s * b:(s + 1) * b], jj[s * b:(s + 1) * b])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"LayerGCN 4-layer d=64, {nU} users / {nI} items / {int(full['rowptr'][-1])} interactions, batch {b}: {dt * 1e3:.2f} ms per step, "
      f"{b / dt:.0f} interactions/s; loss {eng.loss.tolist()}")
