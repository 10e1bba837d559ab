"""LightGCN step at the size of BASELINE configs[2] on one MI355X, two layouts of the same propagation: the bipartite row
blocks of skrec.parallel.ShardedLightGCN (two products per layer: item side + user side; what N > 1 needs) and ONE product
per layer on the square [(U + I), (U + I)] adjacency (what skrec.recommender.LightGCN runs on one GPU).
usage: python tools/lightgcn_square_vs_blocks.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "scikit-recommender_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from skrec import _hip  # noqa: E402
from skrec.parallel import DistContext, ShardedLightGCN  # noqa: E402
from skrec.recommender.LightGCN import build_adjacency_device  # noqa: E402
from skrec.recommender.base import DenseAdam  # noqa: E402
from skrec.utils.py.random import DeviceSampler  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = _hip.require_gpu()
nU, nI, nE, b, D, NL = 1_000_000, 100_000, 50_000_000, 1024, 64, 3
full = bench.synth_dataset(nU, nI, nE, 20260101, dev)
g0 = torch.Generator().manual_seed(2021)
user0 = (torch.rand(nU, D, generator=g0) * 2 - 1) * (6.0 / (nU + D)) ** 0.5
item0 = (torch.rand(nI, D, generator=g0) * 2 - 1) * (6.0 / (nI + D)) ** 0.5
need = (K + 3) * b
end_user = int(torch.searchsorted(full["rowptr"], torch.tensor(need, device=dev))) + 1
nnz = int(full["rowptr"][end_user])
neg = torch.empty(nnz, dtype=torch.int32, device=dev)
DeviceSampler(2020).sample_epoch_exact(nI, end_user, full["rowptr"][:end_user + 1].contiguous(), full["items"][:nnz].contiguous(), nnz, 1, neg)
uu, ii, jj = _hip.shuffle_gather([full["users"][:nnz].contiguous(), full["items"][:nnz].contiguous(), neg], None, seed=5, n_out=(K + 3) * b)


def timed(step):
    for s in range(3):
        step(uu[s * b:(s + 1) * b], ii[s * b:(s + 1) * b], jj[s * b:(s + 1) * b])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(3, K + 3):
        step(uu[s * b:(s + 1) * b], ii[s * b:(s + 1) * b], jj[s * b:(s + 1) * b])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


eng = ShardedLightGCN.from_device_edges(DistContext(0, 1), full["users"], full["items"], nU, nI, user0, item0, NL, 1e-3, 1e-3, b)
print(f"bipartite row blocks (ShardedLightGCN, world 1): {timed(eng.train_step):.2f} ms per step; loss {eng.loss.tolist()}", flush=True)
del eng
torch.cuda.empty_cache()


class Square(object):
    """the step of skrec.recommender.LightGCN.train_step on tables built here (no dataset files)"""

    def __init__(self):
        self.adj, _ = build_adjacency_device(full["users"], full["items"], nU, nI, "pre", dev)
        N = nU + nI
        self.ego = torch.cat([user0, item0], 0).to(dev).contiguous()
        self.opt = DenseAdam(self.ego.view(-1), lr=1e-3)
        self.gE = self.opt.grad.view(N, D)
        z = lambda: torch.zeros((N, D), device=dev)  # noqa: E731
        self.final, self.x, self.gF, self.g = z(), [z(), z()], z(), [z(), z()]
        self.mask = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(2, device=dev)

    def step(self, u, i, j):
        L, st = _hip.lib(), _hip.stream()
        u, i, j = u.contiguous(), i.contiguous(), j.contiguous()
        N, scale = nU + nI, 1.0 / (NL + 1)
        _hip.check(L.skr_clear_marked_rows(_hip.ptr(self.mask), N, 1, _hip.ptr(self.gF), 64, st))
        for ids, off in ((u, 0), (i, nU), (j, nU)):
            _hip.check(L.skr_mark_ids(_hip.ptr(ids), ids.numel(), off, _hip.ptr(self.mask), st))
        x = self.ego
        for k in range(NL):
            y = self.x[k & 1]
            self.adj.spmm(x, y, accum=self.final, accum_scale=scale, accum_base=self.ego if k == 0 else None,
                          row_mask=self.mask if k == NL - 1 else None, accum_mask=self.mask)
            x = y
        self.loss.zero_()
        _hip.check(L.skr_bpr_step_dim(_hip.ptr(self.final[:nU]), _hip.ptr(self.final[nU:]), None, _hip.ptr(self.ego[:nU]), _hip.ptr(self.ego[nU:]),
                                      _hip.ptr(u), _hip.ptr(i), _hip.ptr(j), u.numel(), 64, 1.0 / u.numel(), 1e-3, 1.0 / b,
                                      _hip.ptr(self.gF[:nU]), _hip.ptr(self.gF[nU:]), None, _hip.ptr(self.gE[:nU]), _hip.ptr(self.gE[nU:]),
                                      _hip.ptr(self.loss), 1, None, None, scale, st))
        x = self.gF
        for k in range(NL):
            y = self.g[k & 1]
            self.adj.spmm(x, y, addend=self.gF, accum=self.gE if k == NL - 1 else None, accum_scale=1.0,
                          col_mask=self.mask if k == 0 else None, addend_mask=self.mask)
            x = y
        self.opt.step()


sq = Square()
print(f"square adjacency (one product per layer): {timed(sq.step):.2f} ms per step; loss {sq.loss.tolist()}; plan {sq.adj.plan_info()}", flush=True)
