"""Repeat the full-size top-100 fused evaluation until a list disagrees with the dense path, then describe the
disagreement (which user, which item, where in the sweep, train-row length)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from skrec import _hip  # noqa: E402

U, I, K, nu = 1_000_000, 100_000, 100, 262_144
dev = torch.device("cuda", 0)
big = bench.synth_dataset(U, I, 50_000_000, 20260101, dev)
L = _hip.lib()
g = torch.Generator(device=dev).manual_seed(1)
Ut = torch.randn(U, 64, generator=g, device=dev) * 0.1
Vt = torch.randn(I, 64, generator=g, device=dev) * 0.1
bias = torch.randn(I, generator=g, device=dev) * 0.05
rowptr, items = big["rowptr"], big["items"]
users = torch.arange(nu, dtype=torch.int32, device=dev)
ws = int(L.skr_eval_fused_workspace(nu, K))
work = torch.empty(ws, dtype=torch.uint8, device=dev)
ids = torch.empty((nu, K), dtype=torch.int32, device=dev)
sc = torch.empty((nu, K), dtype=torch.float32, device=dev)
_hip.check(L.skr_eval_fused_topk(_hip.ptr(Ut), _hip.ptr(users), nu, _hip.ptr(Vt), _hip.ptr(bias), I, 64, _hip.ptr(rowptr),
                                 _hip.ptr(items), K, _hip.ptr(ids), _hip.ptr(sc), _hip.ptr(work), ws, _hip.stream()))
torch.cuda.synchronize()
n_bad = 0
CH = 4096
for s0 in range(0, nu, CH):
    us = users[s0:s0 + CH].contiguous()
    dense = _hip.score_matrix(Ut, us.cpu().numpy(), Vt, bias)
    _hip.check(L.skr_mask_train(_hip.ptr(dense), CH, I, I, _hip.ptr(us), _hip.ptr(rowptr), _hip.ptr(items), _hip.stream()))
    ids2 = torch.empty((CH, K), dtype=torch.int32, device=dev)
    _hip.check(L.skr_eval_scores(_hip.ptr(dense), CH, I, I, None, None, None, 0, K, None, _hip.ptr(ids2), None, _hip.stream()))
    torch.cuda.synchronize()
    a_sorted = torch.sort(ids[s0:s0 + CH], dim=1).values
    b_sorted = torch.sort(ids2, dim=1).values
    bad = torch.nonzero((a_sorted != b_sorted).any(1)).reshape(-1)
    for r in bad.tolist():
        u = s0 + r
        fa, fb = set(ids[u].cpu().tolist()), set(ids2[r].cpu().tolist())
        only_dense, only_fused = sorted(fb - fa), sorted(fa - fb)
        kth = float(sc[u, -1])
        dsc = [float(dense[r, m]) for m in only_dense]
        fsc = [float(dense[r, m]) for m in only_fused]
        gap = max(dsc) - kth if dsc else 0.0
        if gap > 5e-6:      # more than fp32 summation noise: a real miss
            n_bad += 1
            ln = int(rowptr[u + 1] - rowptr[u])
            print(f"user {u} wave {u // 64} lane {u % 64} train_len {ln}: dense-only {only_dense} (tiles {[m // 32 for m in only_dense]}, "
                  f"scores {dsc}) fused-only {only_fused} (scores {fsc}); fused K-th score {kth:.6f}")
print("users with a real miss:", n_bad)
