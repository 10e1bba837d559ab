"""Secondary kernels at BASELINE scale (1 M users / 100 k items / ~48 M interactions), one GPU:
exact + fast epoch sampling, CSR SpMM on the LightGCN adjacency, a LightGCN training step, the
score-matrix top-K kernel.  Prints one JSON object; numbers feed DESIGN.md section 4."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from skrec import _hip  # noqa: E402
from skrec.utils.py.random import DeviceSampler  # noqa: E402

dev = torch.device("cuda", 0)
L = _hip.lib()
out = {}


def timeit(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


U, I = 1_000_000, 100_000
ds = bench.synth_dataset(U, I, 50_000_000, 20260101, dev)
nnz = int(ds["rowptr"][-1])
neg = torch.empty(nnz, dtype=torch.int32, device=dev)
s = DeviceSampler(2020)
t = timeit(lambda: s.sample_epoch_exact(I, U, ds["rowptr"], ds["items"], nnz, 1, neg), n=2)
out["sample_epoch_exact"] = {"negatives": nnz, "seconds": t, "M_per_s": nnz / t / 1e6, "GBps_algorithmic": nnz * 8 / t / 1e9}
t = timeit(lambda: _hip.check(L.skr_sample_epoch_fast(1, 0, 0, I, U, _hip.ptr(ds["rowptr"]), _hip.ptr(ds["items"]), nnz, 1,
                                                      _hip.ptr(neg), _hip.stream())), n=5)
out["sample_epoch_fast"] = {"negatives": nnz, "seconds": t, "M_per_s": nnz / t / 1e6, "GBps_algorithmic": nnz * 8 / t / 1e9}

# LightGCN adjacency 'pre' (D^-1/2 A D^-1/2) built on the device
N = U + I
uu, ii = ds["users"].long(), ds["items"].long() + U
rows = torch.cat([uu, ii])
cols = torch.cat([ii, uu])
key = torch.sort(rows * N + cols).values
rows, cols = key // N, (key % N).int()
deg = torch.bincount(rows, minlength=N).float()
dinv = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
val = (dinv[rows] * dinv[cols.long()]).contiguous()
rowptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=N), 0)
A_nnz = int(rowptr[-1])
X = torch.randn(N, 64, device=dev) * 0.1
Y = torch.empty_like(X)
acc = torch.zeros_like(X)


def spmm(x, y, add=None, ac=None, sc=1.0):
    _hip.check(L.skr_csr_spmm(N, _hip.ptr(rowptr), _hip.ptr(cols), _hip.ptr(val), _hip.ptr(x), 64, A_nnz, _hip.ptr(add),
                              _hip.ptr(y), _hip.ptr(ac), sc, _hip.stream()))


t = timeit(lambda: spmm(X, Y), n=5)
alg = A_nnz * 8 + (N + 1) * 8 + 2 * N * 256
out["csr_spmm"] = {"rows": N, "nnz": A_nnz, "seconds": t, "GBps_algorithmic": alg / t / 1e9, "algorithmic_bytes": alg,
                   "gather_bytes_upper": A_nnz * 256, "GFLOPs": 2 * A_nnz * 64 / t / 1e9}
t = timeit(lambda: spmm(X, Y, add=X, ac=acc, sc=0.25), n=5)
out["csr_spmm_fused_epilogue"] = {"seconds": t}

# score-matrix top-K (drop-in for cpp_evaluate_matrix): 4096 x 100 000
B = 4096
sc = torch.randn(B, I, device=dev)
tp = torch.arange(B + 1, dtype=torch.long, device=dev)
ti = torch.randint(0, I, (B,), dtype=torch.int32, device=dev)
rows_o = torch.empty((B, 20), device=dev)
m = _hip.metric_array([2, 4])
t = timeit(lambda: _hip.check(L.skr_eval_scores(_hip.ptr(sc), B, I, I, _hip.ptr(tp), _hip.ptr(ti), m, 2, 10, _hip.ptr(rows_o),
                                                None, None, _hip.stream())), n=5)
out["eval_scores_topk_rows"] = {"users": B, "seconds": t, "users_per_s": B / t, "GBps_algorithmic": B * I * 4 / t / 1e9}
print(json.dumps(out))
