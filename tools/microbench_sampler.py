"""exact-stream epoch at BASELINE size (1 M users / 100 k items / ~48 M negatives): both paths, seconds per epoch"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from skrec.utils.py.random import DeviceSampler  # noqa: E402

dev = torch.device("cuda", 0)
U, I = 1_000_000, 100_000
ds = bench.synth_dataset(U, I, 50_000_000, 20260101, dev)
nnz = int(ds["rowptr"][-1])
neg = torch.empty(nnz, dtype=torch.int32, device=dev)
out = {}
ref = None
for path in ("slab", "serial"):
    os.environ["SKR_EXACT_PATH"] = path
    s = DeviceSampler(2020)
    s.sample_epoch_exact(I, U, ds["rowptr"], ds["items"], nnz, 1, neg)
    torch.cuda.synchronize()
    first = neg.clone()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        s.sample_epoch_exact(I, U, ds["rowptr"], ds["items"], nnz, 1, neg)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[path] = {"seconds": min(ts), "M_per_s": nnz / min(ts) / 1e6, "last_epoch": s.last_epoch(), "draws": s.draws}
    if ref is None:
        ref = first
    else:
        out["paths_identical_first_epoch"] = bool(torch.equal(ref, first))
    # a small call, the size of bench.py's timed slice at the driver's K = 20 (25 k negatives)
    n_small = 600
    nnz_s = int(ds["rowptr"][n_small])
    rp = ds["rowptr"][:n_small + 1].contiguous()
    small = torch.empty(nnz_s, dtype=torch.int32, device=dev)
    s.sample_epoch_exact(I, n_small, rp, ds["items"][:nnz_s], nnz_s, 1, small)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s.sample_epoch_exact(I, n_small, rp, ds["items"][:nnz_s], nnz_s, 1, small)
    torch.cuda.synchronize()
    out[path]["small_call_us"] = (time.perf_counter() - t0) / 20 * 1e6
    out[path]["small_call_negatives"] = nnz_s
print(json.dumps(out))
