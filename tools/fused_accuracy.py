"""Accuracy of the fused evaluator's two arithmetic modes against float64: for random factors of several
scales, the error of the returned top-K scores relative to sum_i |u_i v_i| (the natural scale of a dot product's
rounding error).  Run once per mode: SKR_FUSED_MODE=fp32 | bf16x3."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from gpu_utils import fused_topk  # noqa: E402

rng = np.random.default_rng(0)
B, I, K = 512, 20000, 50
print("mode", os.environ.get("SKR_FUSED_MODE", "fp32"))
for scale in (1e-3, 0.1, 1.0, 30.0):
    U = (rng.standard_normal((B, 64)) * scale).astype(np.float32)
    V = (rng.standard_normal((I, 64)) * scale).astype(np.float32)
    b = (rng.standard_normal(I) * scale * scale).astype(np.float32)
    ids, sc = fused_topk(U, np.arange(B, dtype=np.int32), V, b, None, np.zeros(0, np.int32), K)
    U64, V64 = U.astype(np.float64), V.astype(np.float64)
    rel = []
    for r in range(B):
        exact = V64[ids[r]] @ U64[r] + b[ids[r]].astype(np.float64)
        denom = np.abs(V64[ids[r]]) @ np.abs(U64[r]) + np.abs(b[ids[r]])
        rel.append(np.abs(sc[r] - exact) / denom)
    rel = np.concatenate(rel)
    print(f"scale {scale:g}: max rel err {rel.max():.3e}  mean {rel.mean():.3e}  (2^-24 = {2.0 ** -24:.3e})")
