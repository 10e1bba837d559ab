"""Randomised cross-check of the fused evaluator's default arithmetic (f16x2 behind its guard, rejected rows through bf16x3)
against float64: shapes, top_k, common scales over seven decades, rows and elements of mixed magnitude, bias, train masks.
For every case: the returned scores within the fp32 chain's own noise of the float64 scores of the returned ids (relative to
sum |u_i v_i| + |bias|), and the id lists equal to float64's wherever every rank gap of the list is clear of that noise.
usage: python tools/f16x2_stress.py [cases] [seed]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from gpu_utils import fused_topk  # noqa: E402
from skrec import _hip  # noqa: E402


def run(n_cases=120, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    worst, n_rej_total, n_rows, n_id_rows = 0.0, 0, 0, 0
    for case in range(n_cases):
        K = int(rng.choice([1, 5, 10, 20, 50, 100, 128]))
        B = int(rng.integers(1, 300))
        max_tr = int(rng.choice([0, 0, 20, 120]))
        I = int(rng.integers(K + max_tr + 2, 20000))
        scale_u, scale_v = 10.0 ** rng.uniform(-4, 3, 2)
        spread = float(rng.choice([0.0, 0.0, 1.0, 3.0]))          # log-normal spread of ROW magnitudes
        espread = float(rng.choice([0.0, 0.0, 0.0, 2.0]))         # ... and of single elements
        nU = B + 7
        U = rng.standard_normal((nU, 64)) * scale_u * np.exp(rng.standard_normal((nU, 1)) * spread) * np.exp(rng.standard_normal((nU, 64)) * espread)
        V = rng.standard_normal((I, 64)) * scale_v * np.exp(rng.standard_normal((I, 1)) * spread) * np.exp(rng.standard_normal((I, 64)) * espread)
        U, V = U.astype(np.float32), V.astype(np.float32)
        bias = (rng.standard_normal(I) * scale_u * scale_v).astype(np.float32) if rng.random() < 0.5 else None
        users = rng.permutation(nU)[:B].astype(np.int32)
        if max_tr:
            lens = rng.integers(0, max_tr + 1, nU)
            rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            items = np.concatenate([np.sort(rng.choice(I, n, replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        else:
            rowptr, items = None, np.zeros(0, np.int32)
        ids, sc = fused_topk(U, users, V, bias, rowptr, items, K)
        n_rej = ctypes.c_int32(-1)
        _hip.check(_hip.lib().skr_eval_fused_rejected(ctypes.byref(n_rej), _hip.stream()))
        U64, V64 = U.astype(np.float64), V.astype(np.float64)
        full = U64[users] @ V64.T + (bias.astype(np.float64) if bias is not None else 0.0)
        mag = np.abs(U64[users]) @ np.abs(V64).T + (np.abs(bias.astype(np.float64)) if bias is not None else 0.0)
        if rowptr is not None:
            for r, u in enumerate(users):
                full[r, items[rowptr[u]:rowptr[u + 1]]] = -np.inf
        exact = np.take_along_axis(full, ids.astype(np.int64), 1)
        denom = np.take_along_axis(mag, ids.astype(np.int64), 1)
        assert np.isfinite(exact).all(), f"case {case}: a masked item was returned"
        err = float((np.abs(sc - exact) / denom).max())
        worst = max(worst, err)
        assert err < 1e-6, f"case {case}: score error {err:.3e} (K={K} B={B} I={I} scales {scale_u:.2g}/{scale_v:.2g} spread {spread}/{espread})"
        order = np.argsort(-full, axis=1, kind="stable")[:, :K + 1]
        top = np.take_along_axis(full, order, 1)
        noise = 2e-6 * np.take_along_axis(mag, order, 1).max(axis=1, keepdims=True)       # generous: several times the fp32 chain's noise
        clear = (-np.diff(top, axis=1) > noise).all(axis=1)
        assert np.array_equal(ids[clear], order[clear, :K]), f"case {case}: id lists differ where every gap is clear"
        for r in range(B):
            assert len(set(ids[r])) == K
        n_rej_total += n_rej.value
        n_rows += B
        n_id_rows += int(clear.sum())
        if verbose:
            print(f"case {case:3d}: K={K:3d} B={B:3d} I={I:5d} scales {scale_u:8.2g} {scale_v:8.2g} spread {spread:.0f}/{espread:.0f} bias {bias is not None!s:5} "
                  f"mask {max_tr:3d}: max err {err:.2e}, rejected {n_rej.value:3d}/{B}, ids checked on {int(clear.sum())} rows")
    print(f"{n_cases} cases, {n_rows} rows ({n_id_rows} with every gap clear: ids identical to float64's), worst score error {worst:.2e} of sum|u v|, "
          f"{n_rej_total} rows went through the bf16x3 kernel")
    return n_rows, n_id_rows, worst, n_rej_total


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
