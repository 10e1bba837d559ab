"""The fp16x2 evaluator's absolute error floor, measured with its guard switched off (SKR_F7_GUARD=0, set before the library is
loaded): users of ordinary magnitude against a catalogue whose BEST items are 2^-e times smaller than its largest ones, so
that the returned scores come from elements whose low fp16 piece is a denormal.  Prints, per e, the largest error of the
returned scores in units of the floor the guard assumes (2^-3 / (s_u s_v)) -- it must stay below 1 -- and relative to
the scores themselves (what the guard protects the result from)."""
import os
import sys

os.environ["SKR_FUSED_MODE"] = "f16x2"
os.environ["SKR_F7_GUARD"] = "0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from gpu_utils import fused_topk  # noqa: E402

rng = np.random.default_rng(3)
B, I, K = 128, 2048, 16
U = np.abs(rng.standard_normal((B, 64)) * 0.3).astype(np.float32)


def scale_of(m):
    return 2.0 ** (14 - np.floor(np.log2(m)))


for e in (8, 12, 16, 18, 20, 22, 24):
    V = np.abs(rng.standard_normal((I, 64)) * 0.3).astype(np.float32)
    V[: I // 2] *= -1.0                                  # large items: every score negative
    V[I // 2:] *= np.float32(2.0 ** -e)                  # small items: positive scores -> they are the top-K
    ids, sc = fused_topk(U, np.arange(B, dtype=np.int32), V, None, None, np.zeros(0, np.int32), K)
    assert (ids >= I // 2).all()
    exact = np.einsum("bkd,bd->bk", V.astype(np.float64)[ids], U.astype(np.float64))
    S = scale_of(np.abs(U).max()) * scale_of(np.abs(V).max())
    err = np.abs(sc - exact)
    print(f"small items 2^-{e}: max |error| = {err.max() * S / 2.0 ** -3:.3g} floors, {(err / np.abs(exact)).max():.3g} of the score; "
          f"scores * S = 2^{np.log2(np.abs(exact).min() * S):.1f} .. 2^{np.log2(np.abs(exact).max() * S):.1f} (guard: 2^19)")
