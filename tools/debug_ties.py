import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
from oracle import oracle as O
from gpu_utils import eval_scores
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
B, I, K = 96, 100000, 100
sc = (rng.standard_normal((B, I)) * 0.1).astype(np.float32)
for b in range(B):
    top = np.argsort(-sc[b])[:130]
    for _ in range(rng.integers(1, 4)):
        i, j = rng.choice(130, 2, replace=False)
        sc[b, top[j]] = sc[b, top[i]]
    if b % 3 == 0:
        sc[b, rng.choice(I, 60, replace=False)] = -np.inf
_, ids, _ = eval_scores(sc, [[] for _ in range(B)], [2], K)
bad = 0
for b in range(B):
    want = O.topk_ids_heap(sc[b], K)
    if not np.array_equal(ids[b], want):
        bad += 1
        d = np.flatnonzero(ids[b] != want)
        print("row", b, "first diff at rank", d[0], "got", ids[b][d[:4]], "want", want[d[:4]], "range ok", ids[b].min() >= 0 and ids[b].max() < I)
print("bad rows", bad, "of", B)
