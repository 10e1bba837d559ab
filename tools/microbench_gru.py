"""GRU4RecPlus at BASELINE configs[4] shape (d = 128, 100 k items, batch 128 + 2048 samples; inference over
1 M histories of length 50), one GPU.  Prints one JSON object; numbers feed DESIGN.md section 7."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from skrec.recommender.GRU4RecPlus import SessionGRU  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
out = {}
for d in (64, 128):
    I, b, S = 100_000, 128, 2048
    E_in = rng.normal(0, 0.01, (I, d)).astype(np.float32)
    E_out = rng.normal(0, 0.01, (I, d)).astype(np.float32)
    lim = np.sqrt(6.0 / (4 * d))
    cell = (rng.uniform(-lim, lim, (2 * d, 2 * d)).astype(np.float32), np.ones(2 * d, np.float32),
            rng.uniform(-lim, lim, (2 * d, d)).astype(np.float32), np.zeros(d, np.float32))
    net = SessionGRU(E_in, [cell], E_out, np.zeros(I, np.float32))
    pop = np.cumsum(rng.random(I) ** 3)
    pop /= pop[-1]
    items = torch.randint(0, I, (400, b), dtype=torch.int32, device=dev)
    state = net.zero_states(b)

    def step(k):
        global state
        neg = torch.from_numpy(np.searchsorted(pop, np.random.rand(S)).astype(np.int32)).to(dev)   # as GRU4RecPlus.fit does
        state = net.train_step(items[k], torch.cat([items[k + 1], neg]), state)
    for k in range(20):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for k in range(20, 20 + n):
        step(k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    params = net.flat.numel()
    out[f"train_step_d{d}"] = {"ms": t * 1e3, "events_per_s": b / t, "params": params,
                               "adam_GB_per_step": params * 24 / 1e9}
    # inference sweep: all users at once, one GRU step per history position
    U, T = 1_000_000, 50
    rowptr = torch.arange(0, (U + 1) * T, T, dtype=torch.long, device=dev)
    hist = torch.randint(0, I, (U * T,), dtype=torch.int32, device=dev)
    net.user_embeddings(rowptr, hist, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.user_embeddings(rowptr, hist, T)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    flops = 2.0 * U * T * (2 * d) * (3 * d)
    out[f"user_embeddings_d{d}"] = {"users": U, "history": T, "seconds": t, "users_per_s": U / t, "TFLOPs": flops / t / 1e12}
    del net
    torch.cuda.empty_cache()
print(json.dumps(out))
