"""Where the time of skr_bpr_fused_step goes: k launches of a block timed with HIP events, nothing beside them.
SKR_FUSED_DBG (one-wavefront kernel only; results are then wrong, timing only): 1 no catch-up arithmetic, 2 no gradient
atomics, 4 no owner stores, 8 no loss atomics. 
usage: python tools/fused_lab.py [k] [blocks]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scikit-recommender_amd"))
from skrec import _hip  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nU, nI, b = 1_000_000, 100_000, 1024
dev = torch.device("cuda:0")
L, st = _hip.lib(), _hip.stream()
n_par = (nU + nI) * 64 + nI
g = torch.Generator(device=dev).manual_seed(1)
flat = torch.randn(n_par, device=dev, generator=g) * 0.01
m1, m2 = torch.zeros_like(flat), torch.zeros_like(flat)
n = k * b * n_blocks
u = torch.randint(0, nU, (n,), device=dev, generator=g, dtype=torch.int32)
w = 1.0 / (torch.arange(nI, device=dev, dtype=torch.float64) + 10.0) ** 0.8
i = torch.multinomial(w.float(), n, replacement=True, generator=g).int()
j = torch.randint(0, nI, (n,), device=dev, generator=g, dtype=torch.int32)
cap = k * 5 * b
work = torch.zeros(9 * cap * 64, device=dev)
meta, sb, sf = (torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3))
ns = torch.zeros(1, dtype=torch.int32, device=dev)
nfb = (n_par + 63) // 64
scratch = torch.zeros(28 * nfb // 8 + 1, dtype=torch.int64, device=dev)
loss = torch.zeros(64, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
t_plan, t_steps, t_end = [], [], []
t = 0
for blk in range(n_blocks):
    o = 4 * blk * k * b
    e = [ev() for _ in range(4)]
    e[0].record()
    _hip.check(L.skr_bpr_fused_plan(u.data_ptr() + o, i.data_ptr() + o, j.data_ptr() + o, b, k, 0, nU, nU + nI, nfb, scratch.data_ptr(),
                                    meta.data_ptr(), sb.data_ptr(), sf.data_ptr(), ns.data_ptr(), st))
    e[1].record()
    rc = 0
    for s in range(k):
        q = o + 4 * s * b
        rc |= L.skr_bpr_fused_step(flat.data_ptr(), m1.data_ptr(), m2.data_ptr(), n_par, work.data_ptr(), cap, u.data_ptr() + q,
                                   i.data_ptr() + q, j.data_ptr() + q, meta.data_ptr() + 20 * s * b, b, 0, nU, nU + nI, 1e-3, 0.9, 0.999,
                                   1e-8, t, k, s, 1e-3, loss.data_ptr(), st)
    e[2].record()
    rc |= L.skr_bpr_fused_end(flat.data_ptr(), m1.data_ptr(), m2.data_ptr(), n_par, work.data_ptr(), cap, sb.data_ptr(), sf.data_ptr(),
                              ns.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, t, k, None, 0, 0, st)
    e[3].record()
    _hip.check(rc)
    t += k
    torch.cuda.synchronize()
    t_plan.append(e[0].elapsed_time(e[1])); t_steps.append(e[1].elapsed_time(e[2])); t_end.append(e[2].elapsed_time(e[3]))
h = n_blocks // 2
print(f"k={k} dbg={os.environ.get('SKR_FUSED_DBG', '0')}: "
      f"plan {np.mean(t_plan[h:]) * 1e3:.1f} us/block, steps {np.mean(t_steps[h:]) * 1e3 / k:.2f} us/step, end {np.mean(t_end[h:]) * 1e3:.1f} us/block"
      f"  (slots {int(ns)})")
