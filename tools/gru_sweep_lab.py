"""one GRU cell forward for many sessions (the inference sweep's step): ms per call, matrix-core kernel vs SKR_GRU_MFMA=0
usage: python tools/gru_sweep_lab.py [sessions]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scikit-recommender_amd"))
from skrec import _hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
i_d = h = 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
table = torch.randn((100_000, i_d), device=dev, generator=g) * 0.1
idx = torch.randint(0, 100_000, (B,), device=dev, generator=g, dtype=torch.int32)
hp = torch.randn((B, h), device=dev, generator=g) * 0.1
Wg, Wc = torch.randn((i_d + h, 2 * h), device=dev, generator=g) * 0.05, torch.randn((i_d + h, h), device=dev, generator=g) * 0.05
bg, bc = torch.ones(2 * h, device=dev), torch.zeros(h, device=dev)
out = torch.empty((B, h), device=dev)
L, st = _hip.lib(), _hip.stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    e0.record()
    _hip.check(L.skr_gru_cell_fwd(_hip.ptr(table), _hip.ptr(idx), _hip.ptr(hp), None, B, i_d, h, _hip.ptr(Wg), _hip.ptr(bg), _hip.ptr(Wc),
                                  _hip.ptr(bc), 0, None, None, None, _hip.ptr(out), st))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.mean(ts[1:]))
flop = 2.0 * B * (i_d + h) * 3 * h
print(f"SKR_GRU_MFMA={os.environ.get('SKR_GRU_MFMA', '1')}: {B} sessions, d = {h}: {ms:.3f} ms per step, {flop / ms / 1e9:.1f} TFLOP/s, "
      f"checksum {float(out.double().sum()):.6f}")
