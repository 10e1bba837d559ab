"""item-side product of the BASELINE graph through the plan, timed alone (HIP events, 30 repetitions): the densest rows through
LDS (SKR_SPMM_HOT=1, default) or on the task path (=0); SKR_SPMM_HOT_WGS / SKR_SPMM_HOT_DENSITY as set by the caller."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from skrec.parallel import _csr_from_device_coo  # noqa: E402

dev = torch.device("cuda", 0)
U, I, E = 1_000_000, 100_000, 50_000_000
ds = bench.synth_dataset(U, I, E, 20260101, dev)
u, it = ds["users"].long(), ds["items"].long()
du, di = torch.bincount(u, minlength=U).float(), torch.bincount(it, minlength=I).float()
vals = torch.where(du > 0, du.pow(-0.5), du)[u] * torch.where(di > 0, di.pow(-0.5), di)[it]
a_iu = _csr_from_device_coo(it, u, vals, I, U)
X = torch.randn((U, 64), device=dev)
Y = torch.empty((I, 64), device=dev)
for _ in range(3):
    a_iu.spmm(X, Y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    a_iu.spmm(X, Y)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SKR_SPMM")}, "item_side_ms": e0.elapsed_time(e1) / 30,
                  "plan": a_iu.plan_info(), "checksum": float(Y.double().sum())}), flush=True)
