"""Hole detector for the fused evaluator's candidate lists: the scratch is pre-filled with a key that outranks
every real candidate (id -1), so a list slot that is counted but was never written shows up as id -1 at rank 0."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from skrec import _hip  # noqa: E402

U, I, nu = 1_000_000, 100_000, 262_144
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
big = bench.synth_dataset(U, I, 50_000_000, 20260101, dev)
L = _hip.lib()
g = torch.Generator(device=dev).manual_seed(1)
Ut = torch.randn(U, 64, generator=g, device=dev) * 0.1
Vt = torch.randn(I, 64, generator=g, device=dev) * 0.1
bias = torch.randn(I, generator=g, device=dev) * 0.05
users = torch.arange(nu, dtype=torch.int32, device=dev)
ws = int(L.skr_eval_fused_workspace(nu, K))
work = torch.empty(ws // 8, dtype=torch.int64, device=dev)
tot = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    work.fill_(-4294967296)
    ids = torch.full((nu, K), -7, dtype=torch.int32, device=dev)
    _hip.check(L.skr_eval_fused_topk(_hip.ptr(Ut), _hip.ptr(users), nu, _hip.ptr(Vt), _hip.ptr(bias), I, 64,
                                     _hip.ptr(big["rowptr"]), _hip.ptr(big["items"]), K, _hip.ptr(ids), None, _hip.ptr(work),
                                     ws, _hip.stream()))
    torch.cuda.synchronize()
    holes = torch.nonzero((ids == -1).any(1)).reshape(-1)
    unwritten = torch.nonzero((ids == -7).any(1)).reshape(-1)
    tot += holes.numel() + unwritten.numel()
    if holes.numel() or unwritten.numel():
        print(f"iteration {it}: {holes.numel()} users read a never-written slot, {unwritten.numel()} users have unwritten outputs;",
              "first:", holes[:8].tolist(), [(h // 64, h % 64) for h in holes[:8].tolist()])
print("mode", os.environ.get("SKR_FUSED_MODE", "bf16x3"), "top_k", K, "total affected:", tot, "dataset checksum", int(big["items"].long().sum()), int(big["rowptr"][-1]), "max row", int((big["rowptr"][1:nu+1]-big["rowptr"][:nu]).max()))
