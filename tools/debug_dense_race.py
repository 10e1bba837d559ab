import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import torch
import bench
from skrec import _hip
U, I, K, nu = 1_000_000, 100_000, 100, 262_144
dev = torch.device("cuda", 0)
big = bench.synth_dataset(U, I, 50_000_000, 20260101, dev)
L = _hip.lib()
g = torch.Generator(device=dev).manual_seed(1)
Ut = torch.randn(U, 64, generator=g, device=dev) * 0.1
Vt = torch.randn(I, 64, generator=g, device=dev) * 0.1
bias = torch.randn(I, generator=g, device=dev) * 0.05
users = torch.arange(nu, dtype=torch.int32, device=dev)
CH, REPS = 4096, 6
tot = 0
for s0 in range(0, nu, CH):
    us = users[s0:s0 + CH].contiguous()
    dense = _hip.score_matrix(Ut, us.cpu().numpy(), Vt, bias)
    _hip.check(L.skr_mask_train(_hip.ptr(dense), CH, I, I, _hip.ptr(us), _hip.ptr(big["rowptr"]), _hip.ptr(big["items"]), _hip.stream()))
    outs = []
    for r in range(REPS):
        ids2 = torch.full((CH, K), -7, dtype=torch.int32, device=dev)
        _hip.check(L.skr_eval_scores(_hip.ptr(dense), CH, I, I, None, None, None, 0, K, None, _hip.ptr(ids2), None, _hip.stream()))
        torch.cuda.synchronize()
        outs.append(ids2)
    for r in range(1, REPS):
        bad = torch.nonzero((outs[r] != outs[0]).any(1)).reshape(-1)
        oob = torch.nonzero(((outs[r] < 0) | (outs[r] >= I)).any(1)).reshape(-1)
        if bad.numel() or oob.numel():
            tot += 1
            row = int((bad if bad.numel() else oob)[0])
            top = torch.topk(dense[row], K + 1).values
            ties = int((top[1:] == top[:-1]).sum())
            print(f"chunk {s0} rep {r}: {bad.numel()} rows differ from rep 0, {oob.numel()} rows out of range; row {row}: ties among top-{K+1}: {ties};",
                  "rep0", outs[0][row][:6].tolist(), "this", outs[r][row][:6].tolist(),
                  "diff ranks", torch.nonzero(outs[r][row] != outs[0][row]).reshape(-1)[:6].tolist())
print("inconsistent (chunk, rep) pairs:", tot)
