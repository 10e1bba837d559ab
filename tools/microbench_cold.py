"""Cold pass of the temporally blocked Adam (skr_adam_block_cold) at the bench's full size, by state of the moments.

    python tools/microbench_cold.py            # rows at rest take the cheap exact path
    SKR_COLD_REST=0 python tools/microbench_cold.py   # every cold row takes the full update (A/B)

states: fresh   m = v = 0 (start of training)
        lively  every row touched in the last step
        steady  user rows last touched Exp(mean U/batch) steps ago, item rows Exp(mean I/(2*batch)) steps ago
        old     every row untouched for 2000 steps
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scikit-recommender_amd"))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", default="fresh,lively,steady,old")
    ap.add_argument("--ks", default="8,16")
    ap.add_argument("--t0s", default="0,20000")
    args = ap.parse_args()
    import torch
    from skrec import _hip
    L = _hip.lib()
    dev = _hip.require_gpu()
    U, I, B = 1_000_000, 100_000, 1024
    rows = U + I
    n = rows * 64 + I
    g = torch.Generator(device=dev).manual_seed(0)
    p = torch.randn(n, generator=g, device=dev) * 0.05
    tag = torch.zeros((n + 63) // 64, dtype=torch.int32, device=dev)

    def state(kind):
        if kind == "fresh":
            return torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        if kind == "lively":
            age = torch.zeros(rows, device=dev)
        elif kind == "old":
            age = torch.full((rows,), 2000.0, device=dev)
        else:
            age = torch.empty(rows, device=dev)
            age[:U].exponential_(1.0 / (U / B), generator=g)
            age[U:].exponential_(1.0 / (I / (2 * B)), generator=g)
        age = torch.cat([age.repeat_interleave(64), torch.zeros(I, device=dev)])
        m = torch.randn(n, generator=g, device=dev) * 1e-3 * torch.exp(age * float(np.log(0.9)))
        v = torch.rand(n, generator=g, device=dev) * 1e-6 * torch.exp(age * float(np.log(0.999)))
        return m, v

    out = {"rest_path": os.environ.get("SKR_COLD_REST", "1") != "0", "n_params": n}
    for kind in args.states.split(","):
        m0, v0 = state(kind)
        for k in [int(x) for x in args.ks.split(",")]:
            for t0 in [int(x) for x in args.t0s.split(",")]:
                ms = []
                for rep in range(4):
                    pp, mm, vv = p.clone(), m0.clone(), v0.clone()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _hip.check(L.skr_adam_block_cold(_hip.ptr(pp), _hip.ptr(mm), _hip.ptr(vv), n, 1e-3, 0.9, 0.999, 1e-8, t0, k,
                                                     _hip.ptr(tag), 1, _hip.stream()))
                    e1.record()
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                out[f"{kind}_k{k}_t{t0}_ms"] = round(min(ms[1:]), 4)
                print(kind, k, t0, out[f"{kind}_k{k}_t{t0}_ms"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
