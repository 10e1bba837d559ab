"""two gloo ranks on one GPU: ShardedLightGCN with the exchange pipelined / layer by layer, and a repeat of the first run
(what differs run to run comes from float atomics, what differs between the orders would be a bug)"""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402  (puts the package on the path)
import test_gpu_dist as T  # noqa: E402


def run(flag):
    os.environ["SKR_DIST_PIPELINE"] = flag
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(T._worker, args=(2, T._free_port(), ret), nprocs=2, join=True)
        return {k: ret[k] for k in range(2)}


if __name__ == "__main__":
    os.environ["SKR_SPMM_PLAN"] = sys.argv[1] if len(sys.argv) > 1 else "1"
    os.environ["SKR_FIRST_HOP_SCATTER"] = "0"
    a, b, c = run("1"), run("1"), run("0")
    for key in ("losses", "U1", "V1", "Uf", "Vf"):
        print(key, "run-to-run", float(np.abs(a[0][key] - b[0][key]).max()), "pipelined vs not", float(np.abs(a[0][key] - c[0][key]).max()))
