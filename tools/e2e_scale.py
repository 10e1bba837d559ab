"""End-to-end check of the drop-in API at scale: writes a synthetic dataset in the reference's TSV
format, then constructs and fits BPRMF through RunConfig exactly like run_skrec.py does, timing each
phase (dataset load, model construction, epochs = sampling + training + evaluation)."""
import argparse
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=200_000)
ap.add_argument("--items", type=int, default=20_000)
ap.add_argument("--interactions", type=int, default=10_000_000)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--model", default="BPRMF")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--sampler", default="exact")
args = ap.parse_args()

dev = torch.device("cuda", 0)
t0 = time.time()
ds = bench.synth_dataset(args.users, args.items, args.interactions, 20260101, dev)
root = tempfile.mkdtemp(prefix="skrec_e2e_")
d = os.path.join(root, "syn")
os.makedirs(d)
import pandas as pd  # noqa: E402
u, i = ds["users"].cpu().numpy(), ds["items"].cpu().numpy()
perm = np.random.default_rng(0).permutation(len(u))          # file order is not grouped by user
pd.DataFrame({"u": u[perm], "i": i[perm], "r": 1.0, "t": np.arange(len(u))}).to_csv(os.path.join(d, "syn.train"), sep="\t",
                                                                                 header=False, index=False)
tu = np.arange(args.users)
pd.DataFrame({"u": tu, "i": ds["test_item"].cpu().numpy(), "r": 1.0, "t": len(u) + tu}).to_csv(
    os.path.join(d, "syn.test"), sep="\t", header=False, index=False)
del ds
torch.cuda.empty_cache()
print(f"[e2e] dataset written: {len(u)} train rows, {time.time() - t0:.1f}s", flush=True)

os.chdir(root)
from skrec import RunConfig, ModelRegistry  # noqa: E402
np.random.seed(2021)
torch.manual_seed(2021)
rc = RunConfig(recommender=args.model, data_dir=d, file_column="UIRT", sep="\t", metric=("Recall", "NDCG"), top_k=(10, 20),
               sampler_mode=args.sampler)
reg = ModelRegistry()
reg.load_skrec_model(args.model)
cls, _ = reg.get_model(args.model)
t0 = time.time()
if os.environ.get("E2E_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    model = cls(rc, {"epochs": args.epochs, "batch_size": args.batch})
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
else:
    model = cls(rc, {"epochs": args.epochs, "batch_size": args.batch})
print(f"[e2e] model constructed in {time.time() - t0:.1f}s", flush=True)
from skrec.io import PairwiseIterator  # noqa: E402
t0 = time.time()
it = PairwiseIterator(model.dataset.train_data, batch_size=args.batch, shuffle=True, sampler_mode=args.sampler)
print(f"[e2e] iterator constructed in {time.time() - t0:.1f}s ({len(it)} steps/epoch)", flush=True)
ap_steps = int(os.environ.get("E2E_MAX_STEPS", "0"))
if ap_steps:   # graph models re-propagate the whole graph every step: time a bounded number of steps
    import itertools
    orig_iter = it.iter_device
    it.iter_device = lambda: itertools.islice(orig_iter(), ap_steps)
    it.__class__.__len__ = lambda self: ap_steps
if os.environ.get("E2E_EPOCH_AHEAD", "1") != "0":   # what BPRMF.fit() does: next epoch's negatives and permutation in the background
    it.epoch_ahead(True)
for ep in range(args.epochs):
    torch.cuda.synchronize()
    t0 = time.time()
    if hasattr(model, "pre_epoch_processing"):
        model.pre_epoch_processing()
    model.train_epoch(it)
    torch.cuda.synchronize()
    t1 = time.time()
    rep = model.evaluate()
    torch.cuda.synchronize()
    t2 = time.time()
    n = len(it.all_users) if not ap_steps else ap_steps * args.batch
    if os.environ.get("SKR_COLD_STATS") == "1":
        import ctypes as C
        from skrec import _hip
        cen = (C.c_uint64 * 3)()
        _hip.check(_hip.lib().skr_cold_pass_census(cen, 1))
        tot = max(1, sum(cen))
        print(f"[e2e] epoch {ep}: cold blocks at rest {cen[0] / tot:.3f}, ordinary magnitudes {cen[1] / tot:.3f}, "
              f"general {cen[2] / tot:.3f} (of {tot} block visits)", flush=True)
    print(f"[e2e] epoch {ep}: train {t1 - t0:.2f}s = {n / (t1 - t0) / 1e6:.2f} M interactions/s; "
          f"eval {t2 - t1:.2f}s = {args.users / (t2 - t1) / 1e6:.2f} M users/s; {rep.values_str}", flush=True)
it.epoch_ahead(False)
