"""Phase stamps of exact_assign_kernel (development aid).  Build the instrumented library first:
  cd scikit-recommender_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSKR_SAMPLER_STAMPS -c sampler.hip \
      -o build/sampler_stamps.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libskrec_hip_stamps.so \
      build/api.o build/sampler_stamps.o build/eval_select.o build/eval_fused.o build/train.o
Round-1 result (48.4 M slots, 3320 chunks): fixed-point rounds 472 M cycles (2.6 rounds/chunk), staging of
positives + draws 75 M, output 55 M, row window 25 M, anchor 5 M -- one CU, instruction-issue bound."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")): sys.path.insert(0, p)
from skrec import _hip
_hip.LIB_PATH = _hip.LIB_PATH.replace("libskrec_hip.so", "libskrec_hip_stamps.so")
import torch, bench
from skrec.utils.py.random import DeviceSampler
dev = torch.device("cuda", 0)
ds = bench.synth_dataset(1_000_000, 100_000, 50_000_000, 20260101, dev)
nnz = int(ds["rowptr"][-1]); neg = torch.empty(nnz, dtype=torch.int32, device=dev)
s = DeviceSampler(2020)
for _ in range(2):
    s.sample_epoch_exact(100_000, 1_000_000, ds["rowptr"], ds["items"], nnz, 1, neg)
torch.cuda.synchronize()
