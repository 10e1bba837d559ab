import ctypes, os, sys
REPO="/root/repo"
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd"), os.path.join(REPO, "tests")): sys.path.insert(0, p)
import numpy as np
from gpu_utils import fused_topk
from skrec import _hip
rng=np.random.default_rng(5)
B,I,K=2048,50000,10
for spread in (0.0,0.25,0.5,1.0,1.5,2.0):
    U=(rng.standard_normal((B,64))*0.1*np.exp(rng.standard_normal((B,1))*spread)).astype(np.float32)
    V=(rng.standard_normal((I,64))*0.1*np.exp(rng.standard_normal((I,1))*spread)).astype(np.float32)
    ids,sc=fused_topk(U,np.arange(B,dtype=np.int32),V,None,None,np.zeros(0,np.int32),K)
    n=ctypes.c_int32(-1); _hip.check(_hip.lib().skr_eval_fused_rejected(ctypes.byref(n), _hip.stream()))
    print(f"row-magnitude spread sigma={spread}: rejected {n.value}/{B} ({100*n.value/B:.1f} %), row norm max/median = {np.linalg.norm(U,axis=1).max()/np.median(np.linalg.norm(U,axis=1)):.1f}")
