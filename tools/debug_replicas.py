"""2 gloo ranks on one GPU: ShardedBPRMF.train_block with the default Adam block; after every block compare the replicated
item table bit for bit across the ranks and describe the first rows that differ."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from skrec.parallel import DistContext, ShardedBPRMF  # noqa: E402
dev = torch.device("cuda", 0)
nU, nI, b, K = int(os.environ.get("DBG_U", 20000)), int(os.environ.get("DBG_I", 3000)), int(os.environ.get("DBG_B", 1024)), 96
g = torch.Generator().manual_seed(5)
user0 = torch.randn(nU, 64, generator=g) * 0.01
item0 = torch.randn(nI, 64, generator=g) * 0.01
eng = ShardedBPRMF(DistContext(rank, world), user0, item0, torch.zeros(nI), 1e-3, 1e-3, dev)
gd = torch.Generator(device=dev).manual_seed(9)
uu = torch.randint(0, nU, (K * b,), generator=gd, device=dev, dtype=torch.int32)
ii = torch.randint(0, nI, (K * b,), generator=gd, device=dev, dtype=torch.int32)
jj = torch.randint(0, nI, (K * b,), generator=gd, device=dev, dtype=torch.int32)
bounds = [(s * b, (s + 1) * b) for s in range(K)]
losses = torch.zeros((K, 2), device=dev)
kblk = eng.adam_block
for s0 in range(0, K, kblk):
    eng.train_block(uu, ii, jj, bounds[s0:s0 + kblk], losses[s0:s0 + kblk])
    torch.cuda.synchronize()
    for name, t in (("V", eng.item_rows), ("bias", eng.item_bias.view(-1, 1)), ("mV", eng.optimizer.m[eng.n_local * 64:(eng.n_local + nI) * 64].view(nI, 64)),
                    ("vV", eng.optimizer.v[eng.n_local * 64:(eng.n_local + nI) * 64].view(nI, 64))):
        mine = t.clone()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        diff = (other[0].view(torch.int32) != other[1].view(torch.int32)).any(1)
        if rank == 0:
            nd = int(diff.sum())
            print(f"block {s0 // kblk}: {name} rows differing {nd} / {t.shape[0]}", flush=True)
            if nd and name == "V":
                rows = diff.nonzero().flatten()[:5].tolist()
                blk_items = torch.cat([ii[s0 * b:(s0 + kblk) * b], jj[s0 * b:(s0 + kblk) * b]]).long()
                for r in rows:
                    occ = ((ii.view(K, b)[s0:s0 + kblk] == r) | (jj.view(K, b)[s0:s0 + kblk] == r)).any(1).nonzero().flatten().tolist()
                    print("   row", r, "in block's batches:", bool((blk_items == r).any()), "steps", occ, "max abs diff",
                          float((other[0][r] - other[1][r]).abs().max()), flush=True)
dist.barrier()
dist.destroy_process_group()
