"""Experiment (round 3): the item side of the propagation for rows below a long-row threshold as one sweep over the column blocks
with LDS accumulators (tools/lab/sweep_lab.hip), against the plan's row / task kernels.  Prints JSON lines."""
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from skrec import _hip  # noqa: E402
from skrec.recommender.LightGCN import DeviceCSR  # noqa: E402

dev = torch.device("cuda", 0)
LAB = C.CDLL(os.path.join(REPO, "tools", "lab", "libsweep_lab.so"))
vp = C.c_void_p
LAB.lab_sweep.restype = C.c_int
LAB.lab_sweep.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
U, I, E = int(os.environ.get("LAB_USERS", 1_000_000)), int(os.environ.get("LAB_ITEMS", 100_000)), int(os.environ.get("LAB_INTER", 50_000_000))
R = 128


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def make_csr(rows, cols, vals, n_rows, n_cols):
    order = torch.argsort(rows * n_cols + cols)
    c = DeviceCSR.__new__(DeviceCSR)
    c.shape, c.nnz = (n_rows, n_cols), int(rows.numel())
    c.rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    c.rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_rows), 0)
    c.col, c.val = cols[order].int().contiguous(), vals[order].float().contiguous()
    return c


ds = bench.synth_dataset(U, I, E, 20260101, dev)
u, it = ds["users"].long(), ds["items"].long()
nnz = u.numel()
du, di = torch.bincount(u, minlength=U).float(), torch.bincount(it, minlength=I).float()
vals = torch.where(du > 0, du.pow(-0.5), du)[u] * torch.where(di > 0, di.pow(-0.5), di)[it]
a_iu = make_csr(it, u, vals, I, U)
X = torch.randn((U, 64), device=dev)
Yref = torch.empty((I, 64), device=dev)
t_full = timeit(lambda: a_iu.spmm(X, Yref))
emit(kind="plan_full", ms=t_full, info=a_iu.plan_info())

for T, CB, budget in [(int(x) for x in cfg.split(":")) for cfg in (os.environ.get("LAB_CFGS") or "8192:8192:40000,8192:16384:40000,8192:2048:40000,8192:8192:20000").split(",")]:
    deg = (a_iu.rowptr[1:] - a_iu.rowptr[:-1]).cpu().numpy()
    t0 = time.perf_counter()
    grp_of, lrow_of = np.full(I, -1, np.int64), np.zeros(I, np.int64)
    group_rows = []
    cur, cur_nnz = [], 0

    def close_group():
        g = len(group_rows)
        rows = sorted(cur, key=lambda r: -deg[r])
        load, cnt = [0, 0, 0, 0], [0, 0, 0, 0]
        table = np.full(R, -1, np.int32)
        for r in rows:
            w = min((w_ for w_ in range(4) if cnt[w_] < R // 4), key=lambda w_: load[w_])
            lr = cnt[w] * 4 + w
            cnt[w] += 1
            load[w] += deg[r]
            grp_of[r], lrow_of[r] = g, lr
            table[lr] = r
        group_rows.append(table)
    for r in range(I):
        if deg[r] == 0 or deg[r] >= T:
            continue
        if cur and (cur_nnz + deg[r] > budget or len(cur) == R):
            close_group()
            cur, cur_nnz = [], 0
        cur.append(r)
        cur_nnz += deg[r]
    if cur:
        close_group()
    n_groups = len(group_rows)
    t_host = time.perf_counter() - t0
    d_grp, d_lrow = torch.from_numpy(grp_of).to(dev), torch.from_numpy(lrow_of).to(dev)
    rows_of_entry = torch.repeat_interleave(torch.arange(I, device=dev), a_iu.rowptr[1:] - a_iu.rowptr[:-1])
    sel = d_grp[rows_of_entry] >= 0
    er, ec, ev = rows_of_entry[sel], a_iu.col[sel].long(), a_iu.val[sel]
    n_cb = (U + CB - 1) // CB
    seg = d_grp[er] * 4 + (d_lrow[er] & 3)
    key = ((seg * n_cb + ec // CB) * R + d_lrow[er]) * CB + (ec % CB)
    order = torch.argsort(key)
    scol, sval, slrow = ec[order].int().contiguous(), ev[order].contiguous(), d_lrow[er][order].to(torch.uint8).contiguous()
    seg_sorted = seg[order]
    seg_ptr = torch.searchsorted(seg_sorted, torch.arange(n_groups * 4 + 1, device=dev)).contiguous()
    d_group_rows = torch.from_numpy(np.concatenate(group_rows)).to(dev)
    Y = torch.zeros((I, 64), device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(mode=0):
        assert LAB.lab_sweep(mode, n_groups, seg_ptr.data_ptr(), scol.data_ptr(), sval.data_ptr(), slrow.data_ptr(), d_group_rows.data_ptr(),
                             X.data_ptr(), Y.data_ptr(), st) == 0
    run()
    torch.cuda.synchronize()
    swept = torch.from_numpy((grp_of >= 0)).to(dev)
    err = float((Y[swept] - Yref[swept]).abs().max() / Yref[swept].abs().max())
    t_sweep = timeit(run)
    t_gather_only = timeit(lambda: run(1))
    # the long rows alone through the plan (tasks + reduce)
    keep_long = (~sel)
    lr_, lc_, lv_ = rows_of_entry[keep_long], a_iu.col[keep_long].long(), a_iu.val[keep_long]
    a_long = make_csr(lr_, lc_, lv_, I, U)
    Y2 = torch.empty((I, 64), device=dev)
    t_long = timeit(lambda: a_long.spmm(X, Y2))
    seg_len = (seg_ptr[1:] - seg_ptr[:-1]).float()
    emit(kind="sweep", T=T, CB=CB, budget=budget, n_groups=n_groups, swept_nnz=int(sel.sum()), long_nnz=int(keep_long.sum()), rel_err=err,
         sweep_ms=t_sweep, gather_only_ms=t_gather_only, gather_only_TBps=int(sel.sum()) * 256 / (t_gather_only * 1e-3) / 1e12, long_rows_plan_ms=t_long, total_ms=t_sweep + t_long, plan_full_ms=t_full, host_build_s=t_host,
         wave_entries_max=float(seg_len.max()), wave_entries_mean=float(seg_len.mean()),
         sweep_gather_TBps=int(sel.sum()) * 256 / (t_sweep * 1e-3) / 1e12)
    del a_long, scol, sval, slrow, key, order, seg, er, ec, ev
    torch.cuda.empty_cache()
