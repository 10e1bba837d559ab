"""SpMM experiments on the BASELINE graph (1 M users / 100 k items / ~48 M interactions, 'pre' normalisation):
degree statistics, the product kernel on each side of the bipartite graph, and the lab variants of
tools/lab/spmm_lab.hip.  Prints one JSON object per line; numbers feed DESIGN.md 4.3."""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from skrec import _hip  # noqa: E402

dev = torch.device("cuda", 0)
L = _hip.lib()
LAB = C.CDLL(os.path.join(REPO, "tools", "lab", "libspmm_lab.so"))
vp, i32 = C.c_void_p, C.c_int
LAB.lab_spmm_rows.restype = i32
LAB.lab_spmm_rows.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
WHAT = set((os.environ.get("LAB_WHAT") or "stats,base,user,itemcold").split(","))
U, I = int(os.environ.get("LAB_USERS", 1_000_000)), int(os.environ.get("LAB_ITEMS", 100_000))
E = int(os.environ.get("LAB_INTER", 50_000_000))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def csr(rows, cols, vals, n_rows, n_cols):
    order = torch.argsort(rows * n_cols + cols)
    rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(rows, minlength=n_rows), 0)
    return rp, cols[order].int().contiguous(), vals[order].float().contiguous()


def hot_split(rp, col, val, n_rows, rank, H):
    """entries of every row reordered: hot columns (rank < H) first, renamed to their rank; -> col2, val2, hotcnt"""
    rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), rp[1:] - rp[:-1])
    rk = rank[col.long()]
    cold = (rk >= H)
    key = (rows * 2 + cold.long()) * (1 << 21) + torch.where(cold, col.long(), rk)   # cols < 2^21
    order = torch.argsort(key)
    col2 = torch.where(cold, col.long(), rk)[order].int().contiguous()
    hotcnt = torch.zeros(n_rows, dtype=torch.int64, device=dev).index_add_(0, rows, (~cold).long()).int().contiguous()
    return col2, val[order].contiguous(), hotcnt


ds = bench.synth_dataset(U, I, E, 20260101, dev)
u, it = ds["users"].long(), ds["items"].long()
nnz = u.numel()
du = torch.bincount(u, minlength=U).float()
di = torch.bincount(it, minlength=I).float()
vals = torch.where(du > 0, du.pow(-0.5), du)[u] * torch.where(di > 0, di.pow(-0.5), di)[it]
if "stats" in WHAT:
    srt = torch.sort(di, descending=True).values
    cum = torch.cumsum(srt, 0) / nnz
    emit(kind="stats", nnz=nnz, item_deg_max=float(srt[0]), item_deg_at={str(h): float(srt[h - 1]) for h in (16, 64, 128, 256, 512, 1024, 4096, 16384)},
         top_share={str(h): round(float(cum[h - 1]), 4) for h in (16, 64, 128, 256, 384, 512, 640, 1024, 2048, 4096, 8192, 16384, 32768)},
         user_deg_max=float(du.max()), user_deg_mean=float(du.mean()))
    us = torch.sort(du, descending=True).values
    ucum = torch.cumsum(us, 0) / nnz
    emit(kind="stats_users", top_share={str(h): round(float(ucum[h - 1]), 4) for h in (1024, 16384, 65536, 262144)})

ui_rp, ui_col, ui_val = csr(u, it, vals, U, I)
iu_rp, iu_col, iu_val = csr(it, u, vals, I, U)
rank_i = torch.empty(I, dtype=torch.int64, device=dev)
rank_i[torch.argsort(di, descending=True)] = torch.arange(I, device=dev)
g = torch.Generator(device=dev).manual_seed(5)
X_i = torch.randn(I, 64, generator=g, device=dev) * 0.1
X_u = torch.randn(U, 64, generator=g, device=dev) * 0.1


def product(n_rows, rp, col, val, X, Y, n):
    _hip.check(L.skr_csr_spmm(n_rows, _hip.ptr(rp), _hip.ptr(col), _hip.ptr(val), _hip.ptr(X), 64, n, None, _hip.ptr(Y), None, 1.0,
                              _hip.stream()))


def lab(variant, nf, waves, grid, n_rows, rp, hotcnt, col, val, X, hot_cols, n_hot, Y):
    rc = LAB.lab_spmm_rows(variant, nf, waves, grid, n_rows, _hip.ptr(rp), _hip.ptr(hotcnt), _hip.ptr(col), _hip.ptr(val), _hip.ptr(X),
                           _hip.ptr(hot_cols), n_hot, _hip.ptr(Y), _hip.stream())
    assert rc == 0, (rc, variant, nf, waves)


Yu_ref = torch.empty(U, 64, device=dev)
Yi_ref = torch.empty(I, 64, device=dev)
if "base" in WHAT:
    t = timeit(lambda: product(U, ui_rp, ui_col, ui_val, X_i, Yu_ref, nnz))
    emit(kind="base", side="user", ms=t, gather_TBps=nnz * 256 / t / 1e9)
    t = timeit(lambda: product(I, iu_rp, iu_col, iu_val, X_u, Yi_ref, nnz))
    emit(kind="base", side="item", ms=t, gather_TBps=nnz * 256 / t / 1e9)
else:
    product(U, ui_rp, ui_col, ui_val, X_i, Yu_ref, nnz)
    product(I, iu_rp, iu_col, iu_val, X_u, Yi_ref, nnz)

if "user" in WHAT:
    Y = torch.empty(U, 64, device=dev)
    zero_cnt = torch.zeros(U, dtype=torch.int32, device=dev)
    for H in (0, 256, 512):
        if H:
            col2, val2, hotcnt = hot_split(ui_rp, ui_col, ui_val, U, rank_i, H)
            hot_cols = torch.argsort(di, descending=True)[:H].int().contiguous()
            hot_share = float(hotcnt.sum()) / nnz
        else:
            col2, val2, hotcnt, hot_cols, hot_share = ui_col, ui_val, zero_cnt, None, 0.0
        for variant, nf, waves in ((0, 4, 4), (0, 8, 4), (0, 16, 4), (0, 8, 16), (0, 16, 16), (1, 1, 4), (1, 2, 4), (1, 4, 4), (1, 8, 4),
                                   (1, 2, 16), (1, 4, 16), (1, 8, 16), (0, 8, 8), (0, 16, 8), (1, 2, 8), (1, 4, 8)):
            per_cu = 1 if (H * 256 > 80 * 1024) else (2 if H else 8)
            if H and waves * per_cu < 8:
                continue          # a persistent LDS-cached workgroup needs enough waves per CU
            grids = (256 * per_cu,) if H else (256 * per_cu * (16 // waves) // 4 * 4, )
            for grid in grids:
                Y.zero_()
                try:
                    t = timeit(lambda: lab(variant, nf, waves, grid, U, ui_rp, hotcnt, col2, val2, X_i, hot_cols, H, Y))
                except AssertionError as e:
                    emit(kind="user", H=H, variant=variant, nf=nf, waves=waves, grid=grid, error=str(e))
                    continue
                err = float((Y - Yu_ref).abs().max())
                emit(kind="user", H=H, hot_share=round(hot_share, 4), variant=variant, nf=nf, waves=waves, grid=grid, ms=round(t, 4),
                     gather_TBps=round(nnz * 256 / t / 1e9, 2), cold_TBps=round(nnz * (1 - hot_share) * 256 / t / 1e9, 2), max_err=err)

if "itemcold" in WHAT:
    # the item side without its long rows: how fast do random rows of the 256 MB user table arrive?
    lens = iu_rp[1:] - iu_rp[:-1]
    for thr in (1024, 8192):
        keep_row = lens < thr
        rows = torch.repeat_interleave(torch.arange(I, device=dev), lens)
        sel = keep_row[rows]
        rp2 = torch.zeros(I + 1, dtype=torch.int64, device=dev)
        rp2[1:] = torch.cumsum(torch.where(keep_row, lens, torch.zeros_like(lens)), 0)
        col2, val2 = iu_col[sel].contiguous(), iu_val[sel].contiguous()
        n2 = int(rp2[-1])
        Yb = torch.empty(I, 64, device=dev)
        t = timeit(lambda: product(I, rp2, col2, val2, X_u, Yb, n2))
        emit(kind="itemcold", thr=thr, rows_kept=int(keep_row.sum()), nnz=n2, share=round(n2 / nnz, 4), variant="product", ms=round(t, 4),
             gather_TBps=round(n2 * 256 / t / 1e9, 2))
        Y = torch.empty(I, 64, device=dev)
        zc = torch.zeros(I, dtype=torch.int32, device=dev)
        for variant, nf, waves in ((0, 8, 4), (0, 16, 4), (1, 2, 4), (1, 4, 4), (1, 8, 4), (1, 4, 16), (1, 8, 16)):
            grid = 256 * 8 * 4 // waves
            Y.zero_()
            t = timeit(lambda: lab(variant, nf, waves, grid, I, rp2, zc, col2, val2, X_u, None, 0, Y))
            emit(kind="itemcold", thr=thr, variant=variant, nf=nf, waves=waves, grid=grid, ms=round(t, 4),
                 gather_TBps=round(n2 * 256 / t / 1e9, 2), max_err=float((Y - Yb).abs().max()))

LAB.lab_spmm_rows_pf.restype = i32
LAB.lab_spmm_rows_pf.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
LAB.lab_spmm_blocked.restype = i32
LAB.lab_spmm_blocked.argtypes = [i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp]


def lab_pf(nf, waves, grid, n_rows, rp, col, val, X, Y):
    rc = LAB.lab_spmm_rows_pf(nf, waves, grid, n_rows, _hip.ptr(rp), _hip.ptr(col), _hip.ptr(val), _hip.ptr(X), _hip.ptr(Y), _hip.stream())
    assert rc == 0, (rc, nf, waves)


if "userpf" in WHAT:
    Y = torch.empty(U, 64, device=dev)
    for nf, waves in ((2, 4), (4, 4), (8, 4), (2, 8), (4, 8), (4, 16)):
        for mult in (4, 8):
            grid = 256 * mult * 4 // waves
            Y.zero_()
            t = timeit(lambda: lab_pf(nf, waves, grid, U, ui_rp, ui_col, ui_val, X_i, Y))
            emit(kind="userpf", nf=nf, waves=waves, grid=grid, ms=round(t, 4), gather_TBps=round(nnz * 256 / t / 1e9, 2),
                 max_err=float((Y - Yu_ref).abs().max()))

if "itemblk" in WHAT:
    lens = iu_rp[1:] - iu_rp[:-1]
    rows_all = torch.repeat_interleave(torch.arange(I, device=dev), lens)
    for T in (512, 2048):
        isA = lens >= T
        selA = isA[rows_all]
        # tier B: the short rows, row by row (long rows emptied)
        rpB = torch.zeros(I + 1, dtype=torch.int64, device=dev)
        rpB[1:] = torch.cumsum(torch.where(isA, torch.zeros_like(lens), lens), 0)
        colB, valB = iu_col[~selA].contiguous(), iu_val[~selA].contiguous()
        nB = int(rpB[-1])
        YB = torch.empty(I, 64, device=dev)
        YB_ref = torch.empty(I, 64, device=dev)
        tB0 = timeit(lambda: product(I, rpB, colB, valB, X_u, YB_ref, nB))
        best = None
        for nf, waves, mult in ((4, 4, 8), (4, 8, 8), (2, 4, 8), (8, 4, 8), (4, 4, 4)):
            grid = 256 * mult * 4 // waves
            tB = timeit(lambda: lab_pf(nf, waves, grid, I, rpB, colB, valB, X_u, YB))
            emit(kind="itemB", T=T, nnz=nB, share=round(nB / nnz, 4), nf=nf, waves=waves, grid=grid, ms=round(tB, 4), product_ms=round(tB0, 4),
                 gather_TBps=round(nB * 256 / tB / 1e9, 2), max_err=float((YB - YB_ref).abs().max()))
            best = tB if best is None else min(best, tB)
        # tier A: long rows, column-blocked tasks
        rA, cA, vA = rows_all[selA], iu_col[selA].long(), iu_val[selA]
        nA = rA.numel()
        YA_ref = torch.zeros(I, 64, device=dev)
        YA_ref[isA] = Yi_ref[isA]
        for CBLK in (8192, 16384, 32768):
            n_blocks = (U + CBLK - 1) // CBLK
            blk = cA // CBLK
            order = torch.argsort((blk * I + rA) * U + cA)
            col_s, val_s = cA[order].int().contiguous(), vA[order].contiguous()
            seg_key = (blk * I + rA)[order]
            uniq, counts = torch.unique_consecutive(seg_key, return_counts=True)
            n_t = (counts + 255) // 256
            seg_start = torch.cumsum(counts, 0) - counts
            task_seg = torch.repeat_interleave(torch.arange(uniq.numel(), device=dev), n_t)
            n_tasks = task_seg.numel()
            first_task = torch.cumsum(n_t, 0) - n_t
            idx_in_seg = torch.arange(n_tasks, device=dev) - first_task[task_seg]
            task_beg = (seg_start[task_seg] + idx_in_seg * 256).contiguous()
            task_len = torch.minimum(torch.full_like(idx_in_seg, 256), counts[task_seg] - idx_in_seg * 256).int().contiguous()
            task_row = (uniq % I)[task_seg]
            task_blk = torch.div(uniq, I, rounding_mode="floor")[task_seg]
            tptr = torch.zeros(n_blocks + 1, dtype=torch.int64, device=dev)
            tptr[1:] = torch.cumsum(torch.bincount(task_blk, minlength=n_blocks), 0)
            part = torch.zeros(n_tasks, 64, device=dev)
            for nf, waves, wgs in ((4, 4, 256), (4, 4, 128), (4, 8, 128), (2, 4, 256), (8, 4, 256), (4, 16, 64)):
                def run():
                    rc = LAB.lab_spmm_blocked(nf, waves, wgs, _hip.ptr(tptr), n_blocks, _hip.ptr(task_beg), _hip.ptr(task_len), _hip.ptr(col_s),
                                              _hip.ptr(val_s), _hip.ptr(X_u), _hip.ptr(part), _hip.stream())
                    assert rc == 0
                tA = timeit(run)
                YA = torch.zeros(I, 64, device=dev).index_add_(0, task_row, part)
                emit(kind="itemA", T=T, CBLK=CBLK, n_blocks=n_blocks, rowsA=int(isA.sum()), nnz=nA, share=round(nA / nnz, 4), segs=int(uniq.numel()),
                     tasks=n_tasks, part_MB=round(n_tasks * 256 / 1e6, 1), nf=nf, waves=waves, wgs_per_xcd=wgs, ms=round(tA, 4),
                     gather_TBps=round(nA * 256 / tA / 1e9, 2), max_err=float((YA - YA_ref).abs().max()), total_with_bestB_ms=round(tA + best, 4))

LAB.lab_spmm_rows_v4.restype = i32
LAB.lab_spmm_rows_v4.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
LAB.lab_reduce_parts.restype = i32
LAB.lab_reduce_parts.argtypes = [i32, vp, vp, vp, vp, vp, vp]

if "v4" in WHAT:
    Y = torch.empty(U, 64, device=dev)
    zero_cnt = torch.zeros(U, dtype=torch.int32, device=dev)

    def v4(nf, waves, grid, ls, nt, hotcnt, col2, val2, hot_cols, H):
        rc = LAB.lab_spmm_rows_v4(nf, waves, grid, ls, nt, U, _hip.ptr(ui_rp), _hip.ptr(hotcnt), _hip.ptr(col2), _hip.ptr(val2), _hip.ptr(X_i),
                                  _hip.ptr(hot_cols), H, _hip.ptr(Y), _hip.stream())
        assert rc == 0, rc
    for nf, waves, grid in ((4, 4, 8192), (4, 8, 4096), (2, 4, 8192)):
        Y.zero_()
        t = timeit(lambda: v4(nf, waves, grid, 64, 1, zero_cnt, ui_col, ui_val, None, 0))
        emit(kind="v4_nt", nf=nf, waves=waves, grid=grid, ms=round(t, 4), gather_TBps=round(nnz * 256 / t / 1e9, 2),
             max_err=float((Y - Yu_ref).abs().max()))
    for H in (256, 512, 576):
        col2, val2, hotcnt = hot_split(ui_rp, ui_col, ui_val, U, rank_i, H)
        hot_cols = torch.argsort(di, descending=True)[:H].int().contiguous()
        hot_share = float(hotcnt.sum()) / nnz
        for nf, waves, ls, nt in ((4, 16, 68, 0), (4, 16, 68, 1), (2, 16, 68, 0), (4, 16, 64, 1), (4, 8, 68, 0), (4, 8, 68, 1)):
            per_cu = max(1, (160 * 1024) // (H * ls * 4))
            if waves * per_cu > 32:
                per_cu = 32 // waves
            grid = 256 * per_cu
            Y.zero_()
            try:
                t = timeit(lambda: v4(nf, waves, grid, ls, nt, hotcnt, col2, val2, hot_cols, H))
            except AssertionError as e:
                emit(kind="v4_hot", H=H, nf=nf, waves=waves, ls=ls, nt=nt, error=str(e))
                continue
            emit(kind="v4_hot", H=H, hot_share=round(hot_share, 4), nf=nf, waves=waves, ls=ls, nt=nt, grid=grid, ms=round(t, 4),
                 gather_TBps=round(nnz * 256 / t / 1e9, 2), cold_TBps=round(nnz * (1 - hot_share) * 256 / t / 1e9, 2),
                 max_err=float((Y - Yu_ref).abs().max()))

if "itemctl" in WHAT:
    lens = iu_rp[1:] - iu_rp[:-1]
    rows_all = torch.repeat_interleave(torch.arange(I, device=dev), lens)
    T = 512
    isA = lens >= T
    selA = isA[rows_all]
    rA, cA, vA = rows_all[selA], iu_col[selA].long(), iu_val[selA]
    nA = rA.numel()
    YA_ref = torch.zeros(I, 64, device=dev)
    YA_ref[isA] = Yi_ref[isA]
    for CBLK in (U, 16384):     # U: no column blocking at all (tasks = consecutive 256-entry pieces of the CSR rows)
        n_blocks = (U + CBLK - 1) // CBLK
        blk = cA // CBLK
        order = torch.argsort((blk * I + rA) * U + cA)
        col_s, val_s = cA[order].int().contiguous(), vA[order].contiguous()
        seg_key = (blk * I + rA)[order]
        uniq, counts = torch.unique_consecutive(seg_key, return_counts=True)
        n_t = (counts + 255) // 256
        seg_start = torch.cumsum(counts, 0) - counts
        task_seg = torch.repeat_interleave(torch.arange(uniq.numel(), device=dev), n_t)
        n_tasks = task_seg.numel()
        first_task = torch.cumsum(n_t, 0) - n_t
        idx_in_seg = torch.arange(n_tasks, device=dev) - first_task[task_seg]
        task_beg = (seg_start[task_seg] + idx_in_seg * 256).contiguous()
        task_len = torch.minimum(torch.full_like(idx_in_seg, 256), counts[task_seg] - idx_in_seg * 256).int().contiguous()
        task_row = (uniq % I)[task_seg]
        task_blk = torch.div(uniq, I, rounding_mode="floor")[task_seg]
        if CBLK == U:       # one launch: the task list cut into 8 equal parts
            n_blocks = 8
            tptr = (torch.arange(9, device=dev) * n_tasks // 8).long().contiguous()
        else:
            tptr = torch.zeros(n_blocks + 1, dtype=torch.int64, device=dev)
            tptr[1:] = torch.cumsum(torch.bincount(task_blk, minlength=n_blocks), 0)
        part = torch.zeros(n_tasks, 64, device=dev)
        # row -> its tasks (for the ordered reduce)
        long_rows = torch.nonzero(isA).flatten().int().contiguous()
        slot = torch.full((I,), -1, dtype=torch.int64, device=dev)
        slot[long_rows.long()] = torch.arange(long_rows.numel(), device=dev)
        tr_slot = slot[task_row]
        rt_order = torch.argsort(tr_slot * (1 << 22) + torch.arange(n_tasks, device=dev), stable=True)
        rt_ids = rt_order.int().contiguous()
        rt_ptr = torch.zeros(long_rows.numel() + 1, dtype=torch.int64, device=dev)
        rt_ptr[1:] = torch.cumsum(torch.bincount(tr_slot, minlength=long_rows.numel()), 0)
        YA = torch.zeros(I, 64, device=dev)
        for nf, waves, wgs in ((4, 4, 256), (8, 4, 256), (4, 8, 128)):
            def run():
                rc = LAB.lab_spmm_blocked(nf, waves, wgs, _hip.ptr(tptr), n_blocks, _hip.ptr(task_beg), _hip.ptr(task_len), _hip.ptr(col_s),
                                          _hip.ptr(val_s), _hip.ptr(X_u), _hip.ptr(part), _hip.stream())
                assert rc == 0
            tA = timeit(run)

            def red():
                rc = LAB.lab_reduce_parts(long_rows.numel(), _hip.ptr(long_rows), _hip.ptr(rt_ptr), _hip.ptr(rt_ids), _hip.ptr(part), _hip.ptr(YA),
                                          _hip.stream())
                assert rc == 0
            tR = timeit(red)
            emit(kind="itemctl", T=T, CBLK=CBLK, tasks=n_tasks, nf=nf, waves=waves, wgs_per_xcd=wgs, ms=round(tA, 4), reduce_ms=round(tR, 4),
                 gather_TBps=round(nA * 256 / tA / 1e9, 2), max_err=float((YA - YA_ref).abs().max()))

LAB.lab_spmm_rows_deep.restype = i32
LAB.lab_spmm_rows_deep.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp]
LAB.lab_spmm_blocked_deep.restype = i32
LAB.lab_spmm_blocked_deep.argtypes = [i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp]
LAB.lab_reduce_parts4.restype = i32
LAB.lab_reduce_parts4.argtypes = [i32, vp, vp, vp, vp, vp, vp]


def lab_deep(waves, grid, n_rows, rp, col, val, X, Y):
    rc = LAB.lab_spmm_rows_deep(waves, grid, n_rows, _hip.ptr(rp), _hip.ptr(col), _hip.ptr(val), _hip.ptr(X), _hip.ptr(Y), _hip.stream())
    assert rc == 0, rc


if "deep" in WHAT:
    Y = torch.empty(U, 64, device=dev)
    for waves, grid in ((4, 1024), (4, 2048), (4, 4096), (8, 512), (8, 1024), (2, 2048), (2, 4096)):
        Y.zero_()
        t = timeit(lambda: lab_deep(waves, grid, U, ui_rp, ui_col, ui_val, X_i, Y))
        emit(kind="user_deep", waves=waves, grid=grid, ms=round(t, 4), gather_TBps=round(nnz * 256 / t / 1e9, 2),
             max_err=float((Y - Yu_ref).abs().max()))
    lens = iu_rp[1:] - iu_rp[:-1]
    rows_all = torch.repeat_interleave(torch.arange(I, device=dev), lens)
    for T, CBLKS in ((512, (16384, 32768)), (256, (32768, 65536)), (128, (65536, 131072))):
        isA = lens >= T
        selA = isA[rows_all]
        rpB = torch.zeros(I + 1, dtype=torch.int64, device=dev)
        rpB[1:] = torch.cumsum(torch.where(isA, torch.zeros_like(lens), lens), 0)
        colB, valB = iu_col[~selA].contiguous(), iu_val[~selA].contiguous()
        nB = int(rpB[-1])
        YB, YB_ref = torch.empty(I, 64, device=dev), torch.empty(I, 64, device=dev)
        product(I, rpB, colB, valB, X_u, YB_ref, nB)
        bestB = None
        for waves, grid in ((4, 1024), (4, 2048), (8, 1024)):
            tB = timeit(lambda: lab_deep(waves, grid, I, rpB, colB, valB, X_u, YB))
            emit(kind="itemB_deep", T=T, nnz=nB, share=round(nB / nnz, 4), waves=waves, grid=grid, ms=round(tB, 4),
                 gather_TBps=round(nB * 256 / tB / 1e9, 2), max_err=float((YB - YB_ref).abs().max()))
            bestB = tB if bestB is None else min(bestB, tB)
        rA, cA, vA = rows_all[selA], iu_col[selA].long(), iu_val[selA]
        nA = rA.numel()
        YA_ref = torch.zeros(I, 64, device=dev)
        YA_ref[isA] = Yi_ref[isA]
        for CBLK in CBLKS:
            n_blocks = (U + CBLK - 1) // CBLK
            blk = cA // CBLK
            order = torch.argsort((blk * I + rA) * U + cA)
            col_s, val_s = cA[order].int().contiguous(), vA[order].contiguous()
            seg_key = (blk * I + rA)[order]
            uniq, counts = torch.unique_consecutive(seg_key, return_counts=True)
            n_t = (counts + 255) // 256
            seg_start = torch.cumsum(counts, 0) - counts
            task_seg = torch.repeat_interleave(torch.arange(uniq.numel(), device=dev), n_t)
            n_tasks = task_seg.numel()
            first_task = torch.cumsum(n_t, 0) - n_t
            idx_in_seg = torch.arange(n_tasks, device=dev) - first_task[task_seg]
            task_beg = (seg_start[task_seg] + idx_in_seg * 256).contiguous()
            task_len = torch.minimum(torch.full_like(idx_in_seg, 256), counts[task_seg] - idx_in_seg * 256).int().contiguous()
            task_row = (uniq % I)[task_seg]
            task_blk = torch.div(uniq, I, rounding_mode="floor")[task_seg]
            tptr = torch.zeros(n_blocks + 1, dtype=torch.int64, device=dev)
            tptr[1:] = torch.cumsum(torch.bincount(task_blk, minlength=n_blocks), 0)
            part = torch.zeros(n_tasks, 64, device=dev)
            long_rows = torch.nonzero(isA).flatten().int().contiguous()
            slot = torch.full((I,), -1, dtype=torch.int64, device=dev)
            slot[long_rows.long()] = torch.arange(long_rows.numel(), device=dev)
            tr_slot = slot[task_row]
            rt_ids = torch.argsort(tr_slot * (1 << 22) + torch.arange(n_tasks, device=dev), stable=True).int().contiguous()
            rt_ptr = torch.zeros(long_rows.numel() + 1, dtype=torch.int64, device=dev)
            rt_ptr[1:] = torch.cumsum(torch.bincount(tr_slot, minlength=long_rows.numel()), 0)
            YA = torch.zeros(I, 64, device=dev)
            for waves, wgs in ((4, 128), (4, 256), (8, 64)):
                def run():
                    rc = LAB.lab_spmm_blocked_deep(waves, wgs, _hip.ptr(tptr), n_blocks, _hip.ptr(task_beg), _hip.ptr(task_len), _hip.ptr(col_s),
                                                   _hip.ptr(val_s), _hip.ptr(X_u), _hip.ptr(part), _hip.stream())
                    assert rc == 0
                tA = timeit(run)

                def red():
                    rc = LAB.lab_reduce_parts4(long_rows.numel(), _hip.ptr(long_rows), _hip.ptr(rt_ptr), _hip.ptr(rt_ids), _hip.ptr(part),
                                               _hip.ptr(YA), _hip.stream())
                    assert rc == 0
                tR = timeit(red)
                emit(kind="itemA_deep", T=T, CBLK=CBLK, rowsA=int(isA.sum()), share=round(nA / nnz, 4), tasks=n_tasks, waves=waves, wgs_per_xcd=wgs,
                     ms=round(tA, 4), reduce_ms=round(tR, 4), gather_TBps=round(nA * 256 / tA / 1e9, 2),
                     max_err=float((YA - YA_ref).abs().max()), item_total_ms=round(tA + tR + bestB, 4))

if "plan" in WHAT:
    from skrec.recommender.LightGCN import DeviceCSR, build_adjacency_device

    def mk(rp, col, val, shape):
        c = DeviceCSR.__new__(DeviceCSR)
        c.shape, c.nnz, c.rowptr, c.col, c.val = shape, int(rp[-1]), rp, col, val
        return c
    for name, c, X, Yref in (("user", mk(ui_rp, ui_col, ui_val, (U, I)), X_i, Yu_ref), ("item", mk(iu_rp, iu_col, iu_val, (I, U)), X_u, Yi_ref)):
        Y = torch.empty(c.shape[0], 64, device=dev)
        os.environ["SKR_SPMM_PLAN"] = "1"
        c.spmm(X, Y)
        t = timeit(lambda: c.spmm(X, Y))
        emit(kind="plan", side=name, ms=round(t, 4), gather_TBps=round(nnz * 256 / t / 1e9, 2), info=c.plan_info(),
             max_err=float((Y - Yref).abs().max()))
    adj, _ = build_adjacency_device(ds["users"], ds["items"], U, I, "pre", dev)
    N = U + I
    Xs = torch.cat([X_u, X_i])
    Ys, Y0 = torch.empty(N, 64, device=dev), torch.empty(N, 64, device=dev)
    os.environ["SKR_SPMM_PLAN"] = "0"
    t0 = timeit(lambda: adj.spmm(Xs, Y0))
    os.environ["SKR_SPMM_PLAN"] = "1"
    adj.spmm(Xs, Ys)
    t1 = timeit(lambda: adj.spmm(Xs, Ys))
    acc = torch.zeros_like(Ys)
    t2 = timeit(lambda: adj.spmm(Xs, Ys, addend=Xs, accum=acc, accum_scale=0.25))
    alg = adj.nnz * 8 + (N + 1) * 8 + 2 * N * 256
    emit(kind="plan_square", nnz=adj.nnz, round1_ms=round(t0, 4), plan_ms=round(t1, 4), plan_epilogue_ms=round(t2, 4), speedup=round(t0 / t1, 3),
         alg_GBps=round(alg / t1 / 1e6, 1), frac_hbm=round(alg / t1 / 1e6 / 8000, 4), info=adj.plan_info(), max_err=float((Ys - Y0).abs().max()))
