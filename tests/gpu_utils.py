"""Thin wrappers used by the GPU parity tests: every call goes through the C ABI (skrec._hip)."""
import ctypes as C

import numpy as np
import torch

from skrec import _hip


def dev():
    return _hip.require_gpu()


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def eval_scores(scores, test_lists, metric_ids, K, want_ids=True):
    d = dev()
    sc = to_dev(np.ascontiguousarray(scores, np.float32))
    B, I = sc.shape
    rowptr = np.zeros(B + 1, np.int64)
    rowptr[1:] = np.cumsum([len(np.unique(t)) for t in test_lists])
    items = np.concatenate([np.unique(np.asarray(t, np.int64)).astype(np.int32) for t in test_lists]) \
        if rowptr[-1] else np.zeros(1, np.int32)
    d_ptr, d_items = to_dev(rowptr), to_dev(items.astype(np.int32))
    nm = len(metric_ids)
    rows = torch.zeros((B, nm * K), dtype=torch.float32, device=d)
    ids = torch.zeros((B, K), dtype=torch.int32, device=d)
    sums = torch.zeros(nm * K, dtype=torch.float64, device=d)
    _hip.check(_hip.lib().skr_eval_scores(_hip.ptr(sc), B, I, I, _hip.ptr(d_ptr), _hip.ptr(d_items),
                                          _hip.metric_array(metric_ids), nm, K, _hip.ptr(rows),
                                          _hip.ptr(ids) if want_ids else None, _hip.ptr(sums), _hip.stream()))
    torch.cuda.synchronize()
    return rows.cpu().numpy(), ids.cpu().numpy(), sums.cpu().numpy()


def fused_topk(user_table, users, item_table, bias, train_rowptr, train_items, K):
    d = dev()
    ut, it = to_dev(user_table), to_dev(item_table)
    bs = to_dev(bias) if bias is not None else None
    du = to_dev(np.asarray(users, np.int32))
    B = len(users)
    tp = to_dev(train_rowptr) if train_rowptr is not None else None
    ti = to_dev(train_items if len(train_items) else np.zeros(1, np.int32)) if train_rowptr is not None else None
    ids = torch.full((B, K), -7, dtype=torch.int32, device=d)
    sc = torch.zeros((B, K), dtype=torch.float32, device=d)
    ws = int(_hip.lib().skr_eval_fused_workspace(B, K))
    work = torch.empty(ws, dtype=torch.uint8, device=d)
    _hip.check(_hip.lib().skr_eval_fused_topk(_hip.ptr(ut), _hip.ptr(du), B, _hip.ptr(it), _hip.ptr(bs),
                                              it.shape[0], 64, _hip.ptr(tp), _hip.ptr(ti), K, _hip.ptr(ids), _hip.ptr(sc),
                                              _hip.ptr(work), ws, _hip.stream()))
    torch.cuda.synchronize()
    return ids.cpu().numpy(), sc.cpu().numpy()


class ExactSampler:
    def __init__(self, seed=2020):
        from skrec.utils.py.random import DeviceSampler
        self.s = DeviceSampler(seed)

    def epoch(self, num_items, rowptr, pos_sorted, num_neg):
        nnz = int(rowptr[-1])
        d_ptr, d_pos = to_dev(rowptr), to_dev(pos_sorted if nnz else np.zeros(1, np.int32))
        out = torch.full((max(nnz * num_neg, 1),), -5, dtype=torch.int32, device=dev())
        self.s.sample_epoch_exact(num_items, len(rowptr) - 1, d_ptr, d_pos, nnz, num_neg, out)
        torch.cuda.synchronize()
        return out.cpu().numpy()[:nnz * num_neg]


def fast_epoch(seed, epoch, slot_offset, num_items, rowptr, pos_sorted, num_neg):
    nnz = int(rowptr[-1])
    d_ptr, d_pos = to_dev(rowptr), to_dev(pos_sorted)
    out = torch.full((nnz * num_neg,), -5, dtype=torch.int32, device=dev())
    _hip.check(_hip.lib().skr_sample_epoch_fast(seed, epoch, slot_offset, num_items, len(rowptr) - 1, _hip.ptr(d_ptr),
                                                _hip.ptr(d_pos), nnz, num_neg, _hip.ptr(out), _hip.stream()))
    torch.cuda.synchronize()
    return out.cpu().numpy()
