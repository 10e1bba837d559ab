"""Shared helpers of the test-suite (CPU side)."""
import numpy as np


def csr_from_pairs(users, items, n_users):
    """(rowptr, items in file order per user, items sorted per user, users_ary) like ImplicitFeedback.to_csr_arrays"""
    users = np.asarray(users, np.int64)
    items = np.asarray(items, np.int32)
    order = np.argsort(users, kind="stable")
    rowptr = np.zeros(n_users + 1, np.int64)
    rowptr[1:] = np.cumsum(np.bincount(users, minlength=n_users))
    file_order = items[order]
    srt = items[np.lexsort((items, users))]
    uary = np.repeat(np.arange(n_users, dtype=np.int32), np.diff(rowptr))
    return rowptr, np.ascontiguousarray(file_order), np.ascontiguousarray(srt), uary


def random_csr(rng, n_users, n_items, lo, hi, empty_frac=0.1):
    lens = rng.integers(lo, hi + 1, n_users)
    lens[rng.random(n_users) < empty_frac] = 0
    lens = np.minimum(lens, n_items - 1)
    rowptr = np.zeros(n_users + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    rows = [np.sort(rng.choice(n_items, int(l), replace=False)).astype(np.int32) for l in lens]
    items = np.concatenate(rows) if rowptr[-1] else np.zeros(0, np.int32)
    return rowptr, np.ascontiguousarray(items, np.int32)


def tiny_arrays(golden):
    d = golden("tiny_dataset")
    tr = d["train"]
    U, I = int(d["num_users"]), int(d["num_items"])
    rowptr, file_order, srt, uary = csr_from_pairs(tr[:, 0], tr[:, 1], U)
    return U, I, rowptr, file_order, srt, uary


def lists_from_csr(rowptr, items):
    return [items[rowptr[i]:rowptr[i + 1]] for i in range(len(rowptr) - 1)]

