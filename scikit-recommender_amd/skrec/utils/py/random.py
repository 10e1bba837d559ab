"""``randint_choice`` / ``batch_randint_choice`` -- API of the reference's utils/py/random.py backed by
the HIP sampler (csrc/sampler.hip) instead of the Cython extension.

One process-global MT19937 stream seeded with 2020 mirrors the reference's ``std::mt19937 _gen(2020)``
(utils/py/cython/include/randint.h:20): every call in the process -- these functions and the epoch
samplers of ``skrec.io`` -- consumes it in call order, so a fresh process replays the reference's
numbers.  The stream lives in HBM; there is no CPU implementation behind this module.
"""
import ctypes as C

import numpy as np

from ... import _hip

__all__ = ["randint_choice", "batch_randint_choice", "DeviceSampler", "global_sampler", "reset_global_sampler"]


class DeviceSampler(object):
    """Owner of one ``skr_sampler`` handle (include/skrec_hip.h)."""

    def __init__(self, seed=2020):
        _hip.require_gpu()
        self._h = C.c_void_p()
        _hip.check(_hip.lib().skr_sampler_create(int(seed), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _hip.lib().skr_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def get_state(self):
        words = np.zeros(624, np.uint32)
        pos = C.c_int(0)
        _hip.check(_hip.lib().skr_sampler_get_state(self._h, words.ctypes.data, C.byref(pos)))
        return words, pos.value

    def set_state(self, words, pos):
        words = np.ascontiguousarray(words, np.uint32)
        _hip.check(_hip.lib().skr_sampler_set_state(self._h, words.ctypes.data, int(pos)))

    @property
    def draws(self):
        n = C.c_uint64(0)
        _hip.check(_hip.lib().skr_sampler_draws(self._h, C.byref(n)))
        return int(n.value)

    def randint_choice(self, high, size, replace, p, exclusion):
        """device tensors in, device int32 tensor out; arguments already validated"""
        import torch
        dev = _hip.require_gpu()
        out = torch.empty(size, dtype=torch.int32, device=dev)
        d_p = torch.as_tensor(p, dtype=torch.float32, device=dev).contiguous() if p is not None else None
        d_ex, n_ex = None, 0
        if exclusion is not None and len(exclusion) > 0:
            ex = np.unique(np.asarray(exclusion, dtype=np.int64)).astype(np.int32)  # sorted, unique
            d_ex, n_ex = torch.from_numpy(ex).to(dev), len(ex)
        _hip.check(_hip.lib().skr_randint_choice(self._h, int(high), int(size), int(bool(replace)), _hip.ptr(d_p),
                                                 _hip.ptr(d_ex), n_ex, _hip.ptr(out), _hip.stream()))
        return out

    def sample_epoch_exact(self, num_items, n_users, d_rowptr, d_pos_sorted, nnz, num_neg, d_out):
        """one epoch of negatives on the reference's stream.  The CSR's row statistics are taken on the first call with a
        given ``d_rowptr`` (same storage, same length, not written to since: torch's version counter) and kept: every later
        call only queues work -- no read-back, the host does not wait for the device."""
        key = (d_rowptr.data_ptr(), int(n_users), int(nnz), d_rowptr._version)
        cache = self.__dict__.setdefault("_stats", {})
        hit = cache.get(key)
        if hit is None:
            stats = (C.c_int64 * 2)()
            _hip.check(_hip.lib().skr_csr_row_stats(_hip.ptr(d_rowptr), int(n_users), stats, _hip.stream()))
            if len(cache) >= 64:
                cache.clear()
            # the tensor is kept with its statistics: while the entry lives, the storage's address cannot go to another tensor
            hit = cache[key] = (stats, d_rowptr)
        stats = hit[0]
        _hip.check(_hip.lib().skr_sample_epoch_exact_stats(self._h, int(num_items), int(n_users), _hip.ptr(d_rowptr),
                                                           _hip.ptr(d_pos_sorted), int(nnz), int(num_neg), _hip.ptr(d_out), stats,
                                                           _hip.stream()))

    def last_epoch(self):
        """how the last exact epoch ran: dict(status, handed_over, filled, consumed) -- see skr_sampler_last_epoch"""
        info = (C.c_int64 * 4)()
        _hip.check(_hip.lib().skr_sampler_last_epoch(self._h, info))
        return dict(status=int(info[0]), handed_over=int(info[1]), filled=int(info[2]), consumed=int(info[3]))

    def sample_epoch_exact_counts(self, num_items, n_users, d_rowptr, d_excl_sorted, nnz, d_drawptr, n_draws, d_out):
        """exclusion CSR and per-user draw counts given separately (sequential / knowledge-graph iterators)"""
        _hip.check(_hip.lib().skr_sample_epoch_exact_counts(self._h, int(num_items), int(n_users), _hip.ptr(d_rowptr),
                                                            _hip.ptr(d_excl_sorted), int(nnz), _hip.ptr(d_drawptr),
                                                            int(n_draws), _hip.ptr(d_out), _hip.stream()))


_global = None


def global_sampler() -> DeviceSampler:
    global _global
    if _global is None:
        _global = DeviceSampler(2020)
    return _global


def reset_global_sampler(seed=2020):
    """Start the process-global stream over (the reference can only do this by restarting Python)."""
    global _global
    if _global is not None:
        _global.close()
    _global = DeviceSampler(seed)
    return _global


def randint_choice(high, size=1, replace=True, p=None, exclusion=None):
    """Sample ``size`` integers from ``[0, high)`` (reference: pyx_random.pyx:20-76, same checks,
    same exceptions, scalar when ``size == 1``)."""
    if high <= 1:
        raise ValueError("'high' must be larger than 1.")
    if size <= 0:
        raise ValueError("'size' must be a positive integer.")
    if not isinstance(replace, bool):
        raise TypeError("'replace' must be bool.")
    if p is not None:
        p = np.array(p, dtype=np.float32)
        if p.ndim != 1:
            raise ValueError("'p' must be a 1-dim array_like")
        if len(p) != high:
            raise ValueError("The length of 'p' must be equal with 'high'.")
    if isinstance(exclusion, (int, np.integer)):
        exclusion = [int(exclusion)]
    if exclusion is not None and len(exclusion) >= high:
        raise ValueError("The length of 'exclusion' must be smaller than 'high'.")
    n_ex = len(exclusion) if exclusion is not None else 0
    if replace is False and (high - n_ex <= size):
        raise ValueError("There is not enough integers to be sampled.")
    out = global_sampler().randint_choice(high, size, replace, p, exclusion).cpu().numpy()
    return out[0] if len(out) == 1 else out


def batch_randint_choice(high, size, replace=True, p=None, exclusion=None, thread_num=1):
    """One ``randint_choice`` per row, in row order, from the single global stream (reference:
    pyx_random.pyx:79-149 with thread_num=1; its thread_num>1 path races on the generator and has
    no defined result, so ``thread_num`` is validated and otherwise ignored)."""
    if high <= 1:
        raise ValueError("'high' must be larger than 1.")
    if not isinstance(replace, bool):
        raise TypeError("'replace' must be bool.")
    if not isinstance(thread_num, (int, np.integer)) or thread_num < 1:
        raise ValueError("'thread_num' must be a positive integer.")
    try:
        size = np.array(size, np.int32)
    except Exception:
        raise ValueError("'size' must be a 1-dim array_like of positive integers.")
    if size.ndim != 1 or np.any(size <= 0):
        raise ValueError("'size' must be a 1-dim array_like of positive integers.")
    if p is not None:
        p = np.array(p, dtype=np.float32)
        if p.ndim != 2:
            raise ValueError("'p' must be a 2-dim array_like.")
        if p.shape[0] != len(size):
            raise ValueError("The number of rows of 'p' must be equal with the length of 'size'.")
        if p.shape[1] != high:
            raise ValueError("The number of columns of 'p' must be equal with 'high'.")
    if exclusion is not None:
        if len(exclusion) != len(size):
            raise ValueError("The length of 'exclusion' must be equal with the length of 'size'.")
        for idx, (exc, s) in enumerate(zip(exclusion, size)):
            if len(exc) >= high:
                raise ValueError("The length of 'exclusion' must be smaller than 'high' in %d-th row." % idx)
            if replace is False and (high - len(exc) <= s):
                raise ValueError("There is not enough integers to be sampled in %d-th row." % idx)
    g = global_sampler()
    out = []
    for r, s in enumerate(size):
        row = g.randint_choice(high, int(s), replace, None if p is None else p[r],
                               None if exclusion is None else exclusion[r])
        out.append(row.cpu().numpy())
    return out
