"""Training-instance iterators of the hot path, HIP-backed (reference: skrec/io/data_iterator.py).

``PairwiseIterator`` / ``PointwiseIterator`` keep the reference's constructor, ``len()`` and the
tuples they yield (numpy int32 / float32 arrays), so existing training loops run unchanged.  What
changed underneath:

* the per-user Python loop over ``randint_choice`` (data_iterator.py:81-94) is ONE kernel sequence
  per epoch over a CSR of the train positives resident in HBM (``skr_sample_epoch_exact`` replays
  the reference's MT19937 stream bit for bit; ``sampler_mode="fast"`` uses the slot-keyed
  xoshiro128++ kernel);
* the epoch arrays stay on the device; ``iter_device()`` hands out int32 device slices so that the
  in-scope models never copy a batch through the host.  The shuffle contract is the reference's: one
  ``np.random.permutation(E)`` from numpy's global generator per ``__iter__``
  (utils/py/batch_iterator.py:61-63), consecutive slices of ``batch_size``.
"""
import os
from collections import OrderedDict

import numpy as np

from .. import _hip
from ..utils.py.random import global_sampler
from .dataset import ImplicitFeedback

__all__ = ["PointwiseIterator", "PairwiseIterator", "InteractionIterator",
           "PairwiseSampler", "PointwiseSampler",
           "SequentialPointwiseIterator", "SequentialPairwiseIterator",
           "UserVecIterator", "ItemVecIterator", "KGPairwiseIterator"]


def _default_mode(mode):
    mode = mode or os.environ.get("SKREC_SAMPLER", "exact")
    if mode not in ("exact", "fast"):
        raise ValueError("sampler_mode must be 'exact' or 'fast'")
    return mode


class _EpochSampler(object):
    """CSR of the train positives on the device + the S1 arrays (data_iterator.py:30-42)."""

    def __init__(self, dataset: ImplicitFeedback, num_neg, sampler_mode=None, seed=2020):
        import torch
        if num_neg <= 0:
            raise ValueError("'num_neg' must be a positive integer.")
        rowptr, items_file, items_sorted = dataset.to_csr_arrays()
        assert len(items_file) > 0, "'user_pos_dict' cannot be empty."
        self.num_items = int(dataset.num_items)
        self.n_users = len(rowptr) - 1
        self.nnz = int(rowptr[-1])
        lens = np.diff(rowptr)
        if self.num_items <= 1:
            raise ValueError("'high' must be larger than 1.")
        if lens.max() >= self.num_items:  # pyx_random.pyx:49, raised at the first offending user
            raise ValueError("The length of 'exclusion' must be smaller than 'high'.")
        self.num_neg = int(num_neg)
        self.mode = _default_mode(sampler_mode)
        self.seed = int(seed)
        self.epoch = 0
        self.user_n_pos = OrderedDict((int(u), int(lens[u])) for u in np.flatnonzero(lens))
        self.users_ary = np.repeat(np.arange(self.n_users, dtype=np.int32), lens)
        self.pos_items = items_file
        self.dev = _hip.require_gpu()
        self.d_rowptr = torch.from_numpy(rowptr).to(self.dev)
        self.d_pos_sorted = torch.from_numpy(items_sorted).to(self.dev)
        self.d_users = torch.from_numpy(self.users_ary).to(self.dev)
        self.d_pos = torch.from_numpy(self.pos_items).to(self.dev)

    def sample(self):
        """-> device int32 [E*num_neg]: negatives for one epoch (slot-major, [E, num_neg] row-major)."""
        import torch
        out = torch.empty(self.nnz * self.num_neg, dtype=torch.int32, device=self.dev)
        if self.mode == "exact":
            global_sampler().sample_epoch_exact(self.num_items, self.n_users, self.d_rowptr, self.d_pos_sorted,
                                                self.nnz, self.num_neg, out)
        else:
            _hip.check(_hip.lib().skr_sample_epoch_fast(self.seed, self.epoch, 0, self.num_items, self.n_users,
                                                        _hip.ptr(self.d_rowptr), _hip.ptr(self.d_pos_sorted),
                                                        self.nnz, self.num_neg, _hip.ptr(out), _hip.stream()))
        self.epoch += 1
        return out


def _n_batches(n, batch_size, drop_last):
    return n // batch_size if drop_last else (n + batch_size - 1) // batch_size


def _host_permutation(n):
    """np.random.permutation(n) -- numpy's values and generator state, computed by the native library outside the GIL
    (int32; csrc/host_random.hip)"""
    from .. import _hip
    return _hip.host_permutation(n)


def _shuffled_columns(columns, shuffle, device_shuffle=False, host_perm=None):
    """whole epoch columns, shuffled once on the device (contract: see _device_batches); ``host_perm``: the epoch's
    np.random.permutation(n), if the caller has drawn it already"""
    import torch
    n = columns[0].shape[0]
    if shuffle:
        dev = columns[0].device
        columns = [c.contiguous() for c in columns]
        if device_shuffle:
            # the permutation is a keyed bijection evaluated inside the gather kernel; numpy's global
            # generator supplies the key (one integer per epoch), so np.random.seed still fixes the run
            columns = _hip.shuffle_gather(columns, None, seed=int(np.random.randint(0, 2 ** 31 - 1)))
        else:
            perm = torch.from_numpy(_host_permutation(n) if host_perm is None else host_perm).to(dev)
            columns = _hip.shuffle_gather(columns, perm)
    return columns


class _EpochAhead(object):
    """Next epoch's negatives and permutation, produced while the GPU trains on the current epoch.

    The reference starts every epoch with (a) the per-user negative sampling loop on the global std::mt19937 stream
    (data_iterator.py:81-94) and (b) ONE np.random.permutation(n) from numpy's global generator
    (batch_iterator.py:61-63).  Here (a) is a serial chain on ONE compute unit (0.32 s per 48 M draws, the other 255
    idle) and (b) ~0.35 s of native host code: a third of an epoch between them.  Inside a training loop that is known
    to ask neither generator for anything else between epochs (``fit()``: train, evaluate, train ...), the next
    epoch's draws are the next thing both will be asked for, so they can be made right after the current ones -- on a
    helper thread (both native calls run outside the GIL) and a side stream.  Same numbers, same order.  ``close()``
    puts both generators back to where the reference's would be if what was drawn ahead is never used (early stop,
    last epoch)."""

    def __init__(self, it):
        import torch
        self.it, self._thread, self._out, self._err, self._saved = it, None, None, None, None
        self._side = torch.cuda.Stream(device=it._s.dev)

    def _produce(self, stream=None):
        import torch
        it = self.it
        if stream is not None:
            with torch.cuda.stream(stream):
                neg = it._s.sample()
        else:
            neg = it._s.sample()
        perm = _host_permutation(len(it.all_users)) if (it.shuffle and not it.device_shuffle) else None
        return neg, perm

    def _work(self):
        try:
            self._out = self._produce(self._side)
        except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread by take()
            self._err = e

    def take(self):
        """this epoch's (negatives, permutation or None); starts producing the next epoch's"""
        import threading
        import torch
        if self._thread is not None:
            self._thread.join()
            self._thread = None
            if self._err is not None:
                raise self._err
            neg, perm = self._out
            torch.cuda.current_stream().wait_stream(self._side)   # the fast sampler only ENQUEUES on the side stream
            neg.record_stream(torch.cuda.current_stream())     # allocated under the side stream, consumed here
        else:
            neg, perm = self._produce()
        s = self.it._s
        self._saved = (np.random.get_state(), global_sampler().get_state() if s.mode == "exact" else None, s.epoch)
        self._out = self._err = None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()
        return neg, perm

    def close(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
            np_state, sampler_state, epoch = self._saved
            np.random.set_state(np_state)
            if sampler_state is not None:
                global_sampler().set_state(*sampler_state)
            self.it._s.epoch = epoch
            self._out = None


def _device_batches(columns, batch_size, shuffle, drop_last, device_shuffle=False):
    """Shuffle whole epoch columns once on the device, then yield consecutive slices.
    Default: the reference's contract -- ONE np.random.permutation(n) from numpy's global generator per
    epoch (batch_iterator.py:61-63), uploaded.  ``device_shuffle=True`` evaluates a
    keyed bijection inside the gather kernel instead (skr_shuffle_gather, no permutation array); numpy's global generator
    still seeds it (one integer per epoch), so ``np.random.seed`` keeps runs reproducible, but the order differs from the
    reference's.  Either way all columns move in ONE launch of the library's own kernel."""
    n = columns[0].shape[0]
    columns = _shuffled_columns(columns, shuffle, device_shuffle)
    for start in range(0, n, batch_size):
        stop = min(start + batch_size, n)
        if stop - start < batch_size and drop_last:
            return
        yield tuple(c[start:stop] for c in columns)


class PairwiseIterator(object):
    """(users, pos_items, neg_items) batches; ``neg_items`` is [b, num_neg] when ``num_neg > 1``
    (reference: data_iterator.py:191-234)."""

    def __init__(self, dataset: ImplicitFeedback, num_neg: int = 1, batch_size: int = 1024, shuffle: bool = True,
                 drop_last: bool = False, sampler_mode: str = None, seed: int = 2020, device_shuffle: bool = False):
        self._s = _EpochSampler(dataset, num_neg, sampler_mode, seed)
        self.batch_size, self.shuffle, self.drop_last, self.num_neg = batch_size, shuffle, drop_last, num_neg
        self.device_shuffle = device_shuffle
        self.num_items = self._s.num_items
        self.user_n_pos = self._s.user_n_pos
        self.all_users, self.pos_items = self._s.users_ary, self._s.pos_items
        self._ahead = None

    def __len__(self):
        return _n_batches(len(self.all_users), self.batch_size, self.drop_last)

    def epoch_ahead(self, on=True):
        """``True``: from now on epoch_columns() produces the NEXT epoch's negatives and permutation in the background
        (see _EpochAhead: only valid while nothing else uses the global sampler stream or numpy's global generator
        between epochs).  ``False``: stop, and rewind both generators over what was drawn ahead but never used."""
        if on and self._ahead is None:
            self._ahead = _EpochAhead(self)
        elif not on and self._ahead is not None:
            self._ahead.close()
            self._ahead = None

    def iter_device(self):
        neg = self._s.sample()
        if self.num_neg > 1:
            neg = neg.view(-1, self.num_neg)
        return _device_batches([self._s.d_users, self._s.d_pos, neg], self.batch_size, self.shuffle, self.drop_last,
                               self.device_shuffle)

    def epoch_columns(self):
        """One epoch as whole device columns: (users, pos, neg) after sampling and shuffling, plus the batch
        boundaries [(start, stop), ...] that iter_device() would walk.  Consumes the sampler stream and numpy's
        permutation exactly like one ``iter_device()`` pass; lets a trainer look k batches ahead."""
        neg, perm = self._ahead.take() if self._ahead is not None else (self._s.sample(), None)
        if self.num_neg > 1:
            neg = neg.view(-1, self.num_neg)
        cols = _shuffled_columns([self._s.d_users, self._s.d_pos, neg], self.shuffle, self.device_shuffle, perm)
        n, b = cols[0].shape[0], self.batch_size
        bounds = [(s, min(s + b, n)) for s in range(0, n, b) if not (self.drop_last and s + b > n)]
        return cols, bounds

    def __iter__(self):
        for u, i, j in self.iter_device():
            yield u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()


class PointwiseIterator(object):
    """(users, items, labels) batches: positives labelled 1.0 followed by the negatives labelled 0.0,
    negative-slot-major (reference: data_iterator.py:125-188)."""

    def __init__(self, dataset: ImplicitFeedback, num_neg: int = 1, batch_size: int = 1024, shuffle: bool = True,
                 drop_last: bool = False, sampler_mode: str = None, seed: int = 2020):
        import torch
        assert num_neg > 0, "'num_neg' must be a positive integer."
        self._s = _EpochSampler(dataset, num_neg, sampler_mode, seed)
        self.batch_size, self.shuffle, self.drop_last, self.num_neg = batch_size, shuffle, drop_last, num_neg
        self.num_items = self._s.num_items
        self.user_n_pos = self._s.user_n_pos
        self.pos_items = self._s.pos_items
        self.all_users = np.tile(self._s.users_ary, num_neg + 1)
        n_pos = len(self.pos_items)
        self.all_labels = np.concatenate([np.ones(n_pos, np.float32), np.zeros(n_pos * num_neg, np.float32)])
        self._d_all_users = self._s.d_users.repeat(num_neg + 1)
        self._d_labels = torch.from_numpy(self.all_labels).to(self._s.dev)

    def __len__(self):
        return _n_batches(len(self.all_users), self.batch_size, self.drop_last)

    def iter_device(self):
        import torch
        neg = self._s.sample().view(-1, self.num_neg)          # [E, num_neg]
        neg = neg.t().contiguous().view(-1)                     # data_iterator.py:180
        items = torch.cat([self._s.d_pos, neg], dim=0)
        return _device_batches([self._d_all_users, items, self._d_labels], self.batch_size, self.shuffle,
                               self.drop_last)

    def __iter__(self):
        for u, i, l in self.iter_device():
            yield u.cpu().numpy(), i.cpu().numpy(), l.cpu().numpy()


class InteractionIterator(object):
    """(users, items) batches without negatives (reference: data_iterator.py:97-122)."""

    def __init__(self, dataset: ImplicitFeedback, batch_size: int = 1024, shuffle: bool = True, drop_last: bool = False):
        pairs = dataset.to_user_item_pairs()
        self.users, self.pos_items = pairs[:, 0], pairs[:, 1]
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last

    def __len__(self):
        return _n_batches(len(self.users), self.batch_size, self.drop_last)

    def __iter__(self):
        n = len(self.users)
        order = np.random.permutation(n) if self.shuffle else np.arange(n)
        for start in range(0, n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            yield self.users[idx], self.pos_items[idx]


# the names BASELINE.json uses for the same classes
PairwiseSampler = PairwiseIterator
PointwiseSampler = PointwiseIterator


class _GroupSampler(object):
    """Exact-stream negatives when a group's draw count is not the size of its exclusion set: the reference's
    loop `for key, n in counts.items(): randint_choice(high, size=n * k, exclusion=sets[key])`
    (data_iterator.py:81-94, :406-420) as one `skr_sample_epoch_exact_counts` call."""

    def __init__(self, high, exclusion_dict, counts, per_group):
        import torch
        if high <= 1:
            raise ValueError("'high' must be larger than 1.")
        keys = list(counts.keys())
        rows = [np.unique(np.asarray(exclusion_dict[k], dtype=np.int32)) for k in keys]
        lens = np.array([len(r) for r in rows], dtype=np.int64)
        if len(lens) and lens.max() >= high:   # pyx_random.pyx:49
            raise ValueError("The length of 'exclusion' must be smaller than 'high'.")
        self.high, self.n_groups = int(high), len(keys)
        rowptr = np.zeros(self.n_groups + 1, np.int64)
        np.cumsum(lens, out=rowptr[1:])
        draws = np.array([counts[k] for k in keys], dtype=np.int64) * int(per_group)
        drawptr = np.zeros(self.n_groups + 1, np.int64)
        np.cumsum(draws, out=drawptr[1:])
        self.nnz, self.n_draws = int(rowptr[-1]), int(drawptr[-1])
        self.dev = _hip.require_gpu()
        flat = np.concatenate(rows) if rows and self.nnz else np.zeros(1, np.int32)
        self.d_rowptr = torch.from_numpy(rowptr).to(self.dev)
        self.d_excl = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.int32)).to(self.dev)
        self.d_drawptr = torch.from_numpy(drawptr).to(self.dev)

    def sample(self):
        import torch
        out = torch.empty(self.n_draws, dtype=torch.int32, device=self.dev)
        if self.n_draws:
            global_sampler().sample_epoch_exact_counts(self.high, self.n_groups, self.d_rowptr, self.d_excl, self.nnz,
                                                       self.d_drawptr, self.n_draws, out)
        return out


def _pad_pre(seqs, value, max_len):
    """pad_sequences(..., padding='pre', truncating='pre') of utils/py/generic.py for int32 lists"""
    out = np.full((len(seqs), max_len), value, dtype=np.int32)
    for r, q in enumerate(seqs):
        q = q[-max_len:]
        if len(q):
            out[r, max_len - len(q):] = q
    return out


def _time_ordered_instances(user_pos_dict, num_previous=1, num_next=1, pad=None):
    """_generative_time_order_positive_items (data_iterator.py:44-78): for every user, from the full
    history backwards, the last `num_previous + num_next` items of each prefix; shorter prefixes are kept
    (left-padded) only when `pad` is given and more than `num_next` items remain."""
    assert user_pos_dict, "'user_pos_dict' cannot be empty."
    assert num_previous >= 1
    assert num_next >= 1
    tot = num_previous + num_next
    users, seqs = [], []
    user_n_pos = OrderedDict()
    for user, items in user_pos_dict.items():
        n_items = len(items)
        stop = num_next if pad is not None else tot - 1     # prefixes of length <= stop end the walk
        n_inst = max(n_items - stop, 0)
        if n_inst == 0:
            continue
        user_n_pos[user] = n_inst
        for idx in range(n_items, stop, -1):
            seqs.append(items[max(idx - tot, 0):idx])
        users.extend([user] * n_inst)
    if pad is not None and tot > 2:
        seqs_ary = _pad_pre(seqs, pad, tot)
    else:
        seqs_ary = np.int32(seqs) if seqs else np.zeros((0, tot), np.int32)
    return user_n_pos, np.int32(users), seqs_ary[:, :num_previous], seqs_ary[:, num_previous:]


class SequentialPairwiseIterator(object):
    """(users, item_seqs, pos_next, neg_next) batches (reference: data_iterator.py:292-331)."""

    def __init__(self, dataset: ImplicitFeedback, num_previous: int = 1, num_next: int = 1, pad: int = None,
                 batch_size: int = 1024, shuffle: bool = True, drop_last: bool = False):
        import torch
        self.num_previous, self.num_next, self.pad = num_previous, num_next, pad
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.num_items = dataset.num_items
        self.user_pos_dict = dataset.to_user_dict_by_time()
        self.user_n_pos, self.all_users, seqs, nxt = _time_ordered_instances(self.user_pos_dict, num_previous, num_next, pad)
        self.all_item_seqs, self.pos_next_items = seqs.squeeze(), nxt.squeeze()
        self._g = _GroupSampler(self.num_items, self.user_pos_dict, self.user_n_pos, num_next)
        dev = self._g.dev
        self._cols = [torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                      for a in (self.all_users, self.all_item_seqs, self.pos_next_items)]

    def __len__(self):
        return _n_batches(len(self.all_users), self.batch_size, self.drop_last)

    def iter_device(self):
        neg = self._g.sample()
        if self.num_next > 1:
            neg = neg.view(-1, self.num_next)
        return _device_batches(self._cols + [neg], self.batch_size, self.shuffle, self.drop_last)

    def __iter__(self):
        for cols in self.iter_device():
            yield tuple(c.cpu().numpy() for c in cols)


class SequentialPointwiseIterator(object):
    """(users, item_seqs, next_items, labels) batches (reference: data_iterator.py:237-289).  The reference
    raises inside np.concatenate when num_neg * num_next == 1 (a 1-D negative array meets the [n, 1]
    positives, :277-278); here that case yields the squeezed 1-D columns the other shapes reduce to."""

    def __init__(self, dataset: ImplicitFeedback, num_previous: int = 1, num_next: int = 1, num_neg: int = 1,
                 pad: int = None, batch_size: int = 1024, shuffle: bool = True, drop_last: bool = False):
        import torch
        assert num_neg >= 1
        self.num_previous, self.num_next, self.num_neg, self.pad = num_previous, num_next, num_neg, pad
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.num_items = dataset.num_items
        self.user_pos_dict = dataset.to_user_dict_by_time()
        self.user_n_pos, users, seqs, self.pos_next_items = _time_ordered_instances(self.user_pos_dict, num_previous,
                                                                                    num_next, pad)
        self.all_users = np.tile(users, num_neg + 1)
        self.all_item_seqs = np.tile(seqs, [num_neg + 1, 1]).squeeze()
        n = len(self.pos_next_items)
        self.all_labels = np.concatenate([np.ones([n, num_next], np.float32),
                                          np.zeros([n * num_neg, num_next], np.float32)], axis=0).squeeze()
        self._g = _GroupSampler(self.num_items, self.user_pos_dict, self.user_n_pos, num_neg * num_next)
        dev = self._g.dev
        self._d_users, self._d_seqs, self._d_labels = (torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                                                       for a in (self.all_users, self.all_item_seqs, self.all_labels))
        self._d_pos_next = torch.from_numpy(np.ascontiguousarray(self.pos_next_items)).to(dev)

    def __len__(self):
        return _n_batches(len(self.all_users), self.batch_size, self.drop_last)

    def iter_device(self):
        import torch
        neg = self._g.sample().view(-1, self.num_neg, self.num_next)           # [n, num_neg * num_next] per user row
        neg = neg.permute(1, 0, 2).reshape(-1, self.num_next)                    # split on the last axis, stack (:275)
        nxt = torch.cat([self._d_pos_next, neg], dim=0)
        if self.num_next == 1:
            nxt = nxt.view(-1)
        return _device_batches([self._d_users, self._d_seqs, nxt, self._d_labels], self.batch_size, self.shuffle,
                               self.drop_last)

    def __iter__(self):
        for cols in self.iter_device():
            yield tuple(c.cpu().numpy() for c in cols)


class KGPairwiseIterator(object):
    """(heads, relations, pos_tails, neg_tails) batches (reference: data_iterator.py:423-457).  ``dataset`` is
    anything with ``num_entities`` and ``to_head_dict()`` -> {head: {"relation": int32[], "tail": int32[]}}."""

    def __init__(self, dataset, num_neg: int = 1, batch_size: int = 1024, shuffle: bool = True, drop_last: bool = False):
        import torch
        if num_neg <= 0:
            raise ValueError("'num_neg' must be a positive integer.")
        self.batch_size, self.shuffle, self.drop_last, self.num_neg = batch_size, shuffle, drop_last, num_neg
        self.num_entities = dataset.num_entities
        self.head_pos_dict = dataset.to_head_dict()
        assert self.head_pos_dict, "'head_pos_dict' cannot be empty."
        heads, rels, tails = [], [], []
        self.head_n_pos = OrderedDict()
        for head, d in self.head_pos_dict.items():
            t = np.asarray(d["tail"], dtype=np.int32)
            tails.append(t)
            rels.append(np.asarray(d["relation"], dtype=np.int32))
            heads.append(np.full_like(t, head))
            self.head_n_pos[head] = len(t)
        self.all_heads, self.relations, self.pos_tails = (np.concatenate(x) for x in (heads, rels, tails))
        self._g = _GroupSampler(self.num_entities, {h: d["tail"] for h, d in self.head_pos_dict.items()},
                                self.head_n_pos, num_neg)
        self._cols = [torch.from_numpy(np.ascontiguousarray(a)).to(self._g.dev)
                      for a in (self.all_heads, self.relations, self.pos_tails)]

    def __len__(self):
        return _n_batches(len(self.all_heads), self.batch_size, self.drop_last)

    def iter_device(self):
        neg = self._g.sample()
        if self.num_neg > 1:
            neg = neg.view(-1, self.num_neg)
        return _device_batches(self._cols + [neg], self.batch_size, self.shuffle, self.drop_last)

    def __iter__(self):
        for cols in self.iter_device():
            yield tuple(c.cpu().numpy() for c in cols)


class _VecIterator(object):
    def __init__(self, matrix, n, batch_size, shuffle, drop_last):
        self._m, self._n = matrix, n
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last

    def __len__(self):
        return _n_batches(self._n, self.batch_size, self.drop_last)

    def __iter__(self):
        order = np.random.permutation(self._n) if self.shuffle else np.arange(self._n)
        for start in range(0, self._n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            yield self._m[idx].toarray()


class UserVecIterator(_VecIterator):
    """dense rows of the user-item matrix per batch of users (reference: data_iterator.py:334-352); host-side"""

    def __init__(self, dataset: ImplicitFeedback, batch_size: int = 1024, shuffle: bool = True, drop_last: bool = False):
        super().__init__(dataset.to_csr_matrix(), dataset.num_users, batch_size, shuffle, drop_last)


class ItemVecIterator(_VecIterator):
    """dense rows of the item-user matrix per batch of items (reference: data_iterator.py:355-373); host-side"""

    def __init__(self, dataset: ImplicitFeedback, batch_size: int = 1024, shuffle: bool = True, drop_last: bool = False):
        super().__init__(dataset.to_csr_matrix().transpose().tocsr(), dataset.num_items, batch_size, shuffle, drop_last)
