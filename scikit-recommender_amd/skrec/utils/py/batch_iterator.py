"""``BatchIterator`` -- the reference's DataLoader-like batcher (utils/py/batch_iterator.py), kept for
API parity and for small host-side lists (e.g. test users).  Batches are produced by slicing numpy
arrays instead of per-element Python indexing; the order contract is the reference's: one
``np.random.permutation(n)`` from numpy's global generator per ``__iter__`` when ``shuffle`` is set
(batch_iterator.py:61-63), consecutive slices of ``batch_size``, a short last batch unless
``drop_last`` (batch_iterator.py:98-106)."""
import numpy as np

__all__ = ["BatchIterator"]


class Sampler(object):
    """Base class of index samplers (batch_iterator.py:10-24)."""

    def __iter__(self):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError


class SequentialSampler(Sampler):
    def __init__(self, data_source):
        self.data_source = data_source

    def __iter__(self):
        return iter(range(len(self.data_source)))

    def __len__(self):
        return len(self.data_source)


class RandomSampler(Sampler):
    def __init__(self, data_source):
        self.data_source = data_source

    def __iter__(self):
        return iter(np.random.permutation(len(self.data_source)).tolist())

    def __len__(self):
        return len(self.data_source)


class BatchIterator(object):
    def __init__(self, *data, batch_size=1, shuffle=False, drop_last=False):
        if not isinstance(batch_size, int) or isinstance(batch_size, bool) or batch_size <= 0:
            raise ValueError(f"batch_size should be a positive integeral value, but got batch_size={batch_size}")
        if not isinstance(drop_last, bool):
            raise ValueError(f"drop_last should be a boolean value, but got drop_last={drop_last}")
        n = len(data[0])
        for d in data:
            if len(d) != n:
                raise ValueError("The length of the given data are not equal!")
        self._columns = list(data)
        self._n = n
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.drop_last = drop_last

    def __len__(self):
        if self.drop_last:
            return self._n // self.batch_size
        return (self._n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = np.random.permutation(self._n) if self.shuffle else np.arange(self._n)
        single = len(self._columns) == 1
        for start in range(0, self._n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            batch = [[col[i] for i in idx] if not isinstance(col, np.ndarray) else list(col[idx])
                     for col in self._columns]
            yield batch[0] if single else batch
