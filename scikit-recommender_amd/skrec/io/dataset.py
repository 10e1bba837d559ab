"""Dataset boundary of the hot path (reference: skrec/io/dataset.py).

Same on-disk format and the same views the reference's models consume --
``<data_dir>/<name>.{train,valid,test}`` TSV (dataset.py:388-403), ``to_user_dict`` (:150-156),
``to_user_item_pairs`` (:113-116), ``to_csr_matrix`` / ``to_coo_matrix`` (:131-148) -- but every view
is derived from ONE CSR built with a stable sort (no pandas group-by loop per user), and
``to_csr_arrays()`` hands that CSR to the HIP kernels.  The reference's pickle side files
(``_data_cache/*.bin``) are neither written nor read.
"""
import os
import warnings
from collections import OrderedDict, defaultdict
from typing import Dict, List, Set

import numpy as np
import pandas as pd
import scipy.sparse as sp

__all__ = ["ImplicitFeedback", "KnowledgeGraph", "RSDataset", "UserGroup", "group_users_by_interactions"]

_USER, _ITEM, _RATING, _TIME = "user", "item", "rating", "time"
_DColumns = {"UI": [_USER, _ITEM], "UIR": [_USER, _ITEM, _RATING], "UIT": [_USER, _ITEM, _TIME],
             "UIRT": [_USER, _ITEM, _RATING, _TIME]}
_HEAD, _TAIL, _RELATION = "head", "tail", "relation"


def _sort_device(n):
    """the GPU for ingest-time sorts of large interaction tables (None: numpy; the results are identical)"""
    if n < (1 << 20):
        return None
    try:
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    except Exception:  # noqa: BLE001 -- ingest must not depend on the GPU
        return None


class ImplicitFeedback(object):
    """One split (train / valid / test) of user-item interactions."""

    def __init__(self, data: pd.DataFrame = None, num_users: int = None, num_items: int = None):
        assert data is None or isinstance(data, pd.DataFrame)
        self._views = {}
        if data is None or data.empty:
            self._data = pd.DataFrame()
            self.num_users = self.num_items = self.num_ratings = 0
        else:
            self._data = data
            self.num_users = num_users if num_users is not None else int(data[_USER].max()) + 1
            self.num_items = num_items if num_items is not None else int(data[_ITEM].max()) + 1
            self.num_ratings = len(data)

    def is_empty(self) -> bool:
        return self._data is None or self._data.empty

    def __len__(self):
        return len(self._data)

    def _cached(self, key, build):
        if key not in self._views:
            self._views[key] = build()
        return self._views[key]

    # ---- the one CSR everything else is derived from ---------------------------------------------
    def to_csr_arrays(self):
        """(rowptr int64 [U+1], items int32 [E] in FILE order per user, items_sorted int32 [E]
        ascending per user).  File order per user is what pandas' group-by yields in the reference
        (dataset.py:153-155) and what PairwiseIterator pairs negatives with (data_iterator.py:30-42)."""
        def build():
            if self.is_empty():
                z = np.zeros(0, np.int32)
                return np.zeros(self.num_users + 1, np.int64), z, z
            users = self._data[_USER].to_numpy(dtype=np.int64)
            items = self._data[_ITEM].to_numpy(dtype=np.int32)
            rowptr = np.zeros(self.num_users + 1, np.int64)
            np.cumsum(np.bincount(users, minlength=self.num_users), out=rowptr[1:])
            dev = _sort_device(len(users))
            if dev is not None:   # the two sorts of 5e7 pairs take ~20 s with numpy, well under a second on the GPU
                import torch
                u, it = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
                file_order = it[torch.argsort(u, stable=True)].cpu().numpy()
                m = int(it.max()) + 1 if len(items) else 1
                by_item = (torch.sort(u * m + it.long()).values % m).int().cpu().numpy()
                return rowptr, np.ascontiguousarray(file_order), np.ascontiguousarray(by_item)
            order = np.argsort(users, kind="stable")
            file_order = np.ascontiguousarray(items[order])
            by_item = np.lexsort((items, users))
            return rowptr, file_order, np.ascontiguousarray(items[by_item])
        return self._cached("csr", build)

    def to_user_item_pairs(self) -> np.ndarray:
        return self._cached("pairs", lambda: self._data[[_USER, _ITEM]].to_numpy(copy=True, dtype=np.int32)).copy()

    def to_set_of_users(self) -> Set[int]:
        return set(self._data[_USER].unique())

    def to_user_dict(self) -> Dict[int, np.ndarray]:
        """user -> int32 items in file order; users ascending, users without interactions absent."""
        rowptr, items, _ = self.to_csr_arrays()
        out = OrderedDict()
        for u in np.flatnonzero(np.diff(rowptr)):
            out[int(u)] = items[rowptr[u]:rowptr[u + 1]].copy()
        return out

    def to_user_dict_by_time(self) -> Dict[int, np.ndarray]:
        if _TIME not in self._data:
            raise ValueError("This dataset do not contain timestamp.")
        d = self._data.sort_values(by=[_USER, _TIME], kind="stable")
        users = d[_USER].to_numpy(dtype=np.int64)
        items = d[_ITEM].to_numpy(dtype=np.int32)
        out = OrderedDict()
        bounds = np.flatnonzero(np.diff(users)) + 1
        for chunk_u, chunk_i in zip(np.split(users, bounds), np.split(items, bounds)):
            if len(chunk_u):
                out[int(chunk_u[0])] = chunk_i
        return out

    def to_user_item_pairs_by_time(self) -> np.ndarray:
        if _TIME not in self._data:
            raise ValueError("This dataset do not contain timestamp.")
        d = self._data[[_USER, _ITEM, _TIME]].sort_values(by=[_USER, _TIME], kind="stable")
        return d[[_USER, _ITEM]].to_numpy(copy=True, dtype=np.int32)

    def to_item_dict(self) -> Dict[int, np.ndarray]:
        users = self._data[_USER].to_numpy(dtype=np.int32)
        items = self._data[_ITEM].to_numpy(dtype=np.int64)
        order = np.argsort(items, kind="stable")
        out = OrderedDict()
        bounds = np.flatnonzero(np.diff(items[order])) + 1
        for chunk_i, chunk_u in zip(np.split(items[order], bounds), np.split(users[order], bounds)):
            if len(chunk_i):
                out[int(chunk_i[0])] = chunk_u
        return out

    def to_csr_matrix(self) -> sp.csr_matrix:
        users, items = self._data[_USER].to_numpy(), self._data[_ITEM].to_numpy()
        return sp.csr_matrix((np.ones(len(users), dtype=np.float32), (users, items)),
                             shape=(self.num_users, self.num_items), copy=True)

    def to_csc_matrix(self) -> sp.csc_matrix:
        return self.to_csr_matrix().tocsc()

    def to_dok_matrix(self) -> sp.dok_matrix:
        return self.to_csr_matrix().todok()

    def to_coo_matrix(self) -> sp.coo_matrix:
        return self.to_csr_matrix().tocoo()

    def clear_cache(self):
        self._views.clear()


class KnowledgeGraph(object):
    """(head, relation, tail) triples; only what KGPairwiseIterator needs (reference: dataset.py:199-231).
    Knowledge-graph FILE loading (KGData / CFKGData) stays out of scope."""

    def __init__(self, data: pd.DataFrame = None, num_entities: int = None, num_relations: int = None):
        assert data is None or isinstance(data, pd.DataFrame)
        if data is None or data.empty:
            self._data = pd.DataFrame()
            self.num_entities = self.num_relations = self.num_triplets = 0
        else:
            self._data = data
            self.num_entities = num_entities if num_entities is not None else \
                int(max(data[_HEAD].max(), data[_TAIL].max())) + 1
            self.num_relations = num_relations if num_relations is not None else int(data[_RELATION].max()) + 1
            self.num_triplets = len(data)

    def is_empty(self) -> bool:
        return self._data is None or self._data.empty

    def to_triplets(self) -> np.ndarray:
        return self._data[[_HEAD, _RELATION, _TAIL]].to_numpy(copy=True, dtype=np.int32)

    def to_head_dict(self) -> Dict[int, Dict[str, np.ndarray]]:
        """head -> {"relation": int32[], "tail": int32[]}; heads ascending, triples of a head in file order"""
        heads = self._data[_HEAD].to_numpy(dtype=np.int64)
        order = np.argsort(heads, kind="stable")
        rel = self._data[_RELATION].to_numpy(dtype=np.int32)[order]
        tail = self._data[_TAIL].to_numpy(dtype=np.int32)[order]
        bounds = np.flatnonzero(np.diff(heads[order])) + 1
        out = OrderedDict()
        for h, r, t in zip(np.split(heads[order], bounds), np.split(rel, bounds), np.split(tail, bounds)):
            if len(h):
                out[int(h[0])] = {_RELATION: r, _TAIL: t}
        return out


def _read_table(path, sep, names, missing):
    if os.path.isfile(path):
        # pandas' multi-threaded pyarrow parser (same dtypes and values as the C parser on these all-numeric tables,
        # several times faster on 5e7 rows); anything it does not take -- multi-character separators, odd quoting,
        # an empty file, a missing pyarrow -- goes to the C parser, as the reference reads it (dataset.py:400-417)
        if isinstance(sep, str) and len(sep) == 1 and os.path.getsize(path) > (1 << 20):
            try:
                return pd.read_csv(path, sep=sep, header=None, names=names, engine="pyarrow")
            except Exception:  # noqa: BLE001
                pass
        return pd.read_csv(path, sep=sep, header=None, names=names)
    missing(f"'{path}' does not exist.")
    return pd.DataFrame()


class RSDataset(object):
    """``RSDataset(data_dir, sep, columns)``; splits are loaded lazily on first access
    (reference: dataset.py:582-600, CFData._load_cf_data :388-425)."""

    def __init__(self, data_dir, sep, columns):
        self._data_dir = data_dir
        self.sep = sep
        self.columns = columns
        self._log_print = print
        self._loaded = False

    def set_logger(self, logger):
        self._log_print = logger.info

    @property
    def data_dir(self) -> str:
        return self._data_dir

    @property
    def data_name(self) -> str:
        return os.path.split(self.data_dir)[-1]

    @property
    def file_prefix(self):
        return os.path.join(self.data_dir, self.data_name)

    def _load(self):
        if self._loaded:
            return
        if self.columns not in _DColumns:
            raise ValueError("'columns' must be one of '%s'." % ", ".join(_DColumns))
        names = _DColumns[self.columns]

        def must_exist(msg):
            raise FileNotFoundError(msg)
        train = _read_table(self.file_prefix + ".train", self.sep, names, must_exist)
        valid = _read_table(self.file_prefix + ".valid", self.sep, names, lambda m: None)
        test = _read_table(self.file_prefix + ".test", self.sep, names, must_exist)
        for label, frame in (("Training", train), ("Validation", valid), ("Test", test)):
            if not frame.empty and frame.isnull().values.any():
                warnings.warn(f"{label} data has None value, please check the file or the separator.")
        present = [f for f in (train, valid, test) if not f.empty]
        self._num_users = max(int(f[_USER].max()) for f in present) + 1
        self._num_items = max(int(f[_ITEM].max()) for f in present) + 1
        self._num_ratings = sum(len(f) for f in present)
        mk = lambda f: ImplicitFeedback(f, num_users=self._num_users, num_items=self._num_items)  # noqa: E731
        self._train, self._valid, self._test = mk(train), mk(valid), mk(test)
        self._loaded = True
        self._log_print(self.statistic_info)

    @property
    def cf_data(self):
        self._load()
        return self

    @property
    def train_data(self) -> ImplicitFeedback:
        self._load()
        return self._train

    @property
    def valid_data(self) -> ImplicitFeedback:
        self._load()
        return self._valid

    @property
    def test_data(self) -> ImplicitFeedback:
        self._load()
        return self._test

    @property
    def num_users(self) -> int:
        self._load()
        return self._num_users

    @property
    def num_items(self) -> int:
        self._load()
        return self._num_items

    @property
    def num_ratings(self) -> int:
        self._load()
        return self._num_ratings

    @property
    def statistic_info(self) -> str:
        if not self._loaded or 0 in {self._num_users, self._num_items, self._num_ratings}:
            return ""
        u, i, r = self._num_users, self._num_items, self._num_ratings
        lines = ["Dataset statistic information:", f"Name: {self.data_name}",
                 f"Name: {os.path.abspath(self.data_dir)}", f"The number of users: {u}",
                 f"The number of items: {i}", f"The number of ratings: {r}",
                 f"Average actions of users: {r / u:.2f}", f"Average actions of items: {r / i:.2f}",
                 f"The sparsity of the dataset: {(1 - r / (u * i)) * 100:.6f}%%", "",
                 f"The number of training: {len(self._train)}", f"The number of validation: {len(self._valid)}",
                 f"The number of testing: {len(self._test)}"]
        return "\n".join(lines)


class UserGroup(object):
    def __init__(self, users, num_interactions, activities, label):
        self.label = label
        self.num_users = len(users)
        self.num_interactions = num_interactions
        self.users = users
        self.activities = activities


def group_users_by_interactions(dataset: RSDataset, num_groups=4) -> List[UserGroup]:
    """Split users into ``num_groups`` activity bands of roughly equal interaction mass
    (reference: dataset.py:707-765).  Not on the hot path; used by ``evaluate_group``."""
    rowptr, _, _ = dataset.train_data.to_csr_arrays()
    lens = np.diff(rowptr)
    by_activity = defaultdict(list)
    for u in np.flatnonzero(lens):
        by_activity[int(lens[u])].append(int(u))
    acts = np.array(sorted(by_activity))
    if len(acts) == 0:
        return []
    n_users = np.array([len(by_activity[a]) for a in acts])
    mass = acts * n_users
    cuts, rest, start = [], mass, 0
    for g in range(num_groups - 1):
        if len(rest) <= 1:
            break
        target = rest.sum() / (num_groups - g)
        cum = np.cumsum(rest)
        k = max(int(np.searchsorted(cum, target)), 1)
        k = min(k, len(cum) - 1)
        split = k - 1 if target - cum[k - 1] < cum[k] - target else k
        split += 1
        start += split
        cuts.append(start)
        rest = rest[split:]
    cuts = [c for c in cuts if c < len(acts)]
    bounds = acts[cuts] if cuts else np.array([], dtype=acts.dtype)
    labels = []
    if len(bounds):
        labels.append(f"< {bounds[0]}")
        labels += [f"[{lo}, {hi})" for lo, hi in zip(bounds[:-1], bounds[1:])]
        labels.append(f"≥ {bounds[-1]}")
    else:
        labels.append("all")
    groups = []
    for label, act_chunk, mass_chunk in zip(labels, np.split(acts, cuts), np.split(mass, cuts)):
        users = [u for a in act_chunk for u in by_activity[int(a)]]
        groups.append(UserGroup(np.array(users), int(mass_chunk.sum()), act_chunk, label))
    return groups
