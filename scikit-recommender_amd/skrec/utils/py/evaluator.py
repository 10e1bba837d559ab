"""``RankingEvaluator`` on the GPU (reference: skrec/utils/py/evaluator.py).

Same constructor, same ``evaluate(model, test_users=None) -> MetricReport``, same metric names and
column order.  Two device paths, both through libskrec_hip.so:

* **fused** -- taken when the model exposes ``predict_factors() -> (user_table, item_table, bias|None)``
  (fp32 device tensors, 64 columns): ``skr_eval_fused_topk`` scores every test user against the whole
  catalogue on FP32 MFMA, masks train items and keeps the top-K without ever forming the [B, I]
  score matrix; ``skr_rank_metrics`` turns the lists into metric rows.
* **generic** -- any object with ``predict(users) -> ndarray [B, I]`` (the reference's contract,
  evaluator.py:180-193): the scores are uploaded, train items masked (``skr_mask_train``) and
  ranked by ``skr_eval_scores``, the drop-in for ``cpp_evaluate_matrix``.

Aggregation: the reference takes a float32 ``np.mean`` over the per-user rows (evaluator.py:207-208).
Up to ``_HOST_MEAN_MAX`` users we do exactly that on the (bit-identical) rows; beyond it the rows are
summed in fp64 on the device, which is what the 1e-5 parity at 1 M users is defined against.
"""
import itertools
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from ... import _hip

__all__ = ["MetricReport", "RankingEvaluator", "EarlyStopping"]

_HOST_MEAN_MAX = 1 << 18
_FUSED_CHUNK = 1 << 18     # users per fused launch (512 MB of candidate scratch): 4 workgroups per CU


class MetricReport(object):
    def __init__(self, metrics: Sequence[str], values: Sequence[float]):
        assert len(metrics) == len(values), f"The lengths of metrics and values " \
                                            f"are not equal ({len(metrics)}!={len(values)})."
        self._results = OrderedDict(zip(metrics, values))

    def metrics(self):
        return self._results.keys()

    def values(self):
        return self._results.values()

    def items(self):
        return self._results.items()

    @property
    def results(self) -> Dict[str, float]:
        return self._results

    @property
    def metrics_str(self) -> str:
        return "\t".join(f"{m}".ljust(12) for m in self.metrics())

    @property
    def values_str(self) -> str:
        return "\t".join(f"{v:.8f}".ljust(12) for v in self.values())

    def __getitem__(self, item):
        if item not in self._results:
            raise KeyError(item)
        return self._results[item]

    def __str__(self):
        return str(self._results)


_metric2id = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}
_id2metric = {v: k for k, v in _metric2id.items()}


def _dict_to_device_csr(d, n_rows, dev):
    """dict user -> items  ==>  CSR (rowptr int64 [n_rows+1], items int32 sorted & unique per row) as device
    tensors, plus the longest row.  Sorted and de-duplicated on the GPU: a 48 M-pair np.lexsort takes ~8 s on
    the host, this takes ~1 s for a million users (most of it the walk over the dict)."""
    import torch
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    if len(d) == 0:
        return rowptr, torch.zeros(1, dtype=torch.int32, device=dev), 0
    keys = np.fromiter((int(k) for k in d.keys()), dtype=np.int64, count=len(d))
    vals = [np.asarray(v).reshape(-1) for v in d.values()]
    lens = np.fromiter((len(v) for v in vals), dtype=np.int64, count=len(vals))
    if lens.sum() == 0:
        return rowptr, torch.zeros(1, dtype=torch.int32, device=dev), 0
    items = torch.from_numpy(np.concatenate(vals).astype(np.int64, copy=False)).to(dev)
    users = torch.repeat_interleave(torch.from_numpy(keys).to(dev), torch.from_numpy(lens).to(dev))
    m = int(items.max()) + 1
    key = torch.unique(users * m + items)             # sorted, set semantics per row
    users, items = torch.div(key, m, rounding_mode="floor"), (key % m).int().contiguous()
    counts = torch.bincount(users, minlength=n_rows)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, items, int(counts.max())


class RankingEvaluator(object):
    """Evaluator for the item-ranking task; metrics ``Precision, Recall, MAP, NDCG, MRR``.
    In leave-one-out evaluation ``Recall`` equals HitRatio (evaluator.py:75-76)."""

    def __init__(self, user_train_dict: Optional[Dict[int, np.ndarray]], user_test_dict: Dict[int, np.ndarray],
                 metric: Union[None, str, Tuple[str], List[str]] = None, top_k: Union[int, List[int], Tuple[int]] = 50,
                 batch_size: int = 256, num_thread: int = 8):
        if metric is None:
            metric = ["Precision", "Recall", "MAP", "NDCG", "MRR"]
        elif isinstance(metric, str):
            metric = [metric]
        elif isinstance(metric, (tuple, list)):
            metric = list(metric)
        else:
            raise TypeError("The type of 'metric' (%s) is invalid!" % metric.__class__.__name__)
        for m in metric:
            assert m in _metric2id, f"'{metric}' is not in ('Precision', 'Recall', 'MAP', 'NDCG', 'MRR')."
        self.user_pos_train = dict()
        self.user_pos_test = dict()
        self._dev = None  # device CSRs, built lazily
        self.set_train_data(user_train_dict)
        self.set_test_data(user_test_dict)
        self.metrics_num = len(metric)
        self.metrics = [_metric2id[m] for m in metric]
        self.num_thread = num_thread   # API parity; the GPU path has no thread pool
        self.batch_size = batch_size
        if isinstance(top_k, (int, np.integer)):
            self.max_top = int(top_k)
            self.top_show = np.arange(top_k) + 1
        else:
            self.max_top = int(max(top_k))
            self.top_show = np.sort(top_k)
        # the fused GEMM + top-K kernel ranks up to SKR_MAX_TOPK (128: the reference's default (10, ..., 100) fits); deeper lists
        # go through the score-matrix path (skr_score_matrix / predict() -> skr_mask_train -> skr_eval_scores)
        if self.max_top > _hip.SKR_MAX_TOPK_SCORES:
            raise NotImplementedError(f"top_k up to {_hip.SKR_MAX_TOPK_SCORES} is supported by the HIP kernels "
                                      f"(got {self.max_top})")

    def set_train_data(self, user_train_dict: Optional[Dict[int, np.ndarray]] = None):
        self.user_pos_train = user_train_dict if user_train_dict is not None else dict()
        self._dev = None

    def set_test_data(self, user_test_dict: Dict[int, np.ndarray]):
        assert len(user_test_dict) > 0, "'user_test_dict' can be empty."
        self.user_pos_test = user_test_dict
        self._dev = None

    @property
    def metrics_list(self) -> List[str]:
        return [f"{_id2metric[mid]}@{str(k)}" for mid in self.metrics for k in self.top_show]

    @property
    def metrics_str(self) -> str:
        return "\t".join(f"{m}".ljust(12) for m in self.metrics_list)

    # ---- device state ----------------------------------------------------------------------------
    def _device_state(self):
        if self._dev is None:
            import torch
            dev = _hip.require_gpu()
            keys = itertools.chain(self.user_pos_train.keys(), self.user_pos_test.keys())
            n_rows = max(int(u) for u in keys) + 1
            tr_ptr, tr_items, max_train = _dict_to_device_csr(self.user_pos_train, n_rows, dev)
            te_ptr, te_items, _ = _dict_to_device_csr(self.user_pos_test, n_rows, dev)
            self._dev = dict(dev=dev, n_rows=n_rows, max_train=max_train, tr_ptr=tr_ptr, tr_items=tr_items,
                             te_ptr=te_ptr, te_items=te_items)
        return self._dev

    # ---- evaluation ------------------------------------------------------------------------------
    def evaluate(self, model, test_users: Optional[Iterable[int]] = None) -> MetricReport:
        assert hasattr(model, "predict"), "the model must have attribute 'predict'."
        if test_users is not None:
            test_users = [u for u in test_users if u in self.user_pos_test]
        else:   # all test users, in the dict's order; the int32 array is built once (1 M keys cost ~0.1 s per call)
            if getattr(self, "_all_test_users", None) is None:
                self._all_test_users = np.fromiter(self.user_pos_test.keys(), dtype=np.int32, count=len(self.user_pos_test))
            test_users = self._all_test_users
        assert isinstance(test_users, Iterable), "'test_user' must be iterable."
        rows, sums, n = self.per_user_rows(model, test_users)
        if rows is not None:
            final = np.mean(rows, axis=0)                      # float32, as evaluator.py:208
        else:
            final = (sums / max(n, 1)).astype(np.float32)
        final = final.reshape(self.metrics_num, self.max_top)[:, self.top_show - 1].reshape(-1)
        return MetricReport(self.metrics_list, final)

    _TIE_CHUNK = 2048

    def _rerank_tied_rows(self, st, ut, it, bias, d_users, ids, top_sc):
        """Users whose fused top-K holds EQUAL scores (structural ties: cold users with an all-zero row,
        duplicated items) are ranked again from their dense score row by skr_eval_scores, which reproduces the
        order the reference's partial_sort_copy gives equal scores (evaluate.h:39-45); the fused kernel itself
        orders them by item id.  Tie-free rows -- all of them, normally -- cost one comparison pass."""
        import torch
        tied = _tied_rows(top_sc)
        if tied.numel() == 0:
            return
        K = ids.shape[1]
        n_items = int(it.shape[0])
        for s in range(0, tied.numel(), self._TIE_CHUNK):
            sel = tied[s:s + self._TIE_CHUNK]
            du = d_users[sel].contiguous()
            sc = torch.empty((du.numel(), n_items), dtype=torch.float32, device=ids.device)
            L = _hip.lib()
            _hip.check(L.skr_score_matrix(_hip.ptr(ut), _hip.ptr(du), du.numel(), _hip.ptr(it), _hip.ptr(bias), n_items, 64,
                                          _hip.ptr(sc), n_items, _hip.stream()))
            _hip.check(L.skr_mask_train(_hip.ptr(sc), du.numel(), n_items, n_items, _hip.ptr(du), _hip.ptr(st["tr_ptr"]),
                                        _hip.ptr(st["tr_items"]), _hip.stream()))
            fixed = torch.empty((du.numel(), K), dtype=torch.int32, device=ids.device)
            _hip.check(L.skr_eval_scores(_hip.ptr(sc), du.numel(), n_items, n_items, None, None, None, 0, K, None,
                                         _hip.ptr(fixed), None, _hip.stream()))
            ids[sel] = fixed

    def per_user_rows(self, model, test_users):
        """-> (rows float32 [n, n_metric*max_top] on the host or None, fp64 column sums, n)."""
        import torch
        st = self._device_state()
        dev, K, nm = st["dev"], self.max_top, self.metrics_num
        users = test_users if isinstance(test_users, np.ndarray) and test_users.dtype == np.int32 \
            else np.asarray(list(test_users), dtype=np.int32)
        n = len(users)
        d_sums = torch.zeros(nm * K, dtype=torch.float64, device=dev)
        keep_rows = n <= _HOST_MEAN_MAX
        host_rows = []
        margs = _hip.metric_array(self.metrics)
        factors = model.predict_factors() if hasattr(model, "predict_factors") else None
        if factors is not None:
            ut, it, bias = factors
            n_items = int(it.shape[0])
            fused_ok = (it.shape[1] == 64 and ut.shape[1] == 64 and n_items - st["max_train"] >= K and K <= _hip.SKR_MAX_TOPK)
        if factors is not None and fused_ok:
            # pass 1: top-K lists of every user, chunk after chunk without touching the host.  One more entry than needed
            # is asked for where possible (Kq = K + 1): with the (K+1)-th score in hand, a tie that exists ONLY between the
            # last kept and the first dropped item is seen too, and such a user is re-ranked like any other tied one
            d_users = torch.from_numpy(users).to(dev)
            Kq = K + 1 if (K + 1 <= _hip.SKR_MAX_TOPK and n_items - st["max_train"] >= K + 1) else K
            work = torch.empty(int(_hip.lib().skr_eval_fused_workspace(min(n, _FUSED_CHUNK), Kq)), dtype=torch.uint8,
                               device=dev)
            ids_q = torch.empty((n, Kq), dtype=torch.int32, device=dev)
            top_sc = torch.empty((n, Kq), dtype=torch.float32, device=dev)
            for s in range(0, n, _FUSED_CHUNK):
                b = min(_FUSED_CHUNK, n - s)
                _hip.check(_hip.lib().skr_eval_fused_topk(
                    _hip.ptr(ut), _hip.ptr(d_users[s:s + b]), b, _hip.ptr(it), _hip.ptr(bias), n_items, 64,
                    _hip.ptr(st["tr_ptr"]), _hip.ptr(st["tr_items"]), Kq, _hip.ptr(ids_q[s:s + b]), _hip.ptr(top_sc[s:s + b]),
                    _hip.ptr(work), work.numel(), _hip.stream()))
            ids = ids_q if Kq == K else ids_q[:, :K].contiguous()
            # one look at the scores (the only host synchronisation): users with equal scores are re-ranked
            self._rerank_tied_rows(st, ut, it, bias, d_users, ids, top_sc)
            del top_sc, work, ids_q
            # pass 2: the metrics
            rows = None
            for s in range(0, n, _FUSED_CHUNK):
                b = min(_FUSED_CHUNK, n - s)
                if rows is None or keep_rows:
                    rows = torch.empty((min(_FUSED_CHUNK, n), nm * K), dtype=torch.float32, device=dev)
                _hip.check(_hip.lib().skr_rank_metrics(
                    _hip.ptr(ids[s:s + b]), b, K, _hip.ptr(d_users[s:s + b]), _hip.ptr(st["te_ptr"]), _hip.ptr(st["te_items"]),
                    margs, nm, _hip.ptr(rows), _hip.ptr(d_sums), _hip.stream()))
                if keep_rows:
                    host_rows.append(rows[:b].cpu().numpy())
        else:
            # generic contract of the reference: predict() returns a dense [B, I] ndarray
            bs = max(int(self.batch_size), 1)
            for s in range(0, n, bs):
                bu = users[s:s + bs]
                score = model.predict(list(bu))
                assert isinstance(score, np.ndarray), "'ranking_score' must be an np.ndarray"
                d_sc = torch.from_numpy(np.ascontiguousarray(score, np.float32)).to(dev)
                du = torch.from_numpy(bu).to(dev)
                b, n_items = d_sc.shape
                _hip.check(_hip.lib().skr_mask_train(_hip.ptr(d_sc), b, n_items, n_items, _hip.ptr(du),
                                                     _hip.ptr(st["tr_ptr"]), _hip.ptr(st["tr_items"]), _hip.stream()))
                # gather the truth rows of this batch into a compact CSR so that row b <-> user bu[b]
                rows = torch.empty((b, nm * K), dtype=torch.float32, device=dev)
                te_ptr, te_items = _batch_truth(st, du)
                _hip.check(_hip.lib().skr_eval_scores(_hip.ptr(d_sc), b, n_items, n_items, _hip.ptr(te_ptr),
                                                      _hip.ptr(te_items), margs, nm, K, _hip.ptr(rows), None,
                                                      _hip.ptr(d_sums), _hip.stream()))
                if keep_rows:
                    host_rows.append(rows.cpu().numpy())
        sums = d_sums.cpu().numpy()
        rows = np.concatenate(host_rows, axis=0) if keep_rows and host_rows else None
        return rows, sums, n


def _tied_rows(top_sc):
    """indices of the rows whose top-K scores contain equal neighbours (device tensor, host-synchronising)"""
    import torch
    if top_sc.shape[1] < 2:
        return torch.zeros(0, dtype=torch.int64, device=top_sc.device)
    return torch.nonzero((top_sc[:, 1:] == top_sc[:, :-1]).any(dim=1)).reshape(-1)


def _batch_truth(st, d_users):
    """compact test CSR for the users of one batch (device tensors)"""
    import torch
    ptr, items = st["te_ptr"], st["te_items"]
    u = d_users.long()
    beg, end = ptr[u], ptr[u + 1]
    lens = end - beg
    out_ptr = torch.zeros(len(u) + 1, dtype=torch.int64, device=ptr.device)
    out_ptr[1:] = torch.cumsum(lens, 0)
    total = int(out_ptr[-1])
    if total == 0:
        return out_ptr, torch.zeros(1, dtype=torch.int32, device=ptr.device)
    row_of = torch.repeat_interleave(torch.arange(len(u), device=ptr.device), lens)
    offs = torch.arange(total, device=ptr.device) - out_ptr[row_of]
    return out_ptr, items[beg[row_of] + offs].contiguous()


class EarlyStopping(object):
    """Stop when the monitored metric has not improved for ``patience`` evaluations
    (patience <= 0: never) -- reference: evaluator.py:217-246."""

    def __init__(self, metric: str = "NDCG@10", patience: int = 100):
        self._metric = metric
        self._patience = patience
        self._best_score = None
        self._counter = 0

    def __call__(self, val_result: MetricReport):
        if self._best_score is None:
            self._best_score = val_result
        elif val_result[self.key_metric] <= self._best_score[self.key_metric]:
            self._counter += 1
            if self._counter >= self._patience > 0:
                return True
        else:
            self._best_score = val_result
            self._counter = 0
        return False

    @property
    def key_metric(self) -> str:
        return self._metric

    @property
    def best_result(self) -> MetricReport:
        if self._best_score is not None:
            return self._best_score
        return MetricReport(["None"], [0])
