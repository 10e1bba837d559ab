"""LightGCN on MI355X (reference: skrec/recommender/LightGCN.py).

Paper: LightGCN: Simplifying and Powering Graph Convolution Network for Recommendation (He et al.).
Reference semantics are kept: the full-graph K-layer propagation runs forward AND backward on every
mini-batch (LightGCN.py:89-100, :180-199), the BPR loss is a mean, the L2 term uses the ego rows and
is divided by the CONFIGURED batch size (:191-196), Adam is dense.  Each propagation is
``skr_csr_spmm`` (CSR, one wavefront per 512 non-zeros, 64 lanes = 64 dims) with the layer mean
fused into its epilogue; the backward pass reuses the same kernel because the normalised adjacency
of every ``adj_type`` used here is applied as A^T = A for 'pre'/'plain' and explicitly transposed
otherwise.
"""
import os
from typing import Dict

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from .. import _hip
from ..io import PairwiseIterator
from ..run_config import RunConfig
from ..utils.common import make_sure_dirs, normalize_adj_matrix
from ..utils.py import EarlyStopping, ModelConfig
from ..utils.torch import get_initializer
from .base import AbstractRecommender, DenseAdam, on_compute_stream

__all__ = ["LightGCN", "LightGCNConfig", "DeviceCSR"]


class LightGCNConfig(ModelConfig):
    def __init__(self, lr=1e-3, reg=1e-3, embed_size=64, n_layers=3, adj_type="pre", batch_size=1024, epochs=1000,
                 early_stop=100, **kwargs):
        super().__init__()
        self.lr: float = lr
        self.reg: float = reg
        self.embed_size: int = embed_size
        self.n_layers: int = n_layers
        self.adj_type: str = adj_type  # plain, norm, gcmc, pre
        self.batch_size: int = batch_size
        self.epochs: int = epochs
        self.early_stop: int = early_stop

    def _validate(self):
        assert isinstance(self.lr, float) and self.lr > 0
        assert isinstance(self.reg, float) and self.reg >= 0
        assert isinstance(self.embed_size, int) and self.embed_size > 0
        assert isinstance(self.n_layers, int) and self.n_layers > 0
        assert self.adj_type in {"plain", "norm", "gcmc", "pre"}
        assert isinstance(self.batch_size, int) and self.batch_size > 0
        assert isinstance(self.epochs, int) and self.epochs >= 0
        assert isinstance(self.early_stop, int)


class DeviceCSR(object):
    """A scipy matrix as (rowptr int64, col int32, val fp32) in HBM; duplicates summed, columns
    ascending per row -- the same canonical form as the reference's coalesced COO tensor
    (utils/torch.py:32-35)."""

    def __init__(self, mat, device):
        csr = sp.csr_matrix(mat).astype(np.float32)
        csr.sum_duplicates()
        csr.sort_indices()
        self.shape = csr.shape
        self.nnz = int(csr.nnz)
        self.rowptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(device)
        self.col = torch.from_numpy(csr.indices.astype(np.int32)).to(device)
        self.val = torch.from_numpy(csr.data.astype(np.float32)).to(device)
        if self.nnz == 0:
            self.col = torch.zeros(1, dtype=torch.int32, device=device)
            self.val = torch.zeros(1, dtype=torch.float32, device=device)

    @classmethod
    def from_device_coo(cls, rows, cols, vals, n_rows):
        """build from int64 row / col and fp32 value tensors already in HBM (entries must be unique)"""
        self = cls.__new__(cls)
        key = rows * n_rows + cols
        order = torch.argsort(key)
        rows, cols, vals = rows[order], cols[order], vals[order]
        self.shape = (n_rows, n_rows)
        self.nnz = int(rows.numel())
        self.rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
        self.rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_rows), 0)
        self.col = cols.int().contiguous()
        self.val = vals.float().contiguous()
        if self.nnz == 0:
            self.col = torch.zeros(1, dtype=torch.int32, device=rows.device)
            self.val = torch.zeros(1, dtype=torch.float32, device=rows.device)
        return self

    # matrices with at least this many non-zeros are multiplied through a plan (skr_spmm_plan_*: 16-byte gathers, long rows
    # column-blocked); smaller ones through the plan-free kernel.  SKR_SPMM_PLAN=0 / 1 forces one or the other.
    PLAN_MIN_NNZ = 1 << 16

    def _plan_handle(self, long_rows_from=0):
        """the matrix's plan, built at the first product (one-time analysis on the device)"""
        h = getattr(self, "_plan", None)
        if h is None:
            import ctypes
            h = ctypes.c_void_p()
            if not long_rows_from:      # tuning switch: rows of at least this many entries are cut into column-blocked tasks (0 = 512)
                long_rows_from = int(os.environ.get("SKR_SPMM_LONG_FROM", "0"))
            _hip.check(_hip.lib().skr_spmm_plan_create(self.shape[0], self.shape[1], _hip.ptr(self.rowptr), _hip.ptr(self.col),
                                                       _hip.ptr(self.val), self.nnz, int(long_rows_from), ctypes.byref(h),
                                                       _hip.stream()))
            self._plan = h
        return h

    def plan_info(self):
        import ctypes
        info = (ctypes.c_int64 * 4)()
        _hip.check(_hip.lib().skr_spmm_plan_info(self._plan_handle(), info))
        return dict(long_rows=info[0] & 0xffffffff, hot_rows=info[0] >> 32, tasks=info[1], column_blocks=info[2],
                    long_rows_from=info[3] & 0xffffffff, column_windows=info[3] >> 32)

    def __del__(self):
        h = getattr(self, "_plan", None)
        if h is not None:
            try:
                _hip.lib().skr_spmm_plan_destroy(h)
            except Exception:   # noqa: BLE001 -- interpreter shutdown
                pass
            self._plan = None

    def scatter_marked_rows(self, row_mask, X, Y):
        """Y[c] += A[r, c] * X[r] over the entries of the rows marked in ``row_mask`` (uint8 [n_rows]): the product with the
        TRANSPOSE where X is zero outside the marked rows (skr_csr_scatter_marked_rows); Y must be initialised"""
        _hip.check(_hip.lib().skr_csr_scatter_marked_rows(self.shape[0], _hip.ptr(self.rowptr), _hip.ptr(self.col), _hip.ptr(self.val),
                                                          _hip.ptr(row_mask), _hip.ptr(X), 64, _hip.ptr(Y), _hip.stream()))
        return Y

    def uses_plan(self):
        """whether products of this matrix go through the plan (which honours the row / column masks)"""
        mode = os.environ.get("SKR_SPMM_PLAN", "auto")
        return mode == "1" or (mode != "0" and self.nnz >= self.PLAN_MIN_NNZ)

    def spmm(self, X, Y, addend=None, accum=None, accum_scale=1.0, row_mask=None, col_mask=None, accum_base=None, accum_init=False,
             refine_fwd=None, refine_bwd=None, accum_mask=None, addend_mask=None):
        """Y = A @ X (+ addend); accum += accum_scale * Y.
        ``row_mask`` (uint8 [n_rows]): only rows with a non-zero byte are needed (the others may be left untouched);
        ``col_mask`` (uint8 [n_cols]): rows of X with a zero byte ARE zero, their entries may be skipped.  Both are
        hints: the plan-free kernel computes the whole product.
        Row-local passes that ride in the product's row epilogue (skr_spmm_epilogue, include/skrec_hip.h):
        ``accum_base``: accum = accum_scale * accum_base + accum_scale * Y (accum is not read: the layer mean with its E0 term);
        ``accum_init``: accum = accum_scale * Y, resp. accum = Z (accum is not read);
        ``refine_fwd`` = (E, w, Z): LayerGCN's refinement of the finished rows, w = cos(Y, E), Z = w * Y, accum += Z
        (Y keeps the raw rows; may be None);
        ``refine_bwd`` = (E, w, rawY, dE): the finished rows are dZ of a refinement; Y receives dY, dE its E0 part.
        ``accum_mask`` (uint8 [n_rows]): accum is only needed on the rows with a non-zero byte; ``addend_mask`` (uint8
        [n_rows]): addend IS zero on the rows with a zero byte (hints again: the plan-free path reads / updates every row)."""
        L, st = _hip.lib(), _hip.stream()
        n = self.shape[0]
        W = int(X.shape[1])
        if W != 64:
            return self._spmm_wide(X, Y, W, addend, accum, accum_scale, row_mask, col_mask, accum_base, accum_init, refine_fwd,
                                   refine_bwd, accum_mask, addend_mask)
        if self.uses_plan():
            if Y is not None and accum_base is None and not accum_init and refine_fwd is None and refine_bwd is None \
                    and accum_mask is None and addend_mask is None:
                _hip.check(L.skr_spmm_plan_run_masked(self._plan_handle(), _hip.ptr(X), 64, _hip.ptr(addend), _hip.ptr(Y),
                                                      _hip.ptr(accum), float(accum_scale), _hip.ptr(row_mask),
                                                      _hip.ptr(col_mask), st))
                return Y
            import ctypes
            ep = _hip.SpmmEpilogue()
            ep.mode, ep.accum_init = _hip.EPI_PLAIN, int(bool(accum_init))
            ep.addend, ep.Y, ep.accum, ep.accum_base = _hip.ptr(addend), _hip.ptr(Y), _hip.ptr(accum), _hip.ptr(accum_base)
            ep.accum_scale = float(accum_scale)
            ep.accum_mask, ep.addend_mask = _hip.ptr(accum_mask), _hip.ptr(addend_mask)
            if refine_fwd is not None:
                ep.mode = _hip.EPI_REFINE_FWD
                ep.E, ep.w, ep.Z = (_hip.ptr(t) for t in refine_fwd)
            elif refine_bwd is not None:
                ep.mode = _hip.EPI_REFINE_BWD
                ep.E, ep.w, ep.rawY, ep.dE = (_hip.ptr(t) for t in refine_bwd)
            _hip.check(L.skr_spmm_plan_run_ex(self._plan_handle(), _hip.ptr(X), 64, ctypes.byref(ep), _hip.ptr(row_mask),
                                              _hip.ptr(col_mask), st))
            return Y
        # small graphs: the plan-free kernel computes whole products; the row-local passes are launches of their own
        raw = Y
        if Y is None or refine_bwd is not None:
            if getattr(self, "_tmp", None) is None:
                self._tmp = torch.empty((n, 64), dtype=torch.float32, device=X.device)
            raw = self._tmp
        plain = refine_fwd is None and refine_bwd is None
        if plain and accum is not None and accum_base is not None:
            _hip.check(L.skr_scale_copy(float(accum_scale), _hip.ptr(accum_base), _hip.ptr(accum), n * 64, st))
        elif accum is not None and accum_init:
            accum.zero_()
        _hip.check(L.skr_csr_spmm(n, _hip.ptr(self.rowptr), _hip.ptr(self.col), _hip.ptr(self.val), _hip.ptr(X), 64, self.nnz,
                                  _hip.ptr(addend), _hip.ptr(raw), _hip.ptr(accum) if plain else None, float(accum_scale), st))
        if refine_fwd is not None:
            E, w, Z = refine_fwd
            _hip.check(L.skr_layer_refine_fwd(_hip.ptr(raw), _hip.ptr(E), n, 64, _hip.ptr(Z), _hip.ptr(w), _hip.ptr(accum), st))
        elif refine_bwd is not None:
            E, w, rawY, dE = refine_bwd
            _hip.check(L.skr_layer_refine_bwd(_hip.ptr(rawY), _hip.ptr(E), _hip.ptr(w), _hip.ptr(raw), n, 64, _hip.ptr(Y), _hip.ptr(dE),
                                              st))
        return Y


    def _spmm_wide(self, X, Y, W, addend, accum, accum_scale, row_mask, col_mask, accum_base, accum_init, refine_fwd, refine_bwd,
                   accum_mask, addend_mask):
        """Embedding widths beyond 64 (tables [n, W], W = 64 C; narrower widths are zero-padded by the models): the product is
        separable in the columns, so it runs slice by slice through the 64-column kernels with the tables' row stride
        (skr_spmm_epilogue.ld / skr_csr_spmm_strided); LayerGCN's refinements reduce over whole rows and are launches of
        their own over the full width."""
        import ctypes
        L, st = _hip.lib(), _hip.stream()
        n, C = self.shape[0], W // 64
        assert W % 64 == 0 and 2 <= C <= 4, "pad the embedding width to a multiple of 64 (at most 256)"
        plain = refine_fwd is None and refine_bwd is None
        raw = Y
        if Y is None or refine_bwd is not None:
            if getattr(self, "_tmpw", None) is None or self._tmpw.shape[1] != W:
                self._tmpw = torch.empty((n, W), dtype=torch.float32, device=X.device)
            raw = self._tmpw
        if plain and accum is not None and accum_base is not None:
            _hip.check(L.skr_scale_copy(float(accum_scale), _hip.ptr(accum_base), _hip.ptr(accum), n * W, st))
        elif accum is not None and accum_init:
            accum.zero_()
        sl = lambda t, c: None if t is None else t.data_ptr() + 256 * c      # noqa: E731  the c-th 64-column slice
        for t in (X, raw, addend, accum):
            assert t is None or (t.is_contiguous() and t.shape[1] == W)
        for c in range(C):
            if self.uses_plan():
                ep = _hip.SpmmEpilogue()
                ep.mode, ep.ld = _hip.EPI_PLAIN, W
                ep.addend, ep.Y, ep.accum = sl(addend, c), sl(raw, c), sl(accum, c) if plain else None
                ep.accum_scale = float(accum_scale)
                ep.accum_mask, ep.addend_mask = _hip.ptr(accum_mask) if plain else None, _hip.ptr(addend_mask)
                _hip.check(L.skr_spmm_plan_run_ex(self._plan_handle(), sl(X, c), 64, ctypes.byref(ep), _hip.ptr(row_mask),
                                                  _hip.ptr(col_mask), st))
            else:
                _hip.check(L.skr_csr_spmm_strided(n, _hip.ptr(self.rowptr), _hip.ptr(self.col), _hip.ptr(self.val), sl(X, c), 64, W,
                                                  self.nnz, sl(addend, c), sl(raw, c), sl(accum, c) if plain else None,
                                                  float(accum_scale), st))
        if refine_fwd is not None:
            E, w, Z = refine_fwd
            _hip.check(L.skr_layer_refine_fwd(_hip.ptr(raw), _hip.ptr(E), n, W, _hip.ptr(Z), _hip.ptr(w), _hip.ptr(accum), st))
        elif refine_bwd is not None:
            E, w, rawY, dE = refine_bwd
            _hip.check(L.skr_layer_refine_bwd(_hip.ptr(rawY), _hip.ptr(E), _hip.ptr(w), _hip.ptr(raw), n, W, _hip.ptr(Y), _hip.ptr(dE),
                                              st))
        return Y


def padded_width(d):
    """the kernels' row widths are multiples of 64 floats (one 256-byte access per lane group): an embedding of another width
    -- n_dim / embed_size / embed_dim are free integers in the reference -- lives in zero-padded rows.  Padded columns have
    zero gradients, stay zero under Adam and add nothing to a dot product, a norm or a propagation."""
    dp = 64 * ((int(d) + 63) // 64)
    if dp > 256:
        raise NotImplementedError("embedding widths up to 256 are supported by the MI355X kernels")
    return dp


def pad_columns(t, dp):
    return t if t.shape[1] == dp else torch.nn.functional.pad(t, (0, dp - t.shape[1]))


def build_adjacency(users_np, items_np, num_users, num_items, adj_type):
    """LightGCN._create_adj_mat (LightGCN.py:142-169)"""
    n_nodes = num_users + num_items
    ones = np.ones_like(users_np, dtype=np.float32)
    upper = sp.csr_matrix((ones, (users_np, items_np + num_users)), shape=(n_nodes, n_nodes))
    adj = upper + upper.T
    if adj_type == "plain":
        return adj
    if adj_type == "norm":
        return normalize_adj_matrix(adj + sp.eye(adj.shape[0]), norm_method="left")
    if adj_type == "gcmc":
        return normalize_adj_matrix(adj, norm_method="left")
    if adj_type == "pre":
        return normalize_adj_matrix(adj, norm_method="symmetric")
    mean_adj = normalize_adj_matrix(adj, norm_method="left")
    return mean_adj + sp.eye(mean_adj.shape[0])


def build_adjacency_device(users, items, num_users, num_items, adj_type, dev):
    """The same matrices as ``build_adjacency`` without the host: (adj, adj_transposed) as DeviceCSR, built
    from the train pairs with device sorts.  Used for large graphs, where scipy needs about a minute for 10^8
    non-zeros; values agree with the scipy build to 1 ulp (numpy's and the device's pow differ in the last bit)."""
    n = num_users + num_items
    u = torch.as_tensor(users, dtype=torch.int64).to(dev)
    i = torch.as_tensor(items, dtype=torch.int64).to(dev) + num_users
    key, counts = torch.unique(torch.cat([u * n + i, i * n + u]), return_counts=True)   # duplicate pairs sum (csr_matrix)
    rows, cols, vals = torch.div(key, n, rounding_mode="floor"), key % n, counts.float()
    eye = torch.arange(n, dtype=torch.int64, device=dev)

    def with_identity(r, c, v):
        k = torch.cat([r * n + c, eye * n + eye])
        v = torch.cat([v, torch.ones(n, dtype=torch.float32, device=dev)])
        k, inv = torch.unique(k, return_inverse=True)
        return torch.div(k, n, rounding_mode="floor"), k % n, torch.zeros(k.numel(), device=dev).index_add_(0, inv, v)

    def inv_pow(deg, p):
        s = deg.pow(p)
        return torch.where(torch.isinf(s), torch.zeros_like(s), s)

    if adj_type == "norm":
        rows, cols, vals = with_identity(rows, cols, vals)
    deg = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, rows, vals)
    if adj_type == "pre":
        s_ = inv_pow(deg, -0.5)
        vals = (s_[rows] * vals) * s_[cols]
    elif adj_type != "plain":
        vals = inv_pow(deg, -1.0)[rows] * vals
        if adj_type not in ("norm", "gcmc"):
            rows, cols, vals = with_identity(rows, cols, vals)
    adj = DeviceCSR.from_device_coo(rows, cols, vals, n)
    adj_t = adj if adj_type in ("pre", "plain") else DeviceCSR.from_device_coo(cols, rows, vals, n)
    return adj, adj_t


# graphs with at least this many train pairs are built on the device (no scipy pass, no .npz cache file)
DEVICE_ADJ_MIN_PAIRS = 1 << 21


class LightGCN(AbstractRecommender):
    def __init__(self, run_config: RunConfig, model_config: Dict):
        self.config = LightGCNConfig(**model_config)
        super().__init__(run_config, self.config)
        cfg = self.config
        self.num_users, self.num_items = self.dataset.num_users, self.dataset.num_items
        self.dp = padded_width(cfg.embed_size)       # row width of the tables in HBM (zero-padded to a multiple of 64)
        self.device = _hip.require_gpu()
        # one process per GPU (torchrun): user-sharded propagation, see skrec/parallel.py
        from ..parallel import init_from_env, ShardedLightGCN
        self.dist = init_from_env()
        n_pairs = len(self.dataset.train_data)
        on_device = (not self.dist.active) and n_pairs >= DEVICE_ADJ_MIN_PAIRS
        adj = None if on_device else self._load_adj_mat(cfg.adj_type)
        self._final_is_current = False
        self.sampler_mode = getattr(run_config, "sampler_mode", None)
        self.step_losses = None
        if self.dist.active:
            if self.dp != 64:
                raise NotImplementedError("one process per GPU: the sharded engines take embed_size <= 64 (rows of 64 floats)")
            ue, ie = nn.Embedding(self.num_users, cfg.embed_size), nn.Embedding(self.num_items, cfg.embed_size)
            get_initializer("xavier_uniform")(ue.weight)
            get_initializer("xavier_uniform")(ie.weight)
            # narrower embeddings live in zero-padded 64-float rows, as on one GPU (padded_width)
            self.engine = ShardedLightGCN(self.dist, adj, pad_columns(ue.weight.detach(), 64), pad_columns(ie.weight.detach(), 64),
                                          cfg.n_layers, cfg.lr, cfg.reg, cfg.batch_size, self.device)
            self._full_user_final = None
            return
        self.engine = None
        if on_device:
            pairs = self.dataset.train_data.to_user_item_pairs()
            self.adj, self.adj_t = build_adjacency_device(pairs[:, 0], pairs[:, 1], self.num_users, self.num_items,
                                                          cfg.adj_type, self.device)
        else:
            self.adj = DeviceCSR(adj, self.device)
            # backward needs A^T; 'pre' and 'plain' are symmetric, 'norm'/'gcmc' are not
            self.adj_t = self.adj if cfg.adj_type in ("pre", "plain") else DeviceCSR(sp.csr_matrix(adj).T, self.device)
        # xavier_uniform init in the reference's order (_LightGCN.__init__, LightGCN.py:71-80)
        ue, ie = nn.Embedding(self.num_users, cfg.embed_size), nn.Embedding(self.num_items, cfg.embed_size)
        get_initializer("xavier_uniform")(ue.weight)
        get_initializer("xavier_uniform")(ie.weight)
        N = self.num_users + self.num_items
        dp = self.dp
        self.ego = pad_columns(torch.cat([ue.weight.detach(), ie.weight.detach()], dim=0), dp).to(self.device).contiguous()  # E0
        self.optimizer = DenseAdam(self.ego.view(-1), lr=cfg.lr)
        self._g_ego = self.optimizer.grad.view(N, dp)
        z = lambda: torch.zeros((N, dp), dtype=torch.float32, device=self.device)  # noqa: E731
        self.final = z()          # layer mean, E-bar
        self._x = [z(), z()]      # propagation ping-pong
        self._g_final = z()       # dL/dE-bar, then H = dL/dE-bar / (K+1)
        self._g = [z(), z()]      # backward ping-pong
        self._row_mask = None
        self._final_is_current = False

    # views -----------------------------------------------------------------------------------------
    @property
    def user_embeddings(self):
        if self.engine is not None:
            return self.engine.gather_user_table()[:, :self.config.embed_size]
        return self.ego[:self.num_users, :self.config.embed_size]

    @property
    def item_embeddings(self):
        if self.engine is not None:
            return self.engine.item_rows[:, :self.config.embed_size]
        return self.ego[self.num_users:, :self.config.embed_size]

    def _load_adj_mat(self, adj_type):
        out_dir = os.path.join(self.dataset.data_dir, f"_{self.__class__.__name__}_data")
        make_sure_dirs(out_dir)
        path = os.path.join(out_dir, f"{adj_type}_adj.npz")
        if self.dist.active and self.dist.rank != 0:
            self.dist.barrier()        # rank 0 writes the cache file, the others read it
            return sp.load_npz(path)
        if os.path.exists(path):   # same cache side file as the reference (LightGCN.py:130-140)
            adj = sp.load_npz(path)
        else:
            adj = self._create_adj_mat(adj_type)
            sp.save_npz(path, adj)
        self.dist.barrier()
        return adj

    def _create_adj_mat(self, adj_type):
        pairs = self.dataset.train_data.to_user_item_pairs()
        return build_adjacency(pairs[:, 0], pairs[:, 1], self.num_users, self.num_items, adj_type)

    # propagation -----------------------------------------------------------------------------------
    def propagate(self, last_rows=None):
        """E-bar = mean(E0, A E0, ..., A^K E0)  (_forward_gcn, LightGCN.py:89-100).
        ``last_rows`` (uint8 [N], training only): the rows of E-bar that will be read.  The LAST layer's product is then
        computed for those rows only -- every earlier layer feeds the next one and stays whole -- and E-bar is valid on
        those rows only."""
        K = self.config.n_layers
        scale = 1.0 / (K + 1)
        x = self.ego
        for k in range(K):
            y = self._x[k & 1]
            # the mean's E0 term rides in the first product's row epilogue: E-bar = scale * E0 + scale * A E0, then += per layer
            self.adj.spmm(x, y, accum=self.final, accum_scale=scale, accum_base=self.ego if k == 0 else None,
                          row_mask=last_rows if k == K - 1 else None, accum_mask=last_rows)
            x = y
        return self.final

    def _batch_rows(self, users, pos, neg):
        """uint8 [N]: 1 on the rows of [U; V] a batch touches -- the only rows of E-bar its loss reads, and the only
        non-zero rows of dL/dE-bar (reference: the gathers of _LightGCN.forward, LightGCN.py:82-87)"""
        m, L, st, nu = self._row_mask, _hip.lib(), _hip.stream(), self.num_users
        _hip.check(L.skr_mark_ids(_hip.ptr(users), users.numel(), 0, _hip.ptr(m), st))
        _hip.check(L.skr_mark_ids(_hip.ptr(pos), pos.numel(), nu, _hip.ptr(m), st))
        _hip.check(L.skr_mark_ids(_hip.ptr(neg), neg.numel(), nu, _hip.ptr(m), st))
        return m

    def train_step(self, users, pos, neg, loss_slot):
        cfg, nu = self.config, self.num_users
        K = cfg.n_layers
        n = users.numel()
        # the product A x dense is the reference's; what is skipped is arithmetic whose result is never read (rows of the
        # last forward layer outside the batch) or is a sum of zeros (the first backward hop reads dL/dE-bar, which is zero
        # outside the batch's rows).  SKR_LIGHTGCN_DENSE=1 computes everything.
        gF, gE = self._g_final, self._g_ego
        if os.environ.get("SKR_LIGHTGCN_DENSE") == "1":
            rows = None
            gF.zero_()
            self._row_mask = None
        else:
            # dL/dE-bar is zero outside the previous batch's rows: those rows are cleared (and their marks with them)
            # instead of filling the whole [N, 64] buffer
            if self._row_mask is None:
                self._row_mask = torch.zeros(self.num_users + self.num_items, dtype=torch.uint8, device=self.device)
                gF.zero_()
            else:
                _hip.check(_hip.lib().skr_clear_marked_rows(_hip.ptr(self._row_mask), self._row_mask.numel(), 1, _hip.ptr(gF), self.dp,
                                                            _hip.stream()))
            rows = self._batch_rows(users, pos, neg)
        self.propagate(last_rows=rows)
        self._final_is_current = False
        # the score part of the gradient is written already divided by K + 1: gF holds H = dL/dE-bar / (K + 1)
        _hip.check(_hip.lib().skr_bpr_step_dim(
            _hip.ptr(self.final[:nu]), _hip.ptr(self.final[nu:]), None, _hip.ptr(self.ego[:nu]), _hip.ptr(self.ego[nu:]),
            _hip.ptr(users), _hip.ptr(pos), _hip.ptr(neg), n, self.dp, 1.0 / n, cfg.reg, 1.0 / cfg.batch_size,
            _hip.ptr(gF[:nu]), _hip.ptr(gF[nu:]), None, _hip.ptr(gE[:nu]), _hip.ptr(gE[nu:]), _hip.ptr(loss_slot), 1,
            None, None, 1.0 / (K + 1), _hip.stream()))
        # backward through the mean and the K propagations: dL/dE0 += sum_k (A^T)^k H
        x = gF
        for k in range(K):
            y = self._g[k & 1]
            last = (k == K - 1)
            self.adj_t.spmm(x, y, addend=gF, accum=gE if last else None, accum_scale=1.0, col_mask=rows if k == 0 else None,
                            addend_mask=rows)
            x = y
        self.optimizer.step()

    @on_compute_stream
    def train_epoch(self, data_iter):
        self.step_losses = torch.zeros((len(data_iter), 2), dtype=torch.float32, device=self.device)
        for k, (u, i, j) in enumerate(data_iter.iter_device()):
            if self.engine is not None:      # every rank walks the same global batches and keeps its users
                self.engine.train_step(u, i, j)
                self.step_losses[k] = self.engine.loss
            else:
                self.train_step(u.contiguous(), i.contiguous(), j.contiguous(), self.step_losses[k])

    @on_compute_stream
    def fit(self):
        data_iter = PairwiseIterator(self.dataset.train_data, batch_size=self.config.batch_size, shuffle=True,
                                     drop_last=False, sampler_mode=self.sampler_mode)
        log = self.logger.info if self.dist.rank == 0 else (lambda *_: None)
        log("metrics:".ljust(12) + f"\t{self.evaluator.metrics_str}")
        early_stopping = EarlyStopping(metric="NDCG@10", patience=self.config.early_stop)
        for epoch in range(self.config.epochs):
            self.train_epoch(data_iter)
            cur_result = self.evaluate()
            log(f"epoch {epoch}:".ljust(12) + f"\t{cur_result.values_str}")
            if early_stopping(cur_result):
                log("early stop")
                break
        log("best:".ljust(12) + f"\t{early_stopping.best_result.values_str}")
        return early_stopping.best_result

    def eval(self):
        """recompute and cache the final embeddings (_LightGCN.eval, LightGCN.py:109-111)"""
        if self.engine is not None:
            e = self.engine
            e.propagate()
            full = torch.zeros((self.num_users, 64), dtype=torch.float32, device=self.device)
            full[torch.from_numpy(e.mine).to(self.device)] = e.whole_final()[:e.n_local]
            self._full_user_final = self.dist.all_reduce(full)    # every rank can rank any user
        else:
            self.propagate()
        self._final_is_current = True

    @on_compute_stream
    def evaluate(self, test_users=None):
        self.eval()
        if self.engine is None:
            return self.evaluator.evaluate(self, test_users)
        from ..parallel import sharded_evaluate
        return sharded_evaluate(self.dist, self.evaluator, self, test_users, self.device)

    def predict_factors(self):
        if not self._final_is_current:
            raise ValueError("Please first switch to 'eval' mode.")
        if self.engine is not None:
            return self._full_user_final, self.engine.whole_final()[self.engine.n_local:], None
        return self.final[:self.num_users], self.final[self.num_users:], None

    def predict(self, users):
        uf, vf, _ = self.predict_factors()
        return _hip.score_matrix(uf, users, vf, None).cpu().numpy()
