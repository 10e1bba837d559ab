"""LayerGCN on MI355X (reference: skrec/recommender/LayerGCN.py).

Paper: Layer-refined Graph Convolutional Networks for Recommendation (Zhou et al., ICDE 2023).
Per layer (LayerGCN.py:212-216):  Y = A X ;  w = cos(Y, E0) row-wise ;  X' = w * Y  (the re-weighted
tensor feeds the next layer); output = SUM of the K refined layers (E0 itself excluded, :218);
loss = sum_b -log sigmoid(x_ui - x_uj) + reg * 0.5 * ||ego rows||^2 (:231, :242, :252); dense Adam,
constant learning rate (the LambdaLR factor is 1.0 ** (epoch / 50), :274-276).
``skr_csr_spmm`` does the propagation, ``skr_layer_refine_fwd/_bwd`` the cosine re-weighting and its
hand-derived backward; A is symmetric so the backward propagation is the same kernel.
Edge dropout (``dropout > 0``, :133-152): once per epoch a fraction of the edges is pruned --
alternately by degree-weighted sampling without replacement (``torch.multinomial`` of the normalised
edge values, first epoch) and uniformly at random -- and the kept graph is re-normalised; training
uses the pruned adjacency, evaluation the full one.  ``prune_draws="reference"`` (the default; env
``SKR_PRUNE_DRAWS``) makes the two draws exactly as the reference's CPU run makes them -- Python's
``random.sample`` (:141) and ``torch.multinomial`` of the CPU edge values on torch's global CPU
generator (:144) -- so the pruned graphs, and with them the whole trajectory, replay the reference's
(tests/golden/golden_layergcn_dropout.npz); everything after the draw (gather of the kept edges,
re-normalisation, CSR build, training) stays on the device.  ``prune_draws="device"`` draws on the GPU
instead (the library's keyed-bijection shuffle for the uniform epoch, torch's device multinomial for
the weighted one): equal in law only, for graphs where a host-side draw per epoch is too slow.
"""
import os
import random
from typing import Dict

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from .. import _hip
from ..io import PairwiseIterator
from ..run_config import RunConfig
from ..utils.py import EarlyStopping, ModelConfig
from .base import AbstractRecommender, DenseAdam, on_compute_stream
from .LightGCN import DEVICE_ADJ_MIN_PAIRS, DeviceCSR, pad_columns, padded_width

__all__ = ["LayerGCN", "LayerGCNConfig"]


class LayerGCNConfig(ModelConfig):
    def __init__(self, lr=1e-3, reg=1e-2, embed_dim=64, n_layers=4, dropout=0.0, batch_size=2048, epochs=1000,
                 early_stop=200, prune_draws=None, **kwargs):
        super().__init__()
        self.lr: float = lr
        self.reg: float = reg
        self.embed_dim: int = embed_dim
        self.n_layers: int = n_layers
        self.dropout: float = dropout
        self.batch_size: int = batch_size
        self.epochs: int = epochs
        self.early_stop: int = early_stop
        # how the edge-pruning draws are made when dropout > 0 (not a reference option): "reference" | "device"
        self.prune_draws: str = prune_draws or os.environ.get("SKR_PRUNE_DRAWS", "reference")

    @classmethod
    def param_space(cls):
        return {"n_layers": [4], "reg": [1e-02, 1e-03, 1e-04, 1e-05], "dropout": [0.0, 0.1, 0.2]}


def build_layergcn_adjacency(inter_coo, n_users, n_items):
    """get_norm_adj_mat (LayerGCN.py:173-197): binary bipartite adjacency, D^-1/2 A D^-1/2 with
    1e-7 added to the degrees, computed in float64 and rounded to float32 at the end."""
    n = n_users + n_items
    r = np.asarray(inter_coo.row, dtype=np.int64)
    c = np.asarray(inter_coo.col, dtype=np.int64) + n_users
    A = sp.csr_matrix((np.ones(2 * len(r), dtype=np.float32), (np.concatenate([r, c]), np.concatenate([c, r]))),
                      shape=(n, n))
    A.data[:] = 1.0  # duplicate pairs collapse to one edge (the reference goes through a dict)
    deg = np.asarray((A > 0).sum(axis=1)).reshape(-1) + 1e-7
    d = sp.diags(np.power(deg, -0.5))
    return (d * A * d).astype(np.float32)


class LayerGCN(AbstractRecommender):
    def __init__(self, run_config: RunConfig, model_config: Dict):
        self.config = LayerGCNConfig(**model_config)
        super().__init__(run_config, self.config)
        cfg = self.config
        self.num_users, self.num_items = self.dataset.num_users, self.dataset.num_items
        self.dp = padded_width(cfg.embed_dim)        # row width of the tables in HBM (zero-padded to a multiple of 64)
        if not 0.0 <= cfg.dropout < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        if cfg.prune_draws not in ("reference", "device"):
            raise ValueError("prune_draws must be 'reference' or 'device'")
        self.device = _hip.require_gpu()
        from ..parallel import init_from_env, ShardedLayerGCN
        self.dist = init_from_env()      # one process per GPU under torchrun; world 1 otherwise
        inter = self.dataset.train_data.to_coo_matrix().astype(np.float32)
        # parameters first, like _LayerGCN.__init__ (:114-115): xavier_uniform on plain tensors
        ue = nn.init.xavier_uniform_(torch.empty(self.num_users, cfg.embed_dim))
        ie = nn.init.xavier_uniform_(torch.empty(self.num_items, cfg.embed_dim))
        self.pruning_random = False                     # LayerGCN.py:121
        self.step_losses = None
        self.sampler_mode = getattr(run_config, "sampler_mode", None)
        self.engine = None
        if self.dist.active and cfg.embed_dim > 64:
            raise NotImplementedError("one process per GPU: the sharded engines take embed_dim <= 64 (rows of 64 floats)")
        if self.dist.active:
            # user-sharded rows, replicated item rows (skrec/parallel.py); duplicate pairs collapse to one edge
            pairs = np.unique(np.stack([inter.row, inter.col], 1).astype(np.int64), axis=0)
            self._edge_u = torch.from_numpy(pairs[:, 0].copy()).to(self.device)
            self._edge_i = torch.from_numpy(pairs[:, 1].copy()).to(self.device)
            self._edge_values = self._normalize_edges(self._edge_u, self._edge_i)
            # narrower embeddings live in zero-padded 64-float rows, as on one GPU (LightGCN.padded_width)
            self.engine = ShardedLayerGCN(self.dist, self._edge_u, self._edge_i, self.num_users, self.num_items,
                                          pad_columns(ue, 64), pad_columns(ie, 64), cfg.n_layers, cfg.lr, cfg.reg, self.device)
            self._full_user_out = None
            return
        if inter.nnz >= DEVICE_ADJ_MIN_PAIRS:     # large graph: no scipy pass (same values: float64 degrees, fp32 result)
            self.adj = self._device_adjacency(inter)
        else:
            self.adj = DeviceCSR(build_layergcn_adjacency(inter, self.num_users, self.num_items), self.device)
        self.train_adj = self.adj                       # masked_adj of the reference (LayerGCN.py:119,135)
        # edge list + normalised edge values for the pruning step (get_edge_info, LayerGCN.py:165-171)
        self._edge_u = torch.from_numpy(np.asarray(inter.row, dtype=np.int64)).to(self.device)
        self._edge_i = torch.from_numpy(np.asarray(inter.col, dtype=np.int64)).to(self.device)
        self._edge_values = self._normalize_edges(self._edge_u, self._edge_i)
        N = self.num_users + self.num_items
        dp = self.dp
        self.ego = pad_columns(torch.cat([ue, ie], dim=0), dp).to(self.device).contiguous()
        self.optimizer = DenseAdam(self.ego.view(-1), lr=cfg.lr)
        self._g_ego = self.optimizer.grad.view(N, dp)
        z = lambda: torch.zeros((N, dp), dtype=torch.float32, device=self.device)  # noqa: E731
        K = cfg.n_layers
        self.out = z()                                   # sum of refined layers
        self._y = [z() for _ in range(K)]                # A X_k kept for the backward
        self._w = [torch.zeros(N, dtype=torch.float32, device=self.device) for _ in range(K)]
        self._z = [z(), z()]                             # refined layer ping-pong
        self._g_out = z()
        self._t = [z(), z()]

    @property
    def user_embeddings(self):
        if self.engine is not None:
            return self.engine.gather_user_table()[:, :self.config.embed_dim]
        return self.ego[:self.num_users, :self.config.embed_dim]

    @property
    def item_embeddings(self):
        if self.engine is not None:
            return self.engine.item_rows[:, :self.config.embed_dim]
        return self.ego[self.num_users:, :self.config.embed_dim]

    def _device_adjacency(self, inter):
        """get_norm_adj_mat (LayerGCN.py:173-197) on the device: binary bipartite graph (duplicate pairs are one
        edge), D^-1/2 A D^-1/2 with 1e-7 added to the degrees, float64 arithmetic rounded to float32"""
        dev, nu, n = self.device, self.num_users, self.num_users + self.num_items
        key = torch.unique(torch.from_numpy(np.asarray(inter.row, dtype=np.int64)).to(dev) * self.num_items
                           + torch.from_numpy(np.asarray(inter.col, dtype=np.int64)).to(dev))
        u, i = torch.div(key, self.num_items, rounding_mode="floor"), key % self.num_items
        # degrees as exact integer counts (a float64 index_add_ of ones is the same numbers, through 10^8 double atomics)
        du = (torch.bincount(u, minlength=nu).double() + 1e-7).pow(-0.5)
        di = (torch.bincount(i, minlength=self.num_items).double() + 1e-7).pow(-0.5)
        vals = (du[u] * di[i]).float()
        return DeviceCSR.from_device_coo(torch.cat([u, i + nu]), torch.cat([i + nu, u]), torch.cat([vals, vals]), n)

    def _normalize_edges(self, u, i):
        """_normalize_adj_m (LayerGCN.py:154-163): 1/sqrt((deg_u + 1e-7)(deg_i + 1e-7)) on the given edges"""
        ones = torch.ones(u.numel(), dtype=torch.float32, device=u.device)
        row_sum = 1e-7 + torch.zeros(self.num_users, device=u.device).index_add_(0, u, ones)
        col_sum = 1e-7 + torch.zeros(self.num_items, device=u.device).index_add_(0, i, ones)
        return torch.pow(row_sum, -0.5)[u] * torch.pow(col_sum, -0.5)[i]

    def _edge_values_host(self):
        """edge_values exactly as the reference's CPU run holds them (get_edge_info -> _normalize_adj_m, LayerGCN.py:
        154-171: integer degree sums, + 1e-7 in float32, torch-CPU pow) -- torch.multinomial's result depends on every bit"""
        if getattr(self, "_edge_values_cpu", None) is None:
            u, i = self._edge_u.cpu(), self._edge_i.cpu()
            ones = torch.ones(u.numel(), dtype=torch.float32)
            row_sum = 1e-7 + torch.zeros(self.num_users).index_add_(0, u, ones)
            col_sum = 1e-7 + torch.zeros(self.num_items).index_add_(0, i, ones)
            self._edge_values_cpu = torch.pow(row_sum, -0.5)[u] * torch.pow(col_sum, -0.5)[i]
        return self._edge_values_cpu

    def pre_epoch_processing(self):
        """edge pruning, once per epoch (LayerGCN.py:133-152); a no-op when dropout == 0"""
        if self.config.dropout <= 0.0:
            if self.engine is not None:
                self.engine.set_train_edges(None)
            else:
                self.train_adj = self.adj
            return
        n_edges = self._edge_values.numel()
        keep_len = int(n_edges * (1.0 - self.config.dropout))
        if self.config.prune_draws == "reference":
            # the reference's own draws, on the host like its CPU run: same generators, same calls, same order
            if self.pruning_random:
                keep = torch.tensor(random.sample(range(n_edges), keep_len)).to(self.device)      # LayerGCN.py:141
            else:
                keep = torch.multinomial(self._edge_values_host(), keep_len).to(self.device)     # LayerGCN.py:144
        elif self.pruning_random:
            keep = torch.empty(keep_len, dtype=torch.int32, device=self.device)
            _hip.check(_hip.lib().skr_shuffle_permutation(random.getrandbits(63), n_edges, keep_len, _hip.ptr(keep),
                                                          _hip.stream()))
            keep = keep.long()
        else:   # prune edges of high-degree nodes preferentially: keep ~ normalised edge value
            keep = torch.multinomial(self._edge_values, keep_len)
        self.pruning_random = True ^ self.pruning_random
        if self.engine is not None:
            # every rank must prune the SAME edges: rank 0's draw is broadcast (one int64 per kept edge)
            if self.dist.rank != 0:
                keep.zero_()
            self.dist.all_reduce(keep)
            self.engine.set_train_edges(self._edge_u[keep], self._edge_i[keep])
            return
        u, i = self._edge_u[keep], self._edge_i[keep]
        vals = self._normalize_edges(u, i)
        n = self.num_users + self.num_items
        rows = torch.cat([u, i + self.num_users])
        cols = torch.cat([i + self.num_users, u])
        self.train_adj = DeviceCSR.from_device_coo(rows, cols, torch.cat([vals, vals]), n)

    def forward(self, adj=None, last_rows=None):
        """``last_rows`` (uint8 [N], training only): the rows of ``out`` that will be read -- the LAST layer's product is
        computed for those rows only (every earlier layer feeds the next one and stays whole)"""
        L, st = _hip.lib(), _hip.stream()
        N = self.ego.shape[0]
        adj = adj if adj is not None else self.adj
        x = self.ego
        K = self.config.n_layers
        for k in range(K):
            zk = self._z[k & 1]
            # the refinement (w = cos(A x, E0), z = w * A x, out += z) rides in the product's row epilogue; the first layer
            # initialises `out` (no fill)
            adj.spmm(x, self._y[k], row_mask=last_rows if k == K - 1 else None, accum=self.out, accum_init=(k == 0),
                     refine_fwd=(self.ego, self._w[k], zk), accum_mask=last_rows)
            x = zk
        return self.out

    def train_step(self, users, pos, neg, loss_slot):
        cfg, nu = self.config, self.num_users
        L, st = _hip.lib(), _hip.stream()
        N = self.ego.shape[0]
        adj = self.train_adj
        # not computed: rows of the last layer's product that the batch does not read, and -- in the first backward hop --
        # the products with rows of dY_K that are zero (dL/d out is zero outside the batch's rows, and the refinement's
        # backward is row-local).  SKR_LIGHTGCN_DENSE=1 computes everything.
        gO, gE = self._g_out, self._g_ego
        if os.environ.get("SKR_LIGHTGCN_DENSE") == "1":
            rows = None
            gO.zero_()
            self._row_mask = None
        else:
            # dL/d out is zero outside the previous batch's rows: those are cleared (with their marks), not the whole buffer
            if getattr(self, "_row_mask", None) is None:
                self._row_mask = torch.zeros(self.num_users + self.num_items, dtype=torch.uint8, device=self.device)
                gO.zero_()
            else:
                _hip.check(L.skr_clear_marked_rows(_hip.ptr(self._row_mask), N, 1, _hip.ptr(gO), self.dp, st))
            rows = self._batch_rows(users, pos, neg)
        self.forward(adj, last_rows=rows)
        _hip.check(L.skr_bpr_step_dim(
            _hip.ptr(self.out[:nu]), _hip.ptr(self.out[nu:]), None, _hip.ptr(self.ego[:nu]), _hip.ptr(self.ego[nu:]),
            _hip.ptr(users), _hip.ptr(pos), _hip.ptr(neg), users.numel(), self.dp, 1.0, cfg.reg, 1.0,
            _hip.ptr(gO[:nu]), _hip.ptr(gO[nu:]), None, _hip.ptr(gE[:nu]), _hip.ptr(gE[nu:]), _hip.ptr(loss_slot), 1,
            None, None, 1.0, st))
        # backward: dZ_K = gO ; dY_k, dE0 += refine_bwd(dZ_k) ; dZ_{k-1} = gO + A dY_k ; dE0 += A dY_1.  dZ_K is zero outside the
        # batch's rows, so the top refinement only visits those (the plan's product then skips the other columns of dY_K;
        # the plan-free kernel reads every column: there the skipped rows are written as zeros); every further refinement
        # rides in the row epilogue of the hop that produces its dZ.
        K = cfg.n_layers
        dy, nxt = self._t
        _hip.check(L.skr_layer_refine_bwd_masked(_hip.ptr(self._y[K - 1]), _hip.ptr(self.ego), _hip.ptr(self._w[K - 1]), _hip.ptr(gO), N, self.dp,
                                                 _hip.ptr(dy), _hip.ptr(gE), _hip.ptr(rows), 0 if adj.uses_plan() else 1, st))
        for k in range(K - 1, -1, -1):
            cm = rows if k == K - 1 else None
            if k > 0:
                adj.spmm(dy, nxt, addend=gO, col_mask=cm, refine_bwd=(self.ego, self._w[k - 1], self._y[k - 1], gE), addend_mask=rows)
                dy, nxt = nxt, dy
            else:
                adj.spmm(dy, None, accum=gE, accum_scale=1.0, col_mask=cm)
        self.optimizer.step()

    def _batch_rows(self, users, pos, neg):
        """uint8 [N]: 1 on the rows of [U; V] a batch touches (the gathers of calculate_loss, LayerGCN.py:245-253)"""
        m, L, st, nu = self._row_mask, _hip.lib(), _hip.stream(), self.num_users
        _hip.check(L.skr_mark_ids(_hip.ptr(users), users.numel(), 0, _hip.ptr(m), st))
        _hip.check(L.skr_mark_ids(_hip.ptr(pos), pos.numel(), nu, _hip.ptr(m), st))
        _hip.check(L.skr_mark_ids(_hip.ptr(neg), neg.numel(), nu, _hip.ptr(m), st))
        return m

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
            self.pre_epoch_processing()
            self.train_epoch(data_iter)
            cur_result = self.evaluate()
            log(f"epoch {epoch}:".ljust(12) + f"\t{cur_result.values_str}")
            if early_stopping(cur_result):
                log("early stop")
                break
        log("best:".ljust(12) + f"\t{early_stopping.best_result.values_str}")
        return early_stopping.best_result

    def _refresh_outputs(self):
        """one full-graph propagation per evaluation (the reference repeats it per batch, :255-262)"""
        if self.engine is not None:
            e = self.engine
            e.propagate(train=False)
            self._full_user_out = e.gather_user_rows(e.whole_out()[:e.n_local])   # every rank can rank any user
        else:
            self.forward()

    @on_compute_stream
    def evaluate(self, test_users=None):
        self._refresh_outputs()
        if self.engine is None:
            return self.evaluator.evaluate(self, test_users)
        from ..parallel import sharded_evaluate
        return sharded_evaluate(self.dist, self.evaluator, self, test_users, self.device)

    def predict_factors(self):
        if self.engine is not None:
            return self._full_user_out, self.engine.whole_out()[self.engine.n_local:], None
        return self.out[:self.num_users], self.out[self.num_users:], None

    def predict(self, users):
        self._refresh_outputs()
        uf, vf, _ = self.predict_factors()
        return _hip.score_matrix(uf, users, vf, None).cpu().numpy()
