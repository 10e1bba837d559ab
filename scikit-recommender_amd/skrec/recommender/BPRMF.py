"""BPRMF on MI355X (reference: skrec/recommender/BPRMF.py).

Paper: BPR: Bayesian Personalized Ranking from Implicit Feedback (Rendle et al.).
Same config, same initialisation (weights are drawn on the CPU with torch's generator in the
reference's order, so a given ``--seed`` yields the reference's initial tables), same loss
(sum over the batch of -log sigmoid(x_ui - x_uj) + reg * 0.5 * sum of squares of the gathered rows,
BPRMF.py:117-124) and the same dense Adam.  One training step is two kinds of launches:
``skr_bpr_step`` (gather + score + loss + gradient scatter fused) and ONE ``skr_adam_step`` over the
flat [U | V | b] parameter buffer.
"""
import os
from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..io import PairwiseIterator
from ..run_config import RunConfig
from ..utils.py import EarlyStopping, ModelConfig
from ..utils.torch import get_initializer
from .base import AbstractRecommender, DenseAdam, on_compute_stream

__all__ = ["BPRMF", "BPRMFConfig"]


class BPRMFConfig(ModelConfig):
    def __init__(self, lr=1e-3, reg=1e-3, n_dim=64, batch_size=1024, epochs=1000, early_stop=200, **kwargs):
        super().__init__()
        self.lr: float = lr
        self.reg: float = reg
        self.n_dim: int = n_dim
        self.batch_size: int = batch_size
        self.epochs: int = epochs
        self.early_stop: int = early_stop

    @classmethod
    def param_space(cls):
        return {"lr": [0.001, 0.005, 0.01, 0.05], "reg": [0.0, 0.001, 0.005, 0.01, 0.05]}

    def _validate(self):
        assert isinstance(self.lr, float) and self.lr > 0
        assert isinstance(self.reg, float) and self.reg >= 0
        assert isinstance(self.n_dim, int) and self.n_dim > 0
        assert isinstance(self.batch_size, int) and self.batch_size > 0
        assert isinstance(self.epochs, int) and self.epochs >= 0
        assert isinstance(self.early_stop, int)


def _init_tables(num_users, num_items, dim):
    """CPU-side construction in the reference's order (_MF.__init__, BPRMF.py:57-75): three
    nn.Embedding constructors (each draws N(0,1)), then normal(0, 0.01) x2 and zeros."""
    ue, ie, be = nn.Embedding(num_users, dim), nn.Embedding(num_items, dim), nn.Embedding(num_items, 1)
    get_initializer("normal")(ue.weight)
    get_initializer("normal")(ie.weight)
    get_initializer("zeros")(be.weight)
    return ue.weight.detach(), ie.weight.detach(), be.weight.detach().reshape(-1)


class BPRMF(AbstractRecommender):
    def __init__(self, run_config: RunConfig, model_config: Dict):
        self.config = BPRMFConfig(**model_config)
        super().__init__(run_config, self.config)
        self.num_users, self.num_items = self.dataset.num_users, self.dataset.num_items
        from .LightGCN import pad_columns, padded_width
        self.dp = padded_width(self.config.n_dim)       # row width of the tables in HBM (zero-padded to a multiple of 64)
        self.device = _hip.require_gpu()
        U, V, b = _init_tables(self.num_users, self.num_items, self.config.n_dim)
        self.step_losses = None  # device [n_steps, 2]: (bpr sum, l2) per step of the last epoch
        self.sampler_mode = getattr(run_config, "sampler_mode", None)
        # one process per GPU (torchrun): users sharded, item table replicated -- skrec/parallel.py
        from ..parallel import init_from_env, ShardedBPRMF
        self.dist = init_from_env()
        if self.dist.active and self.dp != 64:
            raise NotImplementedError("one process per GPU: the sharded engines take n_dim <= 64 (rows of 64 floats)")
        # (narrower embeddings live in zero-padded 64-float rows, as on one GPU)
        self.engine = ShardedBPRMF(self.dist, pad_columns(U, 64), pad_columns(V, 64), b, self.config.lr, self.config.reg,
                                   self.device) if self.dist.active else None
        if self.engine is not None:
            self._full_users = None
            return
        # one flat buffer [U | V | b] => one Adam launch per step; the tables are views into it
        nu, ni, d = self.num_users, self.num_items, self.dp
        U, V = pad_columns(U, d), pad_columns(V, d)
        self._flat = torch.cat([U.reshape(-1), V.reshape(-1), b.reshape(-1)]).to(self.device).contiguous()
        # the tables as the kernels see them: [*, dp]; `user_embeddings` / `item_embeddings` are their first n_dim columns
        self._user_rows = self._flat[:nu * d].view(nu, d)
        self._item_rows = self._flat[nu * d:(nu + ni) * d].view(ni, d)
        self.user_embeddings = self._user_rows[:, :self.config.n_dim]
        self.item_embeddings = self._item_rows[:, :self.config.n_dim]
        self.item_biases = self._flat[(nu + ni) * d:]
        # SKR_ADAM_BLOCK = k: look k batches ahead and block the dense Adam over them (1: one dense launch per step).  Rows
        # wider than one 64-float block take the dense launch per step (the blocked forms name rows by their block)
        self.adam_block = max(1, min(64, int(os.environ.get("SKR_ADAM_BLOCK", "32")))) if d == 64 else 1
        # SKR_BPR_FUSED=0: two launches per step (skr_bpr_step_spread + skr_adam_block_hot) instead of one (skr_bpr_fused_step)
        self.fused_step = os.environ.get("SKR_BPR_FUSED", "1") != "0"
        self._fused = None
        self.optimizer = DenseAdam(self._flat, lr=self.config.lr, track_touch=self.adam_block <= 1)
        self._grads = (self.optimizer.grad_view(0, (nu, d)), self.optimizer.grad_view(nu * d, (ni, d)),
                       self.optimizer.grad_view((nu + ni) * d, (ni,)))

    def train_step(self, users, pos, neg, loss_slot):
        """one mini-batch; ``users/pos/neg`` are int32 device tensors"""
        gU, gV, gb = self._grads
        _hip.check(_hip.lib().skr_bpr_step_dim(
            _hip.ptr(self._user_rows), _hip.ptr(self._item_rows), _hip.ptr(self.item_biases),
            _hip.ptr(self._user_rows), _hip.ptr(self._item_rows),
            _hip.ptr(users), _hip.ptr(pos), _hip.ptr(neg), users.numel(), self.dp, 1.0, self.config.reg, 1.0,
            _hip.ptr(gU), _hip.ptr(gV), _hip.ptr(gb), _hip.ptr(gU), _hip.ptr(gV), _hip.ptr(loss_slot), 1,
            _hip.ptr(self.optimizer.touch), _hip.ptr(self.optimizer.grad) if self.optimizer.touch is not None else None,
            1.0, _hip.stream()))
        self.optimizer.step()

    @on_compute_stream
    def train_epoch(self, data_iter):
        """one epoch; the per-step host work is two ctypes calls on cached addresses"""
        self.step_losses = torch.zeros((len(data_iter), 2), dtype=torch.float32, device=self.device)
        if self.engine is not None:   # every rank walks the same global batches and keeps its users
            eng = self.engine
            if eng.adam_block > 1 and data_iter.num_neg == 1:
                (cu, ci, cj), bounds = data_iter.epoch_columns()
                for s0 in range(0, len(bounds), eng.adam_block):
                    eng.train_block(cu, ci, cj, bounds[s0:s0 + eng.adam_block], self.step_losses[s0:s0 + eng.adam_block])
                return
            for k, (u, i, j) in enumerate(data_iter.iter_device()):
                eng.train_step(u, i, j)
                self.step_losses[k] = eng.loss
            return
        L, st, opt = _hip.lib(), _hip.stream(), self.optimizer
        # the batches' loss sums land in 32 pairs of words each (skr_bpr_step_spread) and are added up once per epoch
        S = _hip.SKR_LOSS_SLOTS
        spread = torch.zeros((len(data_iter), S, 2), dtype=torch.float32, device=self.device)
        gU, gV, gb = self._grads
        pU, pV, pb = self._user_rows.data_ptr(), self._item_rows.data_ptr(), self.item_biases.data_ptr()
        pgU, pgV, pgb = gU.data_ptr(), gV.data_ptr(), gb.data_ptr()
        ploss, reg = spread.data_ptr(), self.config.reg
        kblk = self.adam_block
        # large batches: the one-launch step's workspace has 2^20 row slots (5 * batch per step); shorter blocks keep it usable
        k_fit = (1 << 20) // (5 * max(1, self.config.batch_size))
        if self.fused_step and 2 <= k_fit < kblk:
            kblk = k_fit
        if kblk <= 1 or data_iter.num_neg != 1:
            pgrad, pflat, pm, pv = (t.data_ptr() for t in (opt.grad, opt.flat, opt.m, opt.v))
            ptouch = opt.touch.data_ptr() if opt.touch is not None else None    # None: plain dense step, every gradient read
            pgrad_base = pgrad if ptouch is not None else None
            n_par = opt.flat.numel()
            for k, (u, i, j) in enumerate(data_iter.iter_device()):
                # slices of the contiguous epoch columns are themselves contiguous (num_neg == 1)
                rc = L.skr_bpr_step_dim(pU, pV, pb, pU, pV, u.data_ptr(), i.data_ptr(), j.data_ptr(), u.numel(), self.dp, 1.0, reg, 1.0,
                                        pgU, pgV, pgb, pgU, pgV, ploss + 8 * S * k, S, ptouch, pgrad_base, 1.0, st)
                opt.t += 1
                rc |= L.skr_adam_step(pflat, pgrad, pm, pv, n_par, opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.t, 1,
                                      ptouch, st)
                if rc:
                    _hip.check(rc)
            self.step_losses = spread.sum(1)
            return
        # Temporally blocked dense Adam (csrc/train.hip, K2b): the epoch's batches are known, so for every block of
        # `kblk` steps the rows no batch of the block touches get their kblk zero-gradient updates in one pass and
        # only the touched rows are stepped batch by batch -- the same updates in the same arithmetic, bit-identical
        # to the loop above, with 1/kblk of its optimiser traffic.
        (cu, ci, cj), bounds = data_iter.epoch_columns()
        nu, ni = self.num_users, self.num_items
        pcu, pci, pcj = cu.data_ptr(), ci.data_ptr(), cj.data_ptr()
        bpr = L.skr_bpr_step_spread

        def block_ids(u_, i_, j_, dim):
            # 64-float blocks of the flat [U | V | b] buffer the batches touch: user rows, item rows, bias words
            return torch.cat([u_, i_ + nu, j_ + nu, (i_ >> 6) + (nu + ni), (j_ >> 6) + (nu + ni)], dim=dim)
        # all full blocks of the epoch at once (one row of ids per block, step-major: 5 * batch entries per step): no
        # per-block tensor arithmetic in the loop, and a hot step can name just its own and the next batch's rows
        bsz = self.config.batch_size
        n_full_blocks = (len(cu) // bsz) // kblk
        rows = n_full_blocks * kblk * bsz
        ids_all = block_ids(*(c[:rows].view(n_full_blocks, kblk, bsz) for c in (cu, ci, cj)), dim=2).view(n_full_blocks, -1) \
            if n_full_blocks and not (self.fused_step and kblk * 5 * bsz <= (1 << 20)) else None
        first = 0
        if self.fused_step and n_full_blocks and kblk * 5 * bsz <= (1 << 20):      # slot numbers have 20 bits
            # full blocks: ONE launch per step -- the hot rows' Adam is evaluated inside the BPR kernel (csrc/train.hip K2c)
            from .fused import FusedBlocks
            if self._fused is None:
                self._fused = FusedBlocks(opt, 0, nu, nu + ni, reg)
            fb = self._fused
            fb.run_blocks(pcu, pci, pcj, n_full_blocks, kblk, bsz, ploss, 8 * S)
            first = n_full_blocks * kblk
        for s0 in range(first, len(bounds), kblk):
            blk = bounds[s0:s0 + kblk]
            lo, hi = blk[0][0], blk[-1][1]
            if s0 // kblk < n_full_blocks:
                opt.begin_block(ids_all[s0 // kblk], len(blk), per_step=5 * bsz)
            else:
                opt.begin_block(block_ids(cu[lo:hi], ci[lo:hi], cj[lo:hi], 0), len(blk))
            hot, pp, pg, pm, pv, n_par, pids, nids, pclaim, _, t0, kk, per = opt._hot
            lr, b1, b2, eps, t = opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.t
            rc = 0
            for k, (a, b) in enumerate(blk, start=s0):      # two launches per step, on cached integer addresses
                t += 1
                rc |= bpr(pU, pV, pb, pU, pV, pcu + 4 * a, pci + 4 * a, pcj + 4 * a, b - a, 1.0, reg, 1.0,
                          pgU, pgV, pgb, pgU, pgV, ploss + 8 * S * k, None, None, st)
                if per is not None and t < t0 + kk:         # rows of this batch and of the next; the last step names all
                    rc |= hot(pp, pg, pm, pv, n_par, lr, b1, b2, eps, t0, t, pids + 4 * per * (t - t0 - 1), 2 * per, 0, 64, pclaim, st)
                else:
                    rc |= hot(pp, pg, pm, pv, n_par, lr, b1, b2, eps, t0, t, pids, nids, 0, 64, pclaim, st)
            opt.t = t
            if rc:
                _hip.check(rc)
        opt.end_blocks()
        self.step_losses = spread.sum(1)

    @on_compute_stream
    def fit(self):
        data_iter = PairwiseIterator(self.dataset.train_data, batch_size=self.config.batch_size, shuffle=True,
                                     drop_last=False, sampler_mode=self.sampler_mode)
        log = self.logger.info if self.dist.rank == 0 else (lambda *_: None)
        log("metrics:".ljust(12) + f"\t{self.evaluator.metrics_str}")
        early_stopping = EarlyStopping(metric="NDCG@10", patience=self.config.early_stop)
        # between epochs this loop asks numpy's global generator for nothing but the epoch permutations: the next one
        # is drawn on a helper thread while the GPU trains (same numbers, same generator state afterwards)
        data_iter.epoch_ahead(True)
        try:
            for epoch in range(self.config.epochs):
                self.train_epoch(data_iter)
                cur_result = self.evaluate()
                log(f"epoch {epoch}:".ljust(12) + f"\t{cur_result.values_str}")
                if early_stopping(cur_result):
                    log("early stop")
                    break
        finally:
            data_iter.epoch_ahead(False)
        log("best:".ljust(12) + f"\t{early_stopping.best_result.values_str}")
        return early_stopping.best_result

    @on_compute_stream
    def evaluate(self, test_users=None):
        if self.engine is None:
            return self.evaluator.evaluate(self, test_users)
        # sharded: every rank ranks its share of the test users, the fp64 metric sums are all-reduced
        from ..utils.py import MetricReport
        self._full_users = self.engine.gather_user_table()
        ev = self.evaluator
        users = list(ev.user_pos_test.keys()) if test_users is None else [u for u in test_users if u in ev.user_pos_test]
        mine = [u for u in users if u % self.dist.world == self.dist.rank]
        _, sums, n = ev.per_user_rows(self, mine)
        tot = torch.from_numpy(np.concatenate([sums, [float(n)]])).to(self.device)
        tot = self.dist.all_reduce(tot).cpu().numpy()
        final = (tot[:-1] / max(tot[-1], 1.0)).astype(np.float32)
        final = final.reshape(ev.metrics_num, ev.max_top)[:, ev.top_show - 1].reshape(-1)
        return MetricReport(ev.metrics_list, final)

    def predict_factors(self):
        if self.engine is not None:
            if self._full_users is None:
                self._full_users = self.engine.gather_user_table()
            return self._full_users, self.engine.item_rows, self.engine.item_bias
        return self._user_rows, self._item_rows, self.item_biases

    def predict(self, users) -> np.ndarray:
        """dense [len(users), num_items] scores (API surface of BPRMF.py:145-147; the evaluator uses
        the fused kernel through ``predict_factors`` instead)"""
        ut, it, b = self.predict_factors()
        return _hip.score_matrix(ut, users, it, b).cpu().numpy()
