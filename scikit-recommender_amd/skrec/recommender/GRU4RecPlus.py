"""GRU4RecPlus on MI355X (reference: skrec/recommender/GRU4RecPlus.py, a TensorFlow-1.14 graph).

Paper: Recurrent Neural Networks with Top-k Gains for Session-based Recommendations (Hidasi and
Karatzoglou).  Same constructor, config fields, session-parallel ``fit`` loop (:202-254), inference sweep
(:256-302) and ``predict`` (:309-324) as the reference; the graph itself (embedding lookup, GRUCell stack,
logits against the batch's own next items plus ``n_sample`` popularity^alpha negatives, bpr_max / top1_max
loss, l2 term, TF-style dense Adam) runs through ``skr_gru_cell_fwd/_bwd``, ``skr_session_loss``,
``skr_session_out_grads``, ``skr_scatter_add_rows`` and ``skr_adam_step`` (csrc/gru.hip, train.hip).

What differs, and why:

* PARITY UNPINNED against the reference: TensorFlow is not installed, so the reference cannot be run and holds
  no recorded output for this model; tests compare the HIP path with a torch-CPU restatement of the graph.
* Initial values: truncated normal(0, 0.01) tables and glorot-uniform kernels like the reference, but drawn
  from torch's CPU generator (TF's random stream is not reproducible outside TF).
* ``_get_user_embeddings`` advances ALL users together, one GRU step per history position (users whose
  history has ended keep their state), instead of 128 at a time: the embeddings are the same, the sweep is
  max-history-length launches instead of sum-of-lengths / 128.
* negative samples come from numpy's global generator exactly as in the reference (:198-200), so
  ``np.random.seed`` reproduces its sample stream.
"""
from typing import Dict, List

import os

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..run_config import RunConfig
from ..utils.py import EarlyStopping, ModelConfig
from .base import AbstractRecommender, DenseAdam, on_compute_stream

__all__ = ["GRU4RecPlus", "GRU4RecPlusConfig", "SessionGRU", "ShardedSessionGRU"]

_HIDDEN = {"tanh": 0, "relu": 1}
_FINAL = {"linear": 0, "relu": 1, "leaky_relu": 2}
_LOSS = {"bpr_max": 0, "top1_max": 1}


class GRU4RecPlusConfig(ModelConfig):
    def __init__(self, lr=0.001, reg=0.0, bpr_reg=1.0, layers=[64], batch_size=128, loss="bpr_max", hidden_act="tanh",
                 final_act="linear", n_sample=2048, sample_alpha=0.75, epochs=500, early_stop=100, **kwargs):
        super().__init__()
        self.lr: float = lr
        self.reg: float = reg
        self.bpr_reg: float = bpr_reg
        self.layers: List[int] = layers
        self.batch_size: int = batch_size
        self.loss: str = loss                  # top1_max, bpr_max
        self.hidden_act: str = hidden_act      # relu, tanh
        self.final_act: str = final_act        # linear, relu, leaky_relu
        self.n_sample: int = n_sample
        self.sample_alpha: float = sample_alpha
        self.epochs: int = epochs
        self.early_stop: int = early_stop

    def _validate(self):
        assert isinstance(self.lr, float) and self.lr > 0
        assert isinstance(self.reg, float) and self.reg >= 0
        assert isinstance(self.bpr_reg, float) and self.bpr_reg >= 0
        assert isinstance(self.layers, list)
        assert isinstance(self.batch_size, int) and self.batch_size > 0
        assert isinstance(self.loss, str) and self.loss in {"top1_max", "bpr_max"}
        assert isinstance(self.hidden_act, str) and self.hidden_act in {"relu", "tanh"}
        assert isinstance(self.final_act, str) and self.final_act in {"linear", "relu", "leaky_relu"}
        assert isinstance(self.n_sample, int) and self.n_sample >= 0
        assert isinstance(self.sample_alpha, float) and 0 < self.sample_alpha <= 1
        assert isinstance(self.epochs, int) and self.epochs >= 0
        assert isinstance(self.early_stop, int)


class SessionGRU(object):
    """Device state + one training / inference step of the GRU4RecPlus graph.  Parameters live in ONE flat
    buffer [E_in | E_out | b_out | (Wg, bg, Wc, bc) per layer], every section starting on a 64-float block,
    stepped by a single dense Adam launch with TensorFlow's epsilon placement."""

    def __init__(self, E_in, cells, E_out, b_out, hidden_act="tanh", final_act="linear", loss="bpr_max", bpr_reg=1.0,
                 reg=0.0, lr=1e-3, device=None):
        self.device = dev = device if device is not None else _hip.require_gpu()
        if hidden_act not in _HIDDEN:
            raise ValueError("There is not hidden_act named '%s'." % hidden_act)
        if final_act not in _FINAL:
            raise ValueError("There is not final_act named '%s'." % final_act)
        if loss not in _LOSS:
            raise ValueError("There is not loss named '%s'." % loss)
        self.hidden_act, self.final_act, self.loss_kind = _HIDDEN[hidden_act], _FINAL[final_act], _LOSS[loss]
        self.bpr_reg, self.reg = float(bpr_reg), float(reg)
        t32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        E_in, E_out, b_out = t32(E_in), t32(E_out), t32(b_out).reshape(-1)
        cells = [tuple(t32(w) for w in cell) for cell in cells]
        self.n_items, self.in_dim = E_in.shape
        self.hids = [int(c[2].shape[1]) for c in cells]
        dims_in = [self.in_dim] + self.hids[:-1]
        for (Wg, bg, Wc, bc), i_d, h in zip(cells, dims_in, self.hids):
            assert Wg.shape == (i_d + h, 2 * h) and bg.shape == (2 * h,) and Wc.shape == (i_d + h, h) and bc.shape == (h,)
            if h not in (32, 64, 128) or i_d > 128:
                raise NotImplementedError("the MI355X GRU kernels take layer sizes 32, 64 or 128")
        assert E_out.shape == (self.n_items, self.hids[-1]) and b_out.shape == (self.n_items,)
        sections = [E_in, E_out, b_out] + [w for c in cells for w in c]
        offs, n = [], 0
        for s in sections:
            offs.append(n)
            n += (s.numel() + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        views = []
        for s, o in zip(sections, offs):
            v = self.flat[o:o + s.numel()].view(s.shape)
            v.copy_(s)
            views.append(v)
        self._offs = offs
        self.opt = DenseAdam(self.flat, lr=lr, track_touch=True, tf_epsilon=True)
        self._blk_left = 0
        self.opt.touch[offs[3] // 64:] = 2                # the recurrent weights get a gradient every step
        gviews = [self.opt.grad[o:o + s.numel()].view(s.shape) for s, o in zip(sections, offs)]
        self.E_in, self.E_out, self.b_out = views[:3]
        self.gE_in, self.gE_out, self.gb_out = gviews[:3]
        self.cells = [tuple(views[3 + 4 * l:7 + 4 * l]) for l in range(len(cells))]
        self.gcells = [tuple(gviews[3 + 4 * l:7 + 4 * l]) for l in range(len(cells))]
        self.dims_in = dims_in
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._bufs = {}

    def _buf(self, name, shape):
        key = (name, tuple(shape))
        if key not in self._bufs:
            self._bufs[key] = torch.empty(shape, dtype=torch.float32, device=self.device)
        return self._bufs[key]

    def zero_states(self, b):
        return [torch.zeros((b, h), dtype=torch.float32, device=self.device) for h in self.hids]

    # ---- the dense Adam blocked in time (csrc/train.hip K2b, TF arithmetic): the inputs and targets of the next k steps are
    # known (the session-parallel schedule is host logic on the data, the negatives' uniforms can be drawn ahead in the
    # reference's order), so the 64-float blocks of the flat buffer that none of the k steps names get their k
    # zero-gradient updates in ONE pass on a side stream and only the named rows -- at most b + (b + n_sample) item rows,
    # their bias words and the GRU kernels -- are stepped batch by batch.  Every parameter receives every update in
    # skr_adam_step_tf's arithmetic: bit-identical to a dense launch per step (tests/test_gpu_gru.py).
    def block_ids(self, xs, ys):
        """int32 [k, per]: the 64-float blocks of the flat buffer that step s of the block touches -- E_in rows of xs[s],
        E_out rows and b_out words of ys[s], every block of the GRU kernels"""
        k = xs.shape[0]
        dev = self.device
        o_in, o_out, o_b, o_cells = (o // 64 for o in self._offs[:4])
        hn = self.hids[-1]

        def rows(o, idx, d):
            first = o + ((idx.long() * d) >> 6)
            c = max(1, d // 64)
            return (first.unsqueeze(2) + torch.arange(c, device=dev)).reshape(k, -1)
        cells = torch.arange(o_cells, (self.flat.numel() + 63) // 64, device=dev).expand(k, -1)
        return torch.cat([rows(o_in, xs, self.in_dim), rows(o_out, ys, hn), o_b + (ys.long() >> 6), cells], dim=1).int().contiguous()

    def begin_block(self, xs, ys):
        """xs int32 [k, b], ys int32 [k, b + n_sample] (device): the next k calls of train_step will be given exactly
        these inputs / targets, in this order"""
        assert self._blk_left == 0 and 1 <= xs.shape[0] <= 64
        ids = self.block_ids(xs, ys)
        self.opt.begin_block(ids.reshape(-1), xs.shape[0], per_step=ids.shape[1])
        self._blk_left = xs.shape[0]

    def end_blocks(self):
        """join the cold passes' stream (before anything else reads the parameters)"""
        assert self._blk_left == 0
        self.opt.end_blocks()

    def _optimizer_step(self):
        if self._blk_left > 0:
            self.opt.hot_step()
            self._blk_left -= 1
        else:
            self.opt.step()

    def forward(self, x_index, states, active=None, save=False, tag="f"):
        """one step of the stack: -> new states (fresh buffers per `tag`), optionally keeping r, u, c"""
        L, st = _hip.lib(), _hip.stream()
        b = x_index.numel()
        src, idx = self.E_in, x_index
        new_states, saved = [], []
        for l, ((Wg, bg, Wc, bc), i_d, h) in enumerate(zip(self.cells, self.dims_in, self.hids)):
            hn = self._buf(f"{tag}_h{l}", (b, h))
            r = u = c = None
            if save:
                r, u, c = (self._buf(f"{tag}_{n}{l}", (b, h)) for n in "ruc")
            _hip.check(L.skr_gru_cell_fwd(_hip.ptr(src), _hip.ptr(idx), _hip.ptr(states[l]), _hip.ptr(active), b, i_d, h,
                                          _hip.ptr(Wg), _hip.ptr(bg), _hip.ptr(Wc), _hip.ptr(bc), self.hidden_act,
                                          _hip.ptr(r), _hip.ptr(u), _hip.ptr(c), _hip.ptr(hn), st))
            saved.append((src, idx, r, u, c))
            new_states.append(hn)
            src, idx = hn, None
        return new_states, saved

    def train_step(self, x_index, y_index, states):
        """`sess.run([update_opt, final_state])` of the reference (:231): x_index int32 [b], y_index int32
        [b + n_sample] (device), states: list of [b, h_l].  Returns the new states; ``self.loss`` holds the
        mean bpr_max / top1_max loss of the step.  The returned tensors are reused by the next call with the
        other parity, so a caller may keep them for exactly one more step (the fit loop does)."""
        L, st = _hip.lib(), _hip.stream()
        b, n_y, hn = x_index.numel(), y_index.numel(), self.hids[-1]
        self._parity = 1 - getattr(self, "_parity", 0)
        new_states, saved = self.forward(x_index, states, save=True, tag=f"t{self._parity}")
        out = new_states[-1]
        dlog, dout = self._buf("dlogits", (b, n_y)), self._buf("dout", (b, hn))
        g = self.opt                     # (skr_session_loss_grads clears the loss word itself)
        # inside a block the hot step knows the rows by the block's id list: no touch bytes
        p_touch, p_base = (None, None) if self._blk_left > 0 else (_hip.ptr(g.touch), _hip.ptr(g.grad))
        # logits, row losses, then dL/dout together with the output-side gradients (both only read dlogits): three launches
        _hip.check(L.skr_session_loss_grads(_hip.ptr(out), b, hn, _hip.ptr(self.E_out), _hip.ptr(self.b_out), _hip.ptr(y_index),
                                            n_y, self.final_act, self.loss_kind, self.bpr_reg, _hip.ptr(dlog), _hip.ptr(dout),
                                            _hip.ptr(self.loss), 0, b, self.reg, _hip.ptr(self.gE_out), _hip.ptr(self.gb_out),
                                            p_touch, p_base, st))
        dh = dout
        for l in range(len(self.cells) - 1, -1, -1):
            (Wg, bg, Wc, bc), (gWg, gbg, gWc, gbc) = self.cells[l], self.gcells[l]
            src, idx, r, u, c = saved[l]
            i_d, h = self.dims_in[l], self.hids[l]
            dx, work = self._buf(f"dx{l}", (b, i_d)), self._buf(f"work{l}", (3 * b * h,))
            if l == 0:      # the input-embedding gradient (dx scattered to the batch's input rows) rides in the weights launch
                _hip.check(L.skr_gru_cell_bwd_scatter(_hip.ptr(src), _hip.ptr(idx), _hip.ptr(states[l]), b, i_d, h, _hip.ptr(Wg),
                                                      _hip.ptr(Wc), self.hidden_act, _hip.ptr(r), _hip.ptr(u), _hip.ptr(c),
                                                      _hip.ptr(dh), _hip.ptr(gWg), _hip.ptr(gbg), _hip.ptr(gWc), _hip.ptr(gbc),
                                                      _hip.ptr(dx), _hip.ptr(work), self.reg, _hip.ptr(self.gE_in), p_touch, p_base,
                                                      st))
            else:
                _hip.check(L.skr_gru_cell_bwd(_hip.ptr(src), _hip.ptr(idx), _hip.ptr(states[l]), b, i_d, h, _hip.ptr(Wg),
                                              _hip.ptr(Wc), self.hidden_act, _hip.ptr(r), _hip.ptr(u), _hip.ptr(c), _hip.ptr(dh),
                                              _hip.ptr(gWg), _hip.ptr(gbg), _hip.ptr(gWc), _hip.ptr(gbc), _hip.ptr(dx),
                                              _hip.ptr(work), st))
            dh = dx
        self._optimizer_step()
        return new_states

    def user_embeddings(self, d_rowptr, d_items_by_time, max_len):
        """top-layer state after each user's whole history (zero rows for users without one)"""
        n_users = d_rowptr.numel() - 1
        lens = d_rowptr[1:] - d_rowptr[:-1]
        states = self.zero_states(n_users)
        last = d_items_by_time.numel() - 1
        for t in range(int(max_len)):
            active = (lens > t).to(torch.uint8)
            x_index = d_items_by_time[torch.clamp(d_rowptr[:-1] + t, max=last)].contiguous()
            states, _ = self.forward(x_index, states, active=active, tag=f"e{t & 1}")
        return states[-1]


class ShardedSessionGRU(SessionGRU):
    """SessionGRU for one rank of a SESSION-SHARDED job (BASELINE configs[4]; the reference is single-process).

    The b sessions that advance together (GRU4RecPlus.py:210-247) are split into world contiguous groups of b / world
    slots; a rank runs the GRU stack, the logits and their gradients for its own slots only.  Every parameter is
    replicated -- both item tables, the output bias, the GRU kernels -- and kept identical: a step's targets (the batch's
    next items + the shared negatives) and inputs are the same lists on every rank (same data, same numpy stream), so the
    ranks exchange ONE compact block per step -- the rows of dE_out / db_out for the targets, the rows of dE_in for the
    inputs, the dense GRU gradients -- all-gather it and add the ranks' blocks in rank order (``skr_sum_blocks``), then
    run the same dense Adam.  The l2 term of the shared target rows is added by rank 0 only."""

    def __init__(self, ctx, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.ctx = ctx
        self.opt.touch[:] = 2                 # exchanged rows arrive without touch marks: always read every gradient
        self._cells_lo = int(self.gcells[0][0].data_ptr() - self.opt.grad.data_ptr()) // 4
        self._xbuf = None

    def slots(self, b):
        assert b % self.ctx.world == 0, "batch_size must be a multiple of the number of ranks"
        per = b // self.ctx.world
        return self.ctx.rank * per, (self.ctx.rank + 1) * per

    def train_step(self, x_index, y_index, states):
        """x_index int32 [b], y_index int32 [b + n_sample]: the GLOBAL step (identical on every rank); states: this rank's
        [b / world, h_l].  ``self.loss`` = the global mean loss afterwards."""
        import torch.distributed as dist
        L, st = _hip.lib(), _hip.stream()
        b, n_y, hn = x_index.numel(), y_index.numel(), self.hids[-1]
        lo, hi = self.slots(b)
        bl = hi - lo
        x_local = x_index[lo:hi].contiguous()
        self._parity = 1 - getattr(self, "_parity", 0)
        new_states, saved = self.forward(x_local, states, save=True, tag=f"t{self._parity}")
        out = new_states[-1]
        dlog, dout = self._buf("dlogits", (bl, n_y)), self._buf("dout", (bl, hn))
        g = self.opt                     # (skr_session_loss_grads clears the loss word itself)
        reg_y = self.reg if self.ctx.rank == 0 else 0.0
        _hip.check(L.skr_session_loss_grads(_hip.ptr(out), bl, hn, _hip.ptr(self.E_out), _hip.ptr(self.b_out), _hip.ptr(y_index),
                                            n_y, self.final_act, self.loss_kind, self.bpr_reg, _hip.ptr(dlog), _hip.ptr(dout),
                                            _hip.ptr(self.loss), lo, b, reg_y, _hip.ptr(self.gE_out), _hip.ptr(self.gb_out),
                                            _hip.ptr(g.touch), _hip.ptr(g.grad), st))
        dh = dout
        for l in range(len(self.cells) - 1, -1, -1):
            (Wg, bg, Wc, bc), (gWg, gbg, gWc, gbc) = self.cells[l], self.gcells[l]
            src, idx, r, u, c = saved[l]
            i_d, h = self.dims_in[l], self.hids[l]
            dx, work = self._buf(f"dx{l}", (bl, i_d)), self._buf(f"work{l}", (3 * bl * h,))
            if l == 0:      # (the input-embedding gradient rides in the weights launch, as in SessionGRU.train_step)
                _hip.check(L.skr_gru_cell_bwd_scatter(_hip.ptr(src), _hip.ptr(idx), _hip.ptr(states[l]), bl, i_d, h, _hip.ptr(Wg),
                                                      _hip.ptr(Wc), self.hidden_act, _hip.ptr(r), _hip.ptr(u), _hip.ptr(c),
                                                      _hip.ptr(dh), _hip.ptr(gWg), _hip.ptr(gbg), _hip.ptr(gWc), _hip.ptr(gbc),
                                                      _hip.ptr(dx), _hip.ptr(work), self.reg, _hip.ptr(self.gE_in),
                                                      _hip.ptr(g.touch), _hip.ptr(g.grad), st))
            else:
                _hip.check(L.skr_gru_cell_bwd(_hip.ptr(src), _hip.ptr(idx), _hip.ptr(states[l]), bl, i_d, h, _hip.ptr(Wg),
                                              _hip.ptr(Wc), self.hidden_act, _hip.ptr(r), _hip.ptr(u), _hip.ptr(c), _hip.ptr(dh),
                                              _hip.ptr(gWg), _hip.ptr(gbg), _hip.ptr(gWc), _hip.ptr(gbc), _hip.ptr(dx),
                                              _hip.ptr(work), st))
            dh = dx
        # ---- the step's one exchange: [dE_out rows of y | db_out of y | dE_in rows of x | GRU gradients | loss]
        world = self.ctx.world
        n_cells = g.grad.numel() - self._cells_lo
        sizes = (n_y * hn, n_y, b * self.in_dim, n_cells, 1)
        total = sum(sizes)
        if self._xbuf is None or self._xbuf.numel() != total:
            self._xbuf = torch.empty(total, dtype=torch.float32, device=self.device)
            self._xall = torch.empty((world, total), dtype=torch.float32, device=self.device)
        o0, o1, o2, o3 = sizes[0], sizes[0] + sizes[1], sizes[0] + sizes[1] + sizes[2], total - 1
        buf = self._xbuf
        _hip.check(L.skr_gather_rows(_hip.ptr(self.gE_out), _hip.ptr(y_index), n_y, hn, _hip.ptr(buf[:o0]), st))
        _hip.check(L.skr_gather_rows(_hip.ptr(self.gb_out), _hip.ptr(y_index), n_y, 1, _hip.ptr(buf[o0:o1]), st))
        _hip.check(L.skr_gather_rows(_hip.ptr(self.gE_in), _hip.ptr(x_index), b, self.in_dim, _hip.ptr(buf[o1:o2]), st))
        buf[o2:o3].copy_(g.grad[self._cells_lo:])
        buf[o3:].copy_(self.loss)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self._xall, buf)
        else:
            dist.all_gather([self._xall[r] for r in range(world)], buf)
        _hip.check(L.skr_sum_blocks(_hip.ptr(self._xall), world, total, _hip.ptr(buf), st))
        _hip.check(L.skr_scatter_rows(_hip.ptr(buf[:o0]), _hip.ptr(y_index), n_y, hn, _hip.ptr(self.gE_out), st))
        _hip.check(L.skr_scatter_rows(_hip.ptr(buf[o0:o1]), _hip.ptr(y_index), n_y, 1, _hip.ptr(self.gb_out), st))
        _hip.check(L.skr_scatter_rows(_hip.ptr(buf[o1:o2]), _hip.ptr(x_index), b, self.in_dim, _hip.ptr(self.gE_in), st))
        g.grad[self._cells_lo:].copy_(buf[o2:o3])
        self.loss.copy_(buf[o3:])
        self._optimizer_step()
        return new_states


class GRU4RecPlus(AbstractRecommender):
    def __init__(self, run_config: RunConfig, model_config: Dict):
        self.config = GRU4RecPlusConfig(**model_config)
        super().__init__(run_config, self.config)
        config: GRU4RecPlusConfig = self.config
        self.device = _hip.require_gpu()
        self.users_num, self.items_num = self.dataset.num_users, self.dataset.num_items
        self.user_pos_train = self.dataset.train_data.to_user_dict_by_time()
        self.data_ui, self.offset_idx = self._init_data()
        # for sampling negative items (:103-106)
        _, pop = np.unique(self.data_ui[:, 1], return_counts=True)
        pop = np.power(pop, config.sample_alpha)
        pop_cumsum = np.cumsum(pop)
        self.pop_cumsum = pop_cumsum / pop_cumsum[-1]
        # variables (:124-135) and the GRUCell kernels (glorot uniform, gate bias 1, candidate bias 0)
        l1, ln = config.layers[0], config.layers[-1]
        E_in = nn.init.trunc_normal_(torch.empty(self.items_num, l1), mean=0.0, std=0.01, a=-0.02, b=0.02)
        E_out = nn.init.trunc_normal_(torch.empty(self.items_num, ln), mean=0.0, std=0.01, a=-0.02, b=0.02)
        cells, i_d = [], l1
        for h in config.layers:
            cells.append((nn.init.xavier_uniform_(torch.empty(i_d + h, 2 * h)), torch.ones(2 * h),
                          nn.init.xavier_uniform_(torch.empty(i_d + h, h)), torch.zeros(h)))
            i_d = h
        from ..parallel import init_from_env
        self.dist = init_from_env()      # one process per GPU under torchrun: the b parallel sessions are split over the ranks
        net_args = (E_in, cells, E_out, torch.zeros(self.items_num), config.hidden_act, config.final_act, config.loss,
                    config.bpr_reg, config.reg, config.lr, self.device)
        self.net = ShardedSessionGRU(self.dist, *net_args) if self.dist.active else SessionGRU(*net_args)
        self._d_pop_cumsum = torch.from_numpy(np.ascontiguousarray(self.pop_cumsum, dtype=np.float64)).to(self.device)
        # histories by time as a CSR over ALL users for the inference sweep
        rowptr = np.zeros(self.users_num + 1, np.int64)
        for u, items in self.user_pos_train.items():
            rowptr[u + 1] = len(items)
        self._max_len = int(rowptr.max())
        np.cumsum(rowptr, out=rowptr)
        flat = np.concatenate([self.user_pos_train[u] for u in sorted(self.user_pos_train)]).astype(np.int32)
        self._d_rowptr = torch.from_numpy(rowptr).to(self.device)
        self._d_hist = torch.from_numpy(flat).to(self.device)
        self._d_items = torch.from_numpy(np.ascontiguousarray(self.data_ui[:, 1], dtype=np.int32)).to(self.device)
        self.step_losses = []

    def _init_data(self):
        data_ui = self.dataset.train_data.to_user_item_pairs_by_time()
        _, idx = np.unique(data_ui[:, 0], return_index=True)
        offset_idx = np.zeros(len(idx) + 1, dtype=np.int32)
        offset_idx[:-1] = idx
        offset_idx[-1] = len(data_ui)
        return data_ui, offset_idx

    def _sample_neg_items(self, size):
        """device int32 [size]: np.searchsorted(pop_cumsum, np.random.rand(size)) (GRU4RecPlus.py:198-200) -- the uniforms
        come from numpy's global generator like the reference's, the search runs on the device (skr_pop_sample)"""
        u = torch.from_numpy(np.random.rand(size)).to(self.device)
        out = torch.empty(size, dtype=torch.int32, device=self.device)
        _hip.check(_hip.lib().skr_pop_sample(_hip.ptr(self._d_pop_cumsum), self._d_pop_cumsum.numel(), _hip.ptr(u), 0, size,
                                             _hip.ptr(out), _hip.stream()))
        return out

    def _schedule(self):
        """The session-parallel schedule of one epoch (GRU4RecPlus.fit, :208-247) as a generator: per training step the b
        positions in the time-ordered pair list that the step reads (inputs at pos, targets at pos + 1) and the slots whose
        state is zeroed before it.  Host logic on the data only -- it can run any number of steps ahead of the device; the
        epoch's permutation is drawn from numpy's global generator when the first step is asked for, as in the reference."""
        offset_idx, b = self.offset_idx, self.config.batch_size
        user_idx = np.random.permutation(len(offset_idx) - 1)
        iters = np.arange(b, dtype=np.int32)
        maxiter = iters.max()
        start = offset_idx[user_idx[iters]].astype(np.int64)
        end = offset_idx[user_idx[iters] + 1].astype(np.int64)
        finished, reset = False, None
        while not finished:
            min_len = int((end - start).min())
            for i in range(min_len - 1):
                yield start + i, reset
                reset = None
            start = start + min_len - 1
            mask = np.arange(len(iters))[(end - start) <= 1]
            for idx in mask:
                maxiter += 1
                if maxiter >= len(offset_idx) - 1:
                    finished = True
                    break
                iters[idx] = maxiter
                start[idx] = offset_idx[user_idx[maxiter]]
                end[idx] = offset_idx[user_idx[maxiter] + 1]
            if len(mask):
                reset = mask if reset is None else np.union1d(reset, mask)

    @on_compute_stream
    def train_epoch(self):
        """The session-parallel loop of GRU4RecPlus.fit (:210-247).  The schedule is host logic (``_schedule``), so the steps
        are prepared SKR_ADAM_BLOCK (default 32) at a time: one upload of the positions, one gather of the inputs / targets,
        the negatives' uniforms drawn in the reference's order (np.random.rand(n_sample) per step == one draw of k * n_sample)
        and searched on the device, and the dense Adam blocked over those k steps (SessionGRU.begin_block)."""
        import itertools
        cfg, net, dev = self.config, self.net, self.device
        d_items = self._d_items
        b = cfg.batch_size
        lo, hi = net.slots(b) if self.dist.active else (0, b)     # this rank's share of the b parallel sessions
        state = net.zero_states(hi - lo)
        kblk = max(1, min(64, int(os.environ.get("SKR_ADAM_BLOCK", "32"))))
        sched = self._schedule()
        losses = []
        while True:
            chunk = list(itertools.islice(sched, kblk))
            if not chunk:
                break
            k = len(chunk)
            pos = torch.from_numpy(np.stack([c[0] for c in chunk])).to(dev)            # [k, b]
            xs, ys = d_items[pos], d_items[pos + 1]
            if cfg.n_sample:
                ys = torch.cat([ys, self._sample_neg_items(k * cfg.n_sample).view(k, cfg.n_sample)], dim=1)
            xs, ys = xs.contiguous(), ys.contiguous()
            resets = None
            if any(c[1] is not None for c in chunk):
                r = np.zeros((k, b), dtype=bool)
                for s_, c in enumerate(chunk):
                    if c[1] is not None:
                        r[s_, c[1]] = True
                resets = torch.from_numpy(r[:, lo:hi].copy()).to(dev)
            if kblk > 1:
                net.begin_block(xs, ys)
            for s_ in range(k):
                if chunk[s_][1] is not None:
                    state = [st_.masked_fill(resets[s_].unsqueeze(1), 0.0) for st_ in state]
                state = net.train_step(xs[s_], ys[s_], state)
                losses.append(net.loss.clone())
        net.end_blocks()
        self.step_losses = torch.cat(losses).cpu().numpy() if losses else np.zeros(0, np.float32)

    @on_compute_stream
    def fit(self):
        self.logger.info("metrics:".ljust(12) + f"\t{self.evaluator.metrics_str}")
        if len(self.offset_idx) - 1 < self.config.batch_size:
            raise ValueError("batch_size is larger than the number of training sessions")   # the reference's index error
        early_stopping = EarlyStopping(metric="NDCG@10", patience=self.config.early_stop)
        for epoch in range(self.config.epochs):
            self.train_epoch()
            cur_result = self.evaluate()
            self.logger.info(f"epoch {epoch}:".ljust(12) + f"\t{cur_result.values_str}")
            if early_stopping(cur_result):
                self.logger.info("early stop")
                break
        self.logger.info("best:".ljust(12) + f"\t{early_stopping.best_result.values_str}")
        return early_stopping.best_result

    def _get_user_embeddings(self, mine_only=False):
        """top-layer state after each user's whole history.  ``mine_only`` (one process per GPU): the sweep is SHARDED like the
        evaluation that reads it -- this rank advances the users u % world == rank only; the other rows stay zero"""
        if not (mine_only and self.dist.active):
            return self.net.user_embeddings(self._d_rowptr, self._d_hist, self._max_len)
        if getattr(self, "_mine", None) is None:
            mine = torch.arange(self.dist.rank, self.users_num, self.dist.world, device=self.device)
            lens = (self._d_rowptr[1:] - self._d_rowptr[:-1])[mine]
            rp = torch.zeros(mine.numel() + 1, dtype=torch.int64, device=self.device)
            rp[1:] = torch.cumsum(lens, 0)
            # the owned users' histories, packed: entry e of local row r is entry (e - rp[r]) of the user's global row
            row_of = torch.repeat_interleave(torch.arange(mine.numel(), device=self.device), lens)
            src = self._d_rowptr[:-1][mine][row_of] + (torch.arange(int(rp[-1]), device=self.device) - rp[row_of])
            hist = self._d_hist[src].contiguous() if src.numel() else torch.zeros(1, dtype=torch.int32, device=self.device)
            self._mine = (mine, rp, hist, int(lens.max()) if lens.numel() else 0)
        mine, rp, hist, max_len = self._mine
        full = torch.zeros((self.users_num, self.config.layers[-1]), dtype=torch.float32, device=self.device)
        full[mine] = self.net.user_embeddings(rp, hist, max_len)
        return full

    @on_compute_stream
    def evaluate(self, test_users=None):
        if self.dist.active:
            # every rank sweeps and ranks its own share of the users; only the fp64 metric sums are all-reduced
            from ..parallel import sharded_evaluate
            self.cur_user_embeddings = self._get_user_embeddings(mine_only=True)
            return sharded_evaluate(self.dist, self.evaluator, self, test_users, self.device)
        self.cur_user_embeddings = self._get_user_embeddings().clone()
        return self.evaluator.evaluate(self, test_users)

    def predict_factors(self):
        """fused evaluator: a strictly increasing final activation does not change the ranking"""
        if self.config.final_act == "relu" or self.config.layers[-1] != 64:
            return None
        return self.cur_user_embeddings, self.net.E_out, self.net.b_out

    def predict(self, users):
        ue = self.cur_user_embeddings[torch.as_tensor(np.asarray(users, dtype=np.int64)).to(self.device)]
        scores = torch.addmm(self.net.b_out.unsqueeze(0), ue, self.net.E_out.t())      # plain library GEMM
        if self.config.final_act == "relu":
            scores = torch.relu(scores)
        elif self.config.final_act == "leaky_relu":
            scores = torch.maximum(scores, scores * 0.2)
        return scores.cpu().numpy().astype(np.float32)
