"""One process per GPU: user-sharded execution of the hot path.

The reference has no distributed code at all; this module is new design.  Users are hash-sharded
(``u % world``), every rank owns its users' embedding rows, their rows of the normalised adjacency and
their interactions; the item table is replicated and kept identical on every rank by summing its
partial results over RCCL (``torch.distributed`` backend "nccl" on ROCm; "gloo" for rehearsals).

* ``ShardedBPRMF`` -- the BPRMF step: local fused gather/score/scatter on the owned users' interactions
  of the global batch, ONE all-reduce of the [V | b] gradient slice (26 MB at I = 100 k), the same dense
  Adam on every replica.
* ``ShardedLayerGCN`` -- the same partition for LayerGCN: the cosine re-weighting is row-local (users on
  their owner, items redundantly and identically on every rank), 2K + 2 all-reduces of the item block.
* ``ShardedLightGCN`` -- LightGCN's full-graph propagation as a 1-D row partition of the bipartite
  graph: per layer the user side is a local SpMM against the replicated item block, the item side is a
  local SpMM over the rank's user columns followed by ONE all-reduce of the [I, 64] partial result
  (25.6 MB at I = 100 k); the backward pass has the same shape (A is symmetric).  A step costs
  2K + 1 all-reduces of the item block and is numerically the single-GPU step (same global batch, loss
  averaged over the GLOBAL batch size), which ``tests/test_gpu_dist.py`` checks against the reference's
  recorded trajectory on two ranks.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch

from . import _hip
from .recommender.base import DenseAdam
from .recommender.LightGCN import DeviceCSR

__all__ = ["DistContext", "ShardedBPRMF", "ShardedLightGCN", "ShardedLayerGCN", "init_from_env", "sharded_evaluate",
           "unique_padded_rows"]


class DistContext(object):
    """rank / world bookkeeping + the two collectives the path needs"""

    def __init__(self, rank=0, world=1):
        self.rank, self.world = int(rank), int(world)

    @property
    def active(self):
        # SKR_DIST_FORCE_ACTIVE=1: a one-rank group still goes through every collective (rehearsal of the RCCL calls on a
        # box with one GPU: tests/test_gpu_dist.py::test_fit_on_a_single_rank_rccl_group)
        return self.world > 1 or os.environ.get("SKR_DIST_FORCE_ACTIVE") == "1"

    def owned_users(self, num_users):
        return np.arange(self.rank, num_users, self.world, dtype=np.int64)

    def all_reduce(self, t):
        if self.active:
            import torch.distributed as dist
            dist.all_reduce(t)
        return t

    def all_reduce_begin(self, t):
        """start summing ``t`` over the ranks on the collective's own stream (RCCL: after everything already queued on
        the current stream); kernels launched on the current stream until ``all_reduce_end`` run beside it"""
        if not self.active:
            return None
        import torch.distributed as dist
        return dist.all_reduce(t, async_op=True)

    @staticmethod
    def all_reduce_end(work):
        if work is not None:
            work.wait()          # the current stream waits for the collective; the host does not (RCCL)

    def all_gather_rows(self, out, inp):
        """out [world, ...] <- every rank's inp (same shape on every rank)"""
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, inp)
        else:
            dist.all_gather([out[r] for r in range(self.world)], inp)
        return out

    def barrier(self):
        if self.active:
            import torch.distributed as dist
            dist.barrier()


def init_from_env():
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or os.environ.get("SKR_DIST_FORCE_ACTIVE") == "1") and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        backend = os.environ.get("SKR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return DistContext(rank, world)


def _pipelined():
    """several ranks: the next item-side partial product is queued before the wait for the current exchange (default);
    SKR_DIST_PIPELINE=0 keeps the products of a layer behind that layer's exchange (for A / B runs on a real node)"""
    return os.environ.get("SKR_DIST_PIPELINE", "1") != "0"


def unique_padded_rows(ids):
    """[S, C] integer ids -> int32 [S, C]: every row's distinct non-negative ids in ascending order, then -1 (an empty
    slot of skr_pack_grad_rows) -- the layout skr_unpack_grad_rows_sorted needs"""
    big = torch.iinfo(torch.int32).max
    srt = torch.sort(torch.where(ids < 0, torch.full_like(ids, big), ids), dim=1).values
    dup = torch.zeros_like(srt, dtype=torch.bool)
    dup[:, 1:] = srt[:, 1:] == srt[:, :-1]
    srt = torch.sort(torch.where(dup, torch.full_like(srt, big), srt), dim=1).values
    return torch.where(srt == big, torch.full_like(srt, -1), srt).int().contiguous()


def _csr_from_device_coo(rows, cols, vals, n_rows, n_cols):
    """rectangular CSR in HBM from unique (row, col) pairs"""
    order = torch.argsort(rows * n_cols + cols)
    csr = DeviceCSR.__new__(DeviceCSR)
    csr.shape = (int(n_rows), int(n_cols))
    csr.nnz = int(rows.numel())
    csr.rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
    csr.rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_rows), 0)
    csr.col = cols[order].int().contiguous()
    csr.val = vals[order].float().contiguous()
    return csr


class ShardedLightGCN(object):
    """LightGCN step / propagation for one rank of a user-sharded job.

    adj        scipy sparse [(U+I), (U+I)], the reference's normalised adjacency (symmetric types only)
    user0/item0  full initial tables (numpy or CPU tensors), identical on every rank
    """

    def __init__(self, ctx, adj, user0, item0, n_layers, lr, reg, batch_size_cfg, device=None):
        self.ctx = ctx
        self.device = device if device is not None else _hip.require_gpu()
        user0 = torch.as_tensor(np.asarray(user0), dtype=torch.float32)
        item0 = torch.as_tensor(np.asarray(item0), dtype=torch.float32)
        self.num_users, self.num_items = user0.shape[0], item0.shape[0]
        assert user0.shape[1] == 64 and item0.shape[1] == 64, "the MI355X kernels are specialised for 64 dims"
        self.n_layers, self.reg, self.batch_size_cfg = int(n_layers), float(reg), int(batch_size_cfg)
        U, I = self.num_users, self.num_items
        adj = sp.csr_matrix(adj).astype(np.float32)
        block = adj[:U, U:]
        if abs(block - adj[U:, :U].T).max() > 1e-7 or abs(adj[:U, :U]).sum() != 0 or abs(adj[U:, U:]).sum() != 0:
            raise NotImplementedError("sharded propagation needs a symmetric bipartite adjacency ('pre' or 'plain')")
        self.mine = ctx.owned_users(U)
        self.n_local = len(self.mine)
        a_ui = sp.csr_matrix(block[self.mine])                     # [U_loc, I]
        self.a_ui = DeviceCSR(a_ui, self.device)
        self.a_iu = DeviceCSR(sp.csr_matrix(a_ui.T), self.device)    # [I, U_loc]
        nl = self.n_local
        self.ego = torch.cat([user0[self.mine], item0], dim=0).to(self.device).contiguous()   # [U_loc + I, 64]
        self.optimizer = DenseAdam(self.ego.view(-1), lr=lr)
        self._g_ego = self.optimizer.grad.view(nl + I, 64)
        z = lambda n: torch.zeros((n, 64), dtype=torch.float32, device=self.device)  # noqa: E731
        self.final = z(nl + I)
        self._xu, self._xi = [z(nl), z(nl)], [z(I), z(I)]
        self._g_final = z(nl + I)
        self._gu, self._gi = [z(nl), z(nl)], [z(I), z(I)]
        self.loss = torch.zeros(2, dtype=torch.float32, device=self.device)

    @classmethod
    def from_device_edges(cls, ctx, users, items, num_users, num_items, user0, item0, n_layers, lr, reg, batch_size_cfg):
        """Same engine, built entirely on the device from the train pairs (int tensors in HBM) with the
        'pre' normalisation D^-1/2 A D^-1/2 of LightGCN._create_adj_mat (LightGCN.py:160-163): no scipy pass
        over ~10^8 non-zeros.  ``user0`` holds only this rank's rows (owned_users order), ``item0`` all items."""
        self = cls.__new__(cls)
        dev = users.device
        self.ctx, self.device = ctx, dev
        self.num_users, self.num_items = int(num_users), int(num_items)
        self.n_layers, self.reg, self.batch_size_cfg = int(n_layers), float(reg), int(batch_size_cfg)
        u, i = users.long(), items.long()
        ones = torch.ones(u.numel(), dtype=torch.float32, device=dev)
        du = torch.zeros(num_users, device=dev).index_add_(0, u, ones)
        di = torch.zeros(num_items, device=dev).index_add_(0, i, ones)
        du = torch.where(du > 0, du.pow(-0.5), torch.zeros_like(du))
        di = torch.where(di > 0, di.pow(-0.5), torch.zeros_like(di))
        self.mine = ctx.owned_users(num_users)
        nl = self.n_local = len(self.mine)
        sel = (u % ctx.world) == ctx.rank
        ul, il = torch.div(u[sel], ctx.world, rounding_mode="floor"), i[sel]
        vals = du[u[sel]] * di[il]
        self.a_ui = _csr_from_device_coo(ul, il, vals, nl, num_items)
        self.a_iu = _csr_from_device_coo(il, ul, vals, num_items, nl)
        self.ego = torch.cat([user0.to(dev), item0.to(dev)], dim=0).contiguous()
        assert self.ego.shape == (nl + num_items, 64)
        self.optimizer = DenseAdam(self.ego.view(-1), lr=lr)
        self._g_ego = self.optimizer.grad.view(nl + num_items, 64)
        z = lambda n: torch.zeros((n, 64), dtype=torch.float32, device=dev)  # noqa: E731
        self.final = z(nl + num_items)
        self._xu, self._xi = [z(nl), z(nl)], [z(num_items), z(num_items)]
        self._g_final = z(nl + num_items)
        self._gu, self._gi = [z(nl), z(nl)], [z(num_items), z(num_items)]
        self.loss = torch.zeros(2, dtype=torch.float32, device=dev)
        return self

    # ---- helpers ---------------------------------------------------------------------------------
    def _axpy(self, a, x, y):
        _hip.check(_hip.lib().skr_axpy(float(a), _hip.ptr(x), _hip.ptr(y), x.numel(), _hip.stream()))

    @property
    def user_rows(self):
        return self.ego[:self.n_local]

    @property
    def item_rows(self):
        return self.ego[self.n_local:]

    # ---- forward ---------------------------------------------------------------------------------
    def propagate(self, last_rows=None):
        """final = mean(E0, A E0, ..., A^K E0) for the local user rows and the replicated item rows.
        ``last_rows`` = (uint8 [n_local], uint8 [I]) during training: the rows of ``final`` that will be read; the last
        layer's products are computed for those rows only and ``final`` is valid on those rows only."""
        K, nl = self.n_layers, self.n_local
        scale = 1.0 / (K + 1)
        active = self.ctx.active
        fu, fi = self.final[:nl], self.final[nl:]
        xu, xi = self.ego[:nl], self.ego[nl:]
        if active:      # the item half of the mean starts from its E0 term; the user half gets it in the first product's epilogue
            _hip.check(_hip.lib().skr_scale_copy(scale, _hip.ptr(xi), _hip.ptr(fi), xi.numel(), _hip.stream()))
        pipelined = _pipelined()
        ni = mi = None
        for k in range(K):
            nu = self._xu[k & 1]
            mu = last_rows[0] if (last_rows is not None and k == K - 1) else None
            au, ai = last_rows if last_rows is not None else (None, None)    # the mean is only needed on the rows that are read
            base_u, base_i = (self.ego[:nl], self.ego[nl:]) if k == 0 else (None, None)
            if not active:
                # one rank: nothing is exchanged, so the layer mean (with its E0 term) rides in both products' row epilogues
                ni, mi = self._xi[k & 1], (last_rows[1] if (last_rows is not None and k == K - 1) else None)
                self.a_iu.spmm(xu, ni, accum=fi, accum_scale=scale, accum_base=base_i, row_mask=mi, accum_mask=ai)
                self.a_ui.spmm(xi, nu, accum=fu, accum_scale=scale, accum_base=base_u, row_mask=mu, accum_mask=au)
                xu, xi = nu, ni
                continue
            # several ranks, software-pipelined (SKR_DIST_PIPELINE=0: layer by layer): layer k's exchange runs beside the
            # user-side product of layer k AND the item-side partial product of layer k + 1 -- that one only needs the user
            # rows the user-side product has just written, not the summed item rows -- so a layer's critical path is
            # max(exchange, both products) instead of item product + max(exchange, user product).  Same launches on the
            # same operands, hence the same bits.
            if k == 0 or not pipelined:
                ni, mi = self._item_partial(k, xu, last_rows)
            compact = self._batch_item_ids if mi is not None else None
            # In the masked last layer only the GLOBAL batch's item rows carry anything: they travel as a compact
            # [2 * batch, 64] block instead of the [I, 64] one
            if compact is not None:
                work = self._rows_exchange_begin(ni, compact)
            else:
                work = self.ctx.all_reduce_begin(ni)
            self.a_ui.spmm(xi, nu, accum=fu, accum_scale=scale, accum_base=base_u, row_mask=mu, accum_mask=au)   # local users <- replicated items
            ahead = self._item_partial(k + 1, nu, last_rows) if (pipelined and k + 1 < K) else None
            if compact is not None:
                self._rows_exchange_end(work, compact, ni)
            else:
                self.ctx.all_reduce_end(work)
            self._axpy(scale, ni, fi)
            xu, xi = nu, ni
            if ahead is not None:
                ni, mi = ahead
        self._final_whole = last_rows is None
        return self.final

    def _item_partial(self, k, xu, last_rows):
        """this rank's part of layer k's item rows <- its users' rows ``xu`` (the summands of the layer's exchange); in the
        masked last layer only the batch's rows.  Returns (buffer, row mask)."""
        ni = self._xi[k & 1]
        mi = last_rows[1] if (last_rows is not None and k == self.n_layers - 1) else None
        if mi is not None:
            ni.zero_()                                                    # rows that are skipped must not carry old sums
        self.a_iu.spmm(xu, ni, row_mask=mi)
        return ni, mi

    def _batch_rows(self, users, pos, neg, grad_rows=None):
        """(uint8 [n_local], uint8 [I]) for a GLOBAL batch (global user ids): the rows it touches -- this rank's users,
        EVERY rank's items (the item rows are replicated and summed over the ranks, so each rank needs the same set).
        Also keeps the batch's item ids (pos ++ neg, identical on every rank) for the compact exchanges."""
        L, st, world, rank = _hip.lib(), _hip.stream(), self.ctx.world, self.ctx.rank
        if getattr(self, "_mask_u", None) is None:
            self._mask_u = torch.zeros(self.n_local, dtype=torch.uint8, device=self.device)
            self._mask_i = torch.zeros(self.num_items, dtype=torch.uint8, device=self.device)
            if grad_rows is not None:
                grad_rows.zero_()
        elif grad_rows is not None:
            # the buffer the BPR kernel scatters dL/d(output rows) into is zero outside the PREVIOUS batch's rows: those rows
            # are cleared, and their marks with them, instead of filling the whole [n_local + I, 64] buffer
            nl = self.n_local
            _hip.check(L.skr_clear_marked_rows(_hip.ptr(self._mask_u), nl, 1, _hip.ptr(grad_rows[:nl]), 64, st))
            _hip.check(L.skr_clear_marked_rows(_hip.ptr(self._mask_i), self.num_items, 1, _hip.ptr(grad_rows[nl:]), 64, st))
        else:
            self._mask_u.zero_()
            self._mask_i.zero_()
        if world > 1:   # local rows of the users this rank owns; the others become -1 (skipped)
            ul = torch.where((users % world) == rank, torch.div(users, world, rounding_mode="floor"), torch.full_like(users, -1))
        else:
            ul = users
        _hip.check(L.skr_mark_ids(_hip.ptr(ul.contiguous()), ul.numel(), 0, _hip.ptr(self._mask_u), st))
        self._batch_item_ids = torch.cat([pos, neg]).contiguous()
        _hip.check(L.skr_mark_ids(_hip.ptr(self._batch_item_ids), self._batch_item_ids.numel(), 0, _hip.ptr(self._mask_i), st))
        return self._mask_u, self._mask_i

    def _rows_exchange_begin(self, block, ids):
        """Sum the rows `ids` of a replicated [I, 64] block over the ranks, compactly: the rows are gathered into a
        [len(ids), 64] buffer, the ranks' buffers all-gathered (started here, on the collective's stream) ..."""
        import torch.distributed as dist
        n, world = ids.numel(), self.ctx.world
        if getattr(self, "_xrows", None) is None or self._xrows.shape[0] != n:
            self._xrows = torch.empty((n, 64), dtype=torch.float32, device=self.device)
            self._xall = torch.empty((world, n, 64), dtype=torch.float32, device=self.device)
        _hip.check(_hip.lib().skr_gather_rows(_hip.ptr(block), _hip.ptr(ids), n, 64, _hip.ptr(self._xrows), _hip.stream()))
        if dist.get_backend() == "nccl":
            return dist.all_gather_into_tensor(self._xall, self._xrows, async_op=True)
        return dist.all_gather([self._xall[r] for r in range(world)], self._xrows, async_op=True)

    def _rows_exchange_end(self, work, ids, block):
        """... and added up in RANK ORDER by every rank itself (skr_sum_blocks): the replicas end with identical bits
        whatever order the collective library reduces in; the sums go back into the block's rows."""
        work.wait()
        L, st, n = _hip.lib(), _hip.stream(), ids.numel()
        _hip.check(L.skr_sum_blocks(_hip.ptr(self._xall), self.ctx.world, n * 64, _hip.ptr(self._xrows), st))
        _hip.check(L.skr_scatter_rows(_hip.ptr(self._xrows), _hip.ptr(ids), n, 64, _hip.ptr(block), st))

    def whole_final(self):
        """``final`` for readers of ALL its rows (evaluation, tests): after a training step only the batch's rows are valid
        (the last layer is computed for those only, and on several ranks the other item rows may even differ between
        the replicas); an unmasked ``propagate()`` makes it whole again"""
        if not getattr(self, "_final_whole", False):
            raise ValueError("ShardedLightGCN.final holds only a batch's rows after train_step(): call propagate() first")
        return self.final

    # ---- one training step on a GLOBAL batch -----------------------------------------------------------
    def train_step(self, users, pos, neg):
        """users/pos/neg: int32 device tensors of the whole global batch (identical on every rank);
        each rank keeps the interactions of the users it owns.  Returns nothing; ``self.loss`` holds the
        global (bpr mean, l2) of this step."""
        K, nl, world, rank = self.n_layers, self.n_local, self.ctx.world, self.ctx.rank
        n_global = users.numel()
        users, pos, neg = users.contiguous(), pos.contiguous(), neg.contiguous()
        # skipped: rows of the last forward layer that no rank's batch reads, and -- in the first backward hop -- the
        # entries that would multiply rows of dL/dfinal that are zero (everything outside the batch).  SKR_LIGHTGCN_DENSE=1
        # computes everything.
        gF, gE = self._g_final, self._g_ego
        if os.environ.get("SKR_LIGHTGCN_DENSE") == "1":
            masks = None
            gF.zero_()
            self._mask_u = None
        else:
            masks = self._batch_rows(users, pos, neg, grad_rows=gF)
        self.propagate(last_rows=masks)
        self.loss.zero_()
        # the whole GLOBAL batch goes to the kernel, which keeps the triples of the users this rank owns: no selection on
        # the host, no count to read back
        _hip.check(_hip.lib().skr_bpr_step_sharded(
            _hip.ptr(self.final[:nl]), _hip.ptr(self.final[nl:]), None, _hip.ptr(self.ego[:nl]), _hip.ptr(self.ego[nl:]),
            _hip.ptr(users), _hip.ptr(pos), _hip.ptr(neg), n_global, 1.0 / n_global, self.reg, 1.0 / self.batch_size_cfg,
            _hip.ptr(gF[:nl]), _hip.ptr(gF[nl:]), None, _hip.ptr(gE[:nl]), _hip.ptr(gE[nl:]), _hip.ptr(self.loss),
            None, None, world, rank, 1.0 / (K + 1), _hip.stream()))
        self.ctx.all_reduce(self.loss)
        # gF holds H = dL/dfinal / (K+1) (the kernel applied the factor); the item half is a partial sum over ranks
        hu, hi = gF[:nl], gF[nl:]
        active = self.ctx.active
        if masks is not None and active:      # hi is zero outside the global batch's item rows: compact exchange
            self._rows_exchange_end(self._rows_exchange_begin(hi, self._batch_item_ids), self._batch_item_ids, hi)
        else:
            self.ctx.all_reduce(hi)
        gu, gi = hu, hi
        gEu, gEi = gE[:nl], gE[nl:]
        hm_u, hm_i = masks if masks is not None else (None, None)      # H is zero outside the batch's rows
        scatter_ok = os.environ.get("SKR_FIRST_HOP_SCATTER", "1") != "0"

        def item_hop(k, gu_k):
            """several ranks: this rank's part of hop k's item rows (the summands of the hop's exchange)"""
            ni = self._gi[k & 1]
            cu = masks[0] if (masks is not None and k == 0) else None
            if cu is not None and k != K - 1 and scatter_ok:
                ni.zero_()
                self.a_ui.scatter_marked_rows(cu, gu_k, ni)
            else:
                self.a_iu.spmm(gu_k, ni, col_mask=cu)
            if k == K - 1:
                self._axpy(1.0, gEi, ni)           # this rank's regulariser part of the item gradient
            return ni
        pipelined = _pipelined()
        ni = None
        for k in range(K):
            last = (k == K - 1)
            nu = self._gu[k & 1]
            cu, ci = masks if (masks is not None and k == 0) else (None, None)
            # The item side of the FIRST hop multiplies dL/dE-bar's user rows, which are zero outside the batch's <= batch users:
            # instead of every item row scanning its users for marked ones, the batch users' own rows of the user-side block
            # (the same non-zeros) are walked and scattered -- ~50 k entries instead of 48 M looked at
            if not active:
                # one rank: g_{k+1} = A g_k + H in both epilogues; the last hop adds into the ego gradient (which holds the
                # regulariser's part) directly
                ni = self._gi[k & 1]
                if cu is not None and not last and scatter_ok:
                    ni.copy_(hi)
                    self.a_ui.scatter_marked_rows(cu, gu, ni)
                else:
                    self.a_iu.spmm(gu, ni, addend=hi, accum=gEi if last else None, accum_scale=1.0, col_mask=cu, addend_mask=hm_i)
                self.a_ui.spmm(gi, nu, addend=hu, accum=gEu if last else None, accum_scale=1.0, col_mask=ci, addend_mask=hm_u)
                gu, gi = nu, ni
                continue
            # several ranks, software-pipelined as the forward pass: hop k's exchange runs beside its user-side product and
            # the item-side partial product of hop k + 1 (which needs only the user rows just written)
            if k == 0 or not pipelined:
                ni = item_hop(k, gu)
            work = self.ctx.all_reduce_begin(ni)   # summed over the ranks beside the user-side product of the same hop
            self.a_ui.spmm(gi, nu, addend=hu, accum=gEu if last else None, accum_scale=1.0, col_mask=ci, addend_mask=hm_u)
            ahead = item_hop(k + 1, nu) if (pipelined and not last) else None
            self.ctx.all_reduce_end(work)
            self._axpy(1.0, hi, ni)
            gu, gi = nu, ni
            if ahead is not None:
                ni = ahead
        if active:
            gEi.copy_(gi)                          # identical on every rank -> identical Adam update
        self.optimizer.step()

    def gather_user_table(self):
        """full [U, 64] user table on every rank (tests / checkpoints)"""
        full = torch.zeros((self.num_users, 64), dtype=torch.float32, device=self.device)
        full[torch.from_numpy(self.mine).to(self.device)] = self.ego[:self.n_local]
        self.ctx.all_reduce(full)
        return full


class ShardedLayerGCN(object):
    """LayerGCN step / propagation for one rank of a user-sharded job (reference maths: LayerGCN.py:199-252).

    edge_u / edge_i   int64 device tensors, the FULL (possibly pruned) train edge list, identical on every rank
    user0 / item0     full initial tables, identical on every rank
    Rows are laid out [local users ; all items].  Item rows of every intermediate are replicated: each rank
    computes them from all-reduced inputs, so they stay bit-identical without being exchanged again.
    """

    def __init__(self, ctx, edge_u, edge_i, num_users, num_items, user0, item0, n_layers, lr, reg, device=None):
        self.ctx = ctx
        self.device = dev = device if device is not None else _hip.require_gpu()
        user0 = torch.as_tensor(np.asarray(user0), dtype=torch.float32)
        item0 = torch.as_tensor(np.asarray(item0), dtype=torch.float32)
        self.num_users, self.num_items = int(num_users), int(num_items)
        assert user0.shape == (self.num_users, 64) and item0.shape == (self.num_items, 64)
        self.n_layers, self.reg = int(n_layers), float(reg)
        self.mine = ctx.owned_users(self.num_users)
        nl = self.n_local = len(self.mine)
        n = nl + self.num_items
        self.ego = torch.cat([user0[self.mine], item0], dim=0).to(dev).contiguous()
        self.optimizer = DenseAdam(self.ego.view(-1), lr=lr)
        self._g_ego = self.optimizer.grad.view(n, 64)
        z = lambda: torch.zeros((n, 64), dtype=torch.float32, device=dev)  # noqa: E731
        K = self.n_layers
        self.out = z()
        self._y = [z() for _ in range(K)]
        self._w = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(K)]
        self._z = [z(), z()]
        self._g_out = z()
        self._t = [z(), z()]
        self.loss = torch.zeros(2, dtype=torch.float32, device=dev)
        self.full_blocks = self._blocks(edge_u, edge_i)
        self.train_blocks = self.full_blocks

    def _blocks(self, edge_u, edge_i):
        """local row blocks of get_norm_adj_mat / _normalize_adj_m (LayerGCN.py:154-163,173-197):
        value 1/sqrt((deg_u + 1e-7)(deg_i + 1e-7)), float64 then rounded to float32"""
        dev, world, rank = self.device, self.ctx.world, self.ctx.rank
        u, i = edge_u.to(dev).long(), edge_i.to(dev).long()
        # (degrees as exact integer counts: a float64 index_add_ of ones gives the same numbers through 10^8 double atomics,
        #  16 s at 48 M edges)
        du = (torch.bincount(u, minlength=self.num_users).double() + 1e-7).pow(-0.5)
        di = (torch.bincount(i, minlength=self.num_items).double() + 1e-7).pow(-0.5)
        sel = (u % world) == rank
        ul, il = torch.div(u[sel], world, rounding_mode="floor"), i[sel]
        vals = (du[u[sel]] * di[il]).float()
        return (_csr_from_device_coo(ul, il, vals, self.n_local, self.num_items),
                _csr_from_device_coo(il, ul, vals, self.num_items, self.n_local))

    def set_train_edges(self, edge_u=None, edge_i=None):
        """the pruned graph of this epoch (LayerGCN.py:133-152); None restores the full graph"""
        self.train_blocks = self.full_blocks if edge_u is None else self._blocks(edge_u, edge_i)

    @property
    def item_rows(self):
        return self.ego[self.n_local:]

    def _axpy(self, a, x, y):
        _hip.check(_hip.lib().skr_axpy(float(a), _hip.ptr(x), _hip.ptr(y), x.numel(), _hip.stream()))

    def propagate(self, train=False, last_rows=None):
        """out = sum_k  w_k * (A X_k),  w_k = cos(A X_k, E0) row-wise,  X_{k+1} = w_k * (A X_k).
        ``last_rows`` = (uint8 [n_local], uint8 [I]) during training: the rows of ``out`` that will be read; the LAST layer's
        products are computed (and, item side, exchanged -- compactly) for those rows only.  The rows left out keep older,
        finite values in y / w / out; nothing reads ``out`` there and their dL/d out is zero."""
        L, st, nl = _hip.lib(), _hip.stream(), self.n_local
        a_ui, a_iu = self.train_blocks if train else self.full_blocks
        active = self.ctx.active
        x = self.ego
        K = self.n_layers
        pipelined = _pipelined()
        for k in range(K):
            y, zk, wk = self._y[k], self._z[k & 1], self._w[k]
            mu, mi = last_rows if (last_rows is not None and k == K - 1) else (None, None)
            first = (k == 0)       # the first layer initialises `out` (no fill of the buffer)
            au, ai = last_rows if last_rows is not None else (None, None)    # `out` is only needed on the rows that are read
            if not active:
                # one rank: the refinement (w = cos(A x, E0), z = w * A x, out += z) rides in both products' row epilogues
                a_iu.spmm(x[:nl], y[nl:], row_mask=mi, accum=self.out[nl:], accum_init=first,
                          refine_fwd=(self.ego[nl:], wk[nl:], zk[nl:]), accum_mask=ai)
                a_ui.spmm(x[nl:], y[:nl], row_mask=mu, accum=self.out[:nl], accum_init=first,
                          refine_fwd=(self.ego[:nl], wk[:nl], zk[:nl]), accum_mask=au)
                x = zk
                continue
            # several ranks, software-pipelined (as ShardedLightGCN.propagate): layer k's exchange runs beside the user-side
            # product of layer k (whose refinement rides in its row epilogue) and the item-side partial product of layer
            # k + 1, which reads only the user rows of z_k that epilogue has just written
            if k == 0 or not pipelined:
                a_iu.spmm(x[:nl], y[nl:], row_mask=mi)            # partial items <- local users
            compact = self._batch_item_ids if mi is not None else None
            if compact is not None:
                work = ShardedLightGCN._rows_exchange_begin(self, y[nl:], compact)
            else:
                work = self.ctx.all_reduce_begin(y[nl:])
            a_ui.spmm(x[nl:], y[:nl], row_mask=mu, accum=self.out[:nl], accum_init=first,
                      refine_fwd=(self.ego[:nl], wk[:nl], zk[:nl]), accum_mask=au)
            if pipelined and k + 1 < K:
                mi_next = last_rows[1] if (last_rows is not None and k + 1 == K - 1) else None
                a_iu.spmm(zk[:nl], self._y[k + 1][nl:], row_mask=mi_next)
            if compact is not None:
                ShardedLightGCN._rows_exchange_end(self, work, compact, y[nl:])
            else:
                self.ctx.all_reduce_end(work)
            # the item rows' refinement needs the summed rows: a launch of its own over the item block
            if first:
                self.out[nl:].zero_()
            _hip.check(L.skr_layer_refine_fwd(_hip.ptr(y[nl:]), _hip.ptr(self.ego[nl:]), self.num_items, 64, _hip.ptr(zk[nl:]),
                                              _hip.ptr(wk[nl:]), _hip.ptr(self.out[nl:]), st))
            x = zk
        self._out_whole = last_rows is None
        return self.out

    _batch_rows = ShardedLightGCN._batch_rows

    def whole_out(self):
        """``out`` for readers of ALL its rows: valid after ``propagate()`` without ``last_rows`` only (see
        ShardedLightGCN.whole_final)"""
        if not getattr(self, "_out_whole", False):
            raise ValueError("ShardedLayerGCN.out holds only a batch's rows after train_step(): call propagate() first")
        return self.out

    def train_step(self, users, pos, neg):
        """global batch in (global user ids, identical on every rank); ``self.loss`` = global (bpr sum, l2) afterwards"""
        L, st = _hip.lib(), _hip.stream()
        nl, world, rank, K = self.n_local, self.ctx.world, self.ctx.rank, self.n_layers
        n = self.ego.shape[0]
        users, pos, neg = users.contiguous(), pos.contiguous(), neg.contiguous()
        a_ui, a_iu = self.train_blocks
        # not computed: rows of the last layer's product the batch does not read and, in the first backward hop, the
        # products with rows of dY_K that are zero (as in the one-GPU engine, recommender/LayerGCN.py train_step)
        gO, gE = self._g_out, self._g_ego
        if os.environ.get("SKR_LIGHTGCN_DENSE") == "1":
            masks = None
            gO.zero_()
            self._mask_u = None
        else:
            masks = self._batch_rows(users, pos, neg, grad_rows=gO)
        self.propagate(train=True, last_rows=masks)
        self.loss.zero_()
        # the whole GLOBAL batch goes to the kernel, which keeps the triples of the users this rank owns
        _hip.check(L.skr_bpr_step_sharded(
            _hip.ptr(self.out[:nl]), _hip.ptr(self.out[nl:]), None, _hip.ptr(self.ego[:nl]), _hip.ptr(self.ego[nl:]),
            _hip.ptr(users), _hip.ptr(pos), _hip.ptr(neg), users.numel(), 1.0, self.reg, 1.0,
            _hip.ptr(gO[:nl]), _hip.ptr(gO[nl:]), None, _hip.ptr(gE[:nl]), _hip.ptr(gE[nl:]), _hip.ptr(self.loss),
            None, None, world, rank, 1.0, st))
        self.ctx.all_reduce(self.loss)
        active = self.ctx.active
        if masks is not None and active:
            # both item-side blocks are zero outside the global batch's item rows: compact exchanges
            ids = self._batch_item_ids
            ShardedLightGCN._rows_exchange_end(self, ShardedLightGCN._rows_exchange_begin(self, gO[nl:], ids), ids, gO[nl:])
            ShardedLightGCN._rows_exchange_end(self, ShardedLightGCN._rows_exchange_begin(self, gE[nl:], ids), ids, gE[nl:])
        else:
            self.ctx.all_reduce(gO[nl:])          # dL/d out, item rows: now the full value everywhere
            self.ctx.all_reduce(gE[nl:])          # the regulariser's part of the item gradient
        # backward: dZ_K = gO ; dY_k, dE0 += refine_bwd(dZ_k) ; dZ_{k-1} = gO + A dY_k ; dE0 += A dY_1.  The top refinement
        # only visits the batch's rows (dZ_K is zero elsewhere; the plan's product skips the other columns of dY_K, for
        # the plan-free kernel the skipped rows are written as zeros); every further one rides in the row epilogue of
        # the hop that produces its dZ -- user rows always, item rows when nothing has to be summed over ranks first.
        dy, nxt = self._t
        mk_u, mk_i = masks if masks is not None else (None, None)
        zs = 0 if (a_ui.uses_plan() and a_iu.uses_plan()) else 1
        yK, wK = self._y[K - 1], self._w[K - 1]
        _hip.check(L.skr_layer_refine_bwd_masked(_hip.ptr(yK[:nl]), _hip.ptr(self.ego[:nl]), _hip.ptr(wK[:nl]), _hip.ptr(gO[:nl]), nl, 64,
                                                 _hip.ptr(dy[:nl]), _hip.ptr(gE[:nl]), _hip.ptr(mk_u), zs, st))
        _hip.check(L.skr_layer_refine_bwd_masked(_hip.ptr(yK[nl:]), _hip.ptr(self.ego[nl:]), _hip.ptr(wK[nl:]), _hip.ptr(gO[nl:]),
                                                 self.num_items, 64, _hip.ptr(dy[nl:]), _hip.ptr(gE[nl:]), _hip.ptr(mk_i), zs, st))
        scatter_ok = os.environ.get("SKR_FIRST_HOP_SCATTER", "1") != "0"
        pipelined = _pipelined()

        def item_hop(k, dyu, slot):
            """several ranks: this rank's part of hop k's item rows <- its users' rows of dY (the summands of the hop's exchange)"""
            t = self._tmp_items(slot)
            cu = masks[0] if (masks is not None and k == K - 1) else None
            if cu is not None and k > 0 and scatter_ok:
                t.zero_()
                a_ui.scatter_marked_rows(cu, dyu, t)
            else:
                a_iu.spmm(dyu, t, col_mask=cu)
            return t
        tmp_i = None           # several ranks: the hop's item-side partial sums when they were queued ahead
        for k in range(K - 1, -1, -1):
            cu, ci = masks if (masks is not None and k == K - 1) else (None, None)
            if k > 0:
                yb, wb = self._y[k - 1], self._w[k - 1]
                rb_u = (self.ego[:nl], wb[:nl], yb[:nl], gE[:nl])
                rb_i = (self.ego[nl:], wb[nl:], yb[nl:], gE[nl:])
                # (the item side of the FIRST hop -- dY_K's user rows are zero outside the batch's users: the batch users' own rows
                #  are scattered instead of every item row scanning its users, as in ShardedLightGCN.train_step)
                transposed = cu is not None and scatter_ok
                if not active:
                    if transposed:
                        t1 = self._tmp_items()
                        t1.copy_(gO[nl:])
                        a_ui.scatter_marked_rows(cu, dy[:nl], t1)
                        _hip.check(L.skr_layer_refine_bwd(_hip.ptr(yb[nl:]), _hip.ptr(self.ego[nl:]), _hip.ptr(wb[nl:]), _hip.ptr(t1),
                                                          self.num_items, 64, _hip.ptr(nxt[nl:]), _hip.ptr(gE[nl:]), st))
                    else:
                        a_iu.spmm(dy[:nl], nxt[nl:], addend=gO[nl:], col_mask=cu, refine_bwd=rb_i, addend_mask=mk_i)
                    a_ui.spmm(dy[nl:], nxt[:nl], addend=gO[:nl], col_mask=ci, refine_bwd=rb_u, addend_mask=mk_u)
                else:
                    # several ranks, software-pipelined: the item side first, its exchange beside the user side AND beside the
                    # item-side partial product of the hop below, which reads only the user rows that product has just written
                    if tmp_i is None:
                        tmp_i = item_hop(k, dy[:nl], k & 1)
                    work = self.ctx.all_reduce_begin(tmp_i)
                    a_ui.spmm(dy[nl:], nxt[:nl], addend=gO[:nl], col_mask=ci, refine_bwd=rb_u, addend_mask=mk_u)
                    ahead = item_hop(k - 1, nxt[:nl], (k - 1) & 1) if pipelined else None
                    self.ctx.all_reduce_end(work)
                    self._axpy(1.0, gO[nl:], tmp_i)
                    _hip.check(L.skr_layer_refine_bwd(_hip.ptr(yb[nl:]), _hip.ptr(self.ego[nl:]), _hip.ptr(wb[nl:]), _hip.ptr(tmp_i),
                                                      self.num_items, 64, _hip.ptr(nxt[nl:]), _hip.ptr(gE[nl:]), st))
                    tmp_i = ahead
                dy, nxt = nxt, dy
            else:
                if not active:
                    a_iu.spmm(dy[:nl], None, accum=gE[nl:], accum_scale=1.0, col_mask=cu)
                    a_ui.spmm(dy[nl:], None, accum=gE[:nl], accum_scale=1.0, col_mask=ci)
                else:
                    if tmp_i is None:
                        tmp_i = item_hop(0, dy[:nl], 0)
                    work = self.ctx.all_reduce_begin(tmp_i)
                    a_ui.spmm(dy[nl:], None, accum=gE[:nl], accum_scale=1.0, col_mask=ci)
                    self.ctx.all_reduce_end(work)
                    self._axpy(1.0, tmp_i, gE[nl:])
        self.optimizer.step()

    def _tmp_items(self, slot=0):
        """[I, 64] scratch; two slots, so that a hop's partial sums can be queued while the previous hop's are being exchanged"""
        if getattr(self, "_tmp_i", None) is None:
            self._tmp_i = [None, None]
        if self._tmp_i[slot] is None:
            self._tmp_i[slot] = torch.zeros((self.num_items, 64), dtype=torch.float32, device=self.device)
        return self._tmp_i[slot]

    def gather_user_rows(self, local_rows):
        """[U, 64] on every rank from each rank's [U_local, 64] block"""
        full = torch.zeros((self.num_users, 64), dtype=torch.float32, device=self.device)
        full[torch.from_numpy(self.mine).to(self.device)] = local_rows
        self.ctx.all_reduce(full)
        return full

    def gather_user_table(self):
        return self.gather_user_rows(self.ego[:self.n_local])


def sharded_evaluate(ctx, evaluator, model, test_users, device):
    """every rank ranks its share (u % world) of the test users with the fused evaluator; the fp64 metric
    sums and the user count are all-reduced, so every rank returns the same report"""
    from .utils.py import MetricReport
    ev = evaluator
    users = list(ev.user_pos_test.keys()) if test_users is None else [u for u in test_users if u in ev.user_pos_test]
    mine = [u for u in users if u % ctx.world == ctx.rank]
    _, sums, n = ev.per_user_rows(model, mine)
    tot = torch.from_numpy(np.concatenate([sums, [float(n)]])).to(device)
    ctx.all_reduce(tot)
    tot = tot.cpu().numpy()
    final = (tot[:-1] / max(tot[-1], 1.0)).astype(np.float32)
    final = final.reshape(ev.metrics_num, ev.max_top)[:, ev.top_show - 1].reshape(-1)
    return MetricReport(ev.metrics_list, final)


class ShardedBPRMF(object):
    """BPRMF step for one rank: flat parameter buffer [U_local | V | b], item part replicated."""

    def __init__(self, ctx, user0, item0, bias0, lr, reg, device=None, exchange=None):
        self.ctx = ctx
        # how the item gradient is summed over the ranks: "dense" all-reduces the [I, 65] block, "sparse" all-gathers
        # the touched rows (at most 2 * largest local batch), "auto" picks the smaller message per step
        self.exchange = exchange or os.environ.get("SKR_EXCHANGE", "auto")
        if self.exchange not in ("auto", "dense", "sparse"):
            raise ValueError("exchange must be 'auto', 'dense' or 'sparse'")
        self.device = device if device is not None else _hip.require_gpu()
        user0 = torch.as_tensor(np.asarray(user0), dtype=torch.float32)
        item0 = torch.as_tensor(np.asarray(item0), dtype=torch.float32)
        bias0 = torch.as_tensor(np.asarray(bias0), dtype=torch.float32).reshape(-1)
        self.num_users, self.num_items = user0.shape[0], item0.shape[0]
        assert user0.shape[1] == 64 and item0.shape[1] == 64, "the MI355X kernels are specialised for 64 dims"
        self.reg = float(reg)
        self.mine = ctx.owned_users(self.num_users)
        nl, ni = len(self.mine), self.num_items
        self.n_local = nl
        self.flat = torch.cat([user0[self.mine].reshape(-1), item0.reshape(-1), bias0]).to(self.device).contiguous()
        self.user_rows = self.flat[:nl * 64].view(nl, 64)
        self.item_rows = self.flat[nl * 64:(nl + ni) * 64].view(ni, 64)
        self.item_bias = self.flat[(nl + ni) * 64:]
        # SKR_ADAM_BLOCK = k > 1: temporally blocked dense Adam through train_block() (no touch bytes then)
        self.adam_block = max(1, min(32, int(os.environ.get("SKR_ADAM_BLOCK", "32"))))
        self.optimizer = DenseAdam(self.flat, lr=lr, track_touch=self.adam_block <= 1)
        g = self.optimizer.grad
        self._gU, self._gV, self._gb = g[:nl * 64].view(nl, 64), g[nl * 64:(nl + ni) * 64].view(ni, 64), g[(nl + ni) * 64:]
        self._g_item = g[nl * 64:]                      # [V | b]: what the ranks exchange
        self._dense_marked = False
        self.loss = torch.zeros(2, dtype=torch.float32, device=self.device)

    def _exchange_item_grads(self, n_global, il, jl, ids=None):
        """sum the [V | b] gradient over the ranks; every replica ends with bit-identical values.  ``ids`` (int32 [1, 2 *
        n_global], ascending then -1): the rows to exchange, if the caller has them already (train_block: the GLOBAL batch's
        items, the same list on every rank -- rows a rank did not touch travel as zeros)"""
        world, ni = self.ctx.world, self.num_items
        opt = self.optimizer
        cap = 2 * int(n_global)          # slots per rank: a rank holds at most the whole global batch (no device read-back)
        sparse = self.exchange == "sparse" or (self.exchange == "auto" and world * cap * 66 < ni * 65)
        if not sparse:
            if opt.touch is not None and not self._dense_marked:
                opt.touch[self.n_local:] = 2            # summed gradients are dense: always read them
                self._dense_marked = True
            self.ctx.all_reduce(self._g_item)
            return
        if self._dense_marked:
            opt.touch[self.n_local:] = 0
            opt.grad[self.n_local * 64:].zero_()
            self._dense_marked = False
        if getattr(self, "_xbuf_cap", 0) != cap:     # exchange buffers: allocated once per batch size, not per step
            self._xbuf_cap = cap
            self._xbuf_ids = torch.empty((1, max(cap, 1)), dtype=torch.int32, device=self.device)
            self._xbuf_pack = torch.empty((cap, 66), dtype=torch.float32, device=self.device)
            self._xbuf_gathered = torch.empty((world, cap, 66), dtype=torch.float32, device=self.device)
        if ids is None:
            ids = self._xbuf_ids.fill_(-1)
            n = il.numel()
            ids[0, :n], ids[0, n:2 * n] = il, jl
            ids = unique_padded_rows(ids)
        pack, gathered = self._xbuf_pack, self._xbuf_gathered
        L, st = _hip.lib(), _hip.stream()
        _hip.check(L.skr_pack_grad_rows(_hip.ptr(ids), cap, _hip.ptr(self._gV), _hip.ptr(self._gb), 64, _hip.ptr(pack), st))
        self.ctx.all_gather_rows(gathered, pack)
        _hip.check(L.skr_unpack_grad_rows_sorted(_hip.ptr(gathered), cap, world, _hip.ptr(self._gV), _hip.ptr(self._gb), 64,
                                          _hip.ptr(opt.touch), _hip.ptr(opt.grad) if opt.touch is not None else None, st))

    def train_block(self, users, pos, neg, bounds, loss_out):
        """k consecutive GLOBAL batches (``bounds`` = [(start, stop), ...] into the epoch columns users / pos / neg,
        identical on every rank) with the temporally blocked Adam: every rank sees the whole block, so the hot rows
        (its own users of the block, every item of the block) are known without communication.  Same results as
        ``train_step`` per batch.  ``loss_out[i]`` receives the global (bpr sum, l2) of batch i."""
        world, rank, nl, ni = self.ctx.world, self.ctx.rank, self.n_local, self.num_items
        lo, hi = bounds[0][0], bounds[-1][1]
        ub, ib, jb = users[lo:hi], pos[lo:hi], neg[lo:hi]
        kk, bsz = len(bounds), bounds[0][1] - bounds[0][0]
        if all(b - a == bsz for a, b in bounds):
            # step-major list (5 * batch ids per step): a hot step names the rows of its own batch and of the next one.
            # Users of other ranks are replaced by an id the step names anyway (its first positive item): duplicates
            # are harmless, and every step keeps the same number of entries.
            U, I, J = ub.view(kk, bsz), ib.view(kk, bsz), jb.view(kk, bsz)
            if world > 1:
                U = torch.where((U % world) == rank, torch.div(U, world, rounding_mode="floor").int(), I[:, :1] + nl)
            ids = torch.cat([U, I + nl, J + nl, (I >> 6) + (nl + ni), (J >> 6) + (nl + ni)], dim=1).reshape(-1)
            self.optimizer.begin_block(ids, kk, per_step=5 * bsz)
        else:   # a ragged last batch: every hot step names the whole list
            mine = ub[(ub % world) == rank] if world > 1 else ub
            local = torch.div(mine, world, rounding_mode="floor").int() if world > 1 else mine
            self.optimizer.begin_block(torch.cat([local, ib + nl, jb + nl, (ib >> 6) + (nl + ni), (jb >> 6) + (nl + ni)]), kk)
        if world > 1 and all(b - a == bsz for a, b in bounds):
            # the whole GLOBAL batch goes to the kernel, which keeps the triples of the users this rank owns
            # (skr_bpr_step_sharded): no selection, no compaction, no count read back.  The rows to exchange are the global
            # batch's items -- the same list on every rank, made unique for all k steps of the block in one call
            U, I, J = ub.view(kk, bsz), ib.view(kk, bsz), jb.view(kk, bsz)
            step_ids = unique_padded_rows(torch.cat([I, J], dim=1))
            opt = self.optimizer
            # the loss sums are only reported: every step adds its part into its own row, ONE all-reduce per block sums the
            # rows over the ranks (a collective per step just for them doubled the step's collectives)
            lo = loss_out[:kk]
            lo.zero_()
            for k in range(kk):
                _hip.check(_hip.lib().skr_bpr_step_sharded(
                    _hip.ptr(self.user_rows), _hip.ptr(self.item_rows), _hip.ptr(self.item_bias), _hip.ptr(self.user_rows),
                    _hip.ptr(self.item_rows), _hip.ptr(U[k]), _hip.ptr(I[k]), _hip.ptr(J[k]), bsz, 1.0, self.reg, 1.0,
                    _hip.ptr(self._gU), _hip.ptr(self._gV), _hip.ptr(self._gb), _hip.ptr(self._gU), _hip.ptr(self._gV),
                    _hip.ptr(lo[k]), _hip.ptr(opt.touch), _hip.ptr(opt.grad) if opt.touch is not None else None,
                    world, rank, 1.0, _hip.stream()))
                self._exchange_item_grads(bsz, None, None, ids=step_ids[k:k + 1])
                self.optimizer.hot_step()
            self.ctx.all_reduce(lo)
        else:
            for k, (a, b) in enumerate(bounds):
                self._step_grads(users[a:b], pos[a:b], neg[a:b])
                loss_out[k] = self.loss
                self.optimizer.hot_step()
        self.optimizer.end_blocks()

    def _step_grads(self, users, pos, neg):
        """the gradients of one global batch in the dense buffer (local part computed, item part exchanged)"""
        world, rank = self.ctx.world, self.ctx.rank
        if world > 1:
            sel = (users % world) == rank
            ul = torch.div(users[sel], world, rounding_mode="floor").int().contiguous()
            il, jl = pos[sel].contiguous(), neg[sel].contiguous()
        else:
            ul, il, jl = users.contiguous(), pos.contiguous(), neg.contiguous()
        self._step_local(ul, il, jl, users.numel())

    def _step_local(self, ul, il, jl, n_global):
        """ul / il / jl: this rank's triples of a global batch of n_global (local user indices), contiguous int32"""
        world = self.ctx.world
        self.loss.zero_()
        opt = self.optimizer
        if ul.numel() > 0:
            _hip.check(_hip.lib().skr_bpr_step(
                _hip.ptr(self.user_rows), _hip.ptr(self.item_rows), _hip.ptr(self.item_bias), _hip.ptr(self.user_rows),
                _hip.ptr(self.item_rows), _hip.ptr(ul), _hip.ptr(il), _hip.ptr(jl), ul.numel(), 1.0, self.reg, 1.0,
                _hip.ptr(self._gU), _hip.ptr(self._gV), _hip.ptr(self._gb), _hip.ptr(self._gU), _hip.ptr(self._gV),
                _hip.ptr(self.loss), _hip.ptr(opt.touch), _hip.ptr(opt.grad) if opt.touch is not None else None, _hip.stream()))
        if world > 1:
            self._exchange_item_grads(n_global, il, jl)    # the one exchange step
        self.ctx.all_reduce(self.loss)

    def train_step(self, users, pos, neg):
        """users/pos/neg: int32 device tensors of the GLOBAL batch, identical on every rank"""
        self._step_grads(users, pos, neg)
        self.optimizer.step()

    def gather_user_table(self):
        full = torch.zeros((self.num_users, 64), dtype=torch.float32, device=self.device)
        full[torch.from_numpy(self.mine).to(self.device)] = self.user_rows
        self.ctx.all_reduce(full)
        return full
