"""Host side of the one-launch BPRMF step (csrc/train.hip K2c: ``skr_bpr_fused_step`` / ``skr_bpr_fused_end``).

The reference's step is ``loss.backward(); optimizer.step()`` with a DENSE Adam (BPRMF.py:108-127): every row moves at
every step.  The blocked optimiser (base.DenseAdam.begin_block) already gives the rows no batch of a k-step block touches
their k zero-gradient updates in one pass; this module prepares what lets the TOUCHED rows be evaluated lazily inside the
BPR kernel itself, so that a step is one launch instead of two dependent ones: for every reference (step, row) of a block
the row's slot in the block's workspace, how many earlier steps named the row, the step of the previous naming, and one
owner per (step, row) pair.  All of it follows from the epoch's batches, which are known before the epoch starts
(data_iterator.py:230-234 builds them from one permutation); ``skr_bpr_fused_plan`` computes it per block on the device --
no value is read back.
"""
import torch

from .. import _hip

SLOT_BITS = 20


def build_fused_meta(cu, ci, cj, n_blocks, k, bsz, user_block0, item_block0, bias_block0, chunk_blocks=64, prev_hot=None):
    """The words ``skr_bpr_fused_plan`` writes, derived independently with sorts and scans (tests compare the two, modulo
    the numbering of the slots).  cu / ci / cj: the epoch's int32 columns (step-major, contiguous); the first n_blocks * k * bsz entries are the full
    blocks.  Returns (meta [n_blocks, k, 5, bsz] int32, slot_block [n_blocks, L] int32 (-1 padded), slot_fin [n_blocks, L]
    int32, n_slots [n_blocks] int32) with L = k * 5 * bsz; formats: include/skrec_hip.h, skr_bpr_fused_step."""
    L = k * 5 * bsz
    assert L <= (1 << SLOT_BITS) and k <= 64
    dev = cu.device
    rows = n_blocks * k * bsz
    meta = torch.empty((n_blocks, L), dtype=torch.int32, device=dev)
    slot_block = torch.empty((n_blocks, L), dtype=torch.int32, device=dev)
    slot_fin = torch.empty((n_blocks, L), dtype=torch.int32, device=dev)
    n_slots = torch.empty(n_blocks, dtype=torch.int32, device=dev)
    U, I, J = (c[:rows].view(n_blocks, k, bsz) for c in (cu, ci, cj))
    step = torch.arange(k, device=dev, dtype=torch.int64).view(1, k, 1, 1)
    idx = torch.arange(L, device=dev, dtype=torch.int64).view(1, L)
    for b0 in range(0, n_blocks, chunk_blocks):
        b1 = min(n_blocks, b0 + chunk_blocks)
        nb = b1 - b0
        u, i, j = U[b0:b1].long(), I[b0:b1].long(), J[b0:b1].long()
        refs = torch.stack([u + user_block0, i + item_block0, j + item_block0, (i >> 6) + bias_block0, (j >> 6) + bias_block0],
                           dim=2)                                            # [nb, k, 5, bsz]: 64-float blocks of the flat buffer
        key = (refs * 64 + step).view(nb, L)
        skey, perm = torch.sort(key, dim=1)
        blk, stp = skey >> 6, skey & 63
        new_pair = torch.ones((nb, L), dtype=torch.bool, device=dev)
        new_pair[:, 1:] = skey[:, 1:] != skey[:, :-1]
        new_row = torch.ones((nb, L), dtype=torch.bool, device=dev)
        new_row[:, 1:] = blk[:, 1:] != blk[:, :-1]
        c_pair = torch.cumsum(new_pair, dim=1)                               # pairs seen so far (1-based)
        zero = torch.zeros((), dtype=torch.int64, device=dev)
        row_first = torch.cummax(torch.where(new_row, c_pair, zero), dim=1).values
        n0 = c_pair - row_first                                              # earlier namings of the row in this block
        slot = torch.cumsum(new_row, dim=1) - 1
        pair_start = torch.cummax(torch.where(new_pair, idx, zero), dim=1).values
        prev_step = torch.gather(stp, 1, (pair_start - 1).clamp_(min=0))
        first_naming = torch.gather(new_row, 1, pair_start)
        prev1 = torch.where(first_naming, zero, prev_step + 1)
        n0f = n0 % 6
        if prev_hot is not None:      # [n_blocks, n_flat_blocks] bool: the rows the block BEFORE named (skr_bpr_fused_plan2)
            cold_before = ~torch.gather(prev_hot[b0:b1], 1, blk)
            n0f = torch.where((n0 == 0) & (stp > 0) & cold_before, torch.full_like(n0f, 7), n0f)
        word = slot | (n0f << SLOT_BITS) | (new_pair.long() << 23) | (prev1 << 24)
        meta[b0:b1].scatter_(1, perm, word.int())
        # per slot: its block of the flat buffer; number of namings and the step of the last one
        sb = torch.full((nb, L + 1), -1, dtype=torch.int32, device=dev)
        sb.scatter_(1, torch.where(new_row, slot, torch.full_like(slot, L)), blk.int())
        slot_block[b0:b1] = sb[:, :L]
        row_last = torch.ones((nb, L), dtype=torch.bool, device=dev)
        row_last[:, :-1] = new_row[:, 1:]
        sf = torch.zeros((nb, L + 1), dtype=torch.int32, device=dev)
        sf.scatter_(1, torch.where(row_last, slot, torch.full_like(slot, L)), (((n0 + 1) % 6) | (stp << 8)).int())
        sf1 = torch.zeros((nb, L + 1), dtype=torch.int32, device=dev)        # ... | (step of the first naming << 16)
        sf1.scatter_(1, torch.where(new_row, slot, torch.full_like(slot, L)), (stp << 16).int())
        slot_fin[b0:b1] = (sf | sf1)[:, :L]
        n_slots[b0:b1] = (slot[:, -1] + 1).int()
    return meta.view(n_blocks, k, 5, bsz), slot_block, slot_fin, n_slots


class FusedBlocks(object):
    """Runs k-step blocks of full batches through ``skr_bpr_fused_step``.  Three streams:

    * planning stream: block n + 1's words (``skr_bpr_fused_plan``) and its hot-block tags, while block n runs;
    * current stream: the k step launches of block n, then the write-back of the rows block n + 1 touches too;
    * the optimiser's side stream: block n's cold pass, then the write-back of the REST of block n's rows (beside block
      n + 1's steps, before block n + 1's cold pass, which updates those rows).

    Two sets of words / slot tables / tags / workspaces alternate, so that a block's leftovers can be written back while the
    next block accumulates.  ``opt``: the model's DenseAdam over the flat [U | V | bias] buffer; table offsets in 64-float
    blocks.

    The split write-back is OFF by default (``SKR_FUSED_SPLIT_END=1`` turns it on): it takes 50 of the write-back's 63 us
    off the current stream, but the side stream -- whose cold pass (0.54 ms of a 0.70 ms block in the steady state of an
    epoch) is the other bound -- gets them, and the next cold pass starts that much later: an epoch took 1.19 s instead of
    1.09 s.  Without it the whole write-back runs on the current stream after the block's last step."""

    def __init__(self, opt, user_block0, item_block0, bias_block0, reg):
        import os
        self.opt = opt
        self.offsets = (int(user_block0), int(item_block0), int(bias_block0))
        self.reg = float(reg)
        self.cap = 0
        dev = opt.flat.device
        self.n_flat_blocks = (opt.flat.numel() + 63) // 64
        self.scratch = torch.zeros(28 * self.n_flat_blocks // 8 + 1, dtype=torch.int64, device=dev)   # zero between calls
        self.tags = torch.zeros((2, self.n_flat_blocks), dtype=torch.int32, device=dev)
        self.serial = 0
        self.split_end = os.environ.get("SKR_FUSED_SPLIT_END", "0") == "1"
        # the zero-gradient updates a row needs before its FIRST naming in a block are made one block ahead, on the optimiser's
        # side stream behind the previous block's cold pass, for the rows that block did not touch (skr_bpr_fused_pre);
        # SKR_FUSED_PRE=0: the step launch that first names a row catches it up itself
        self.pre_advance = os.environ.get("SKR_FUSED_PRE", "1") != "0"
        self._ev_pre = [torch.cuda.Event(), torch.cuda.Event()]
        self._plan_stream = torch.cuda.Stream(device=dev)
        self._ev_plan = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev_block = torch.cuda.Event()
        self._used = [False, False]

    def _size(self, k, bsz):
        # the workspace is indexed by slot = distinct row of the block: at most one per reference (k * 5 * bsz) and at most one
        # per 64-float block of the flat buffer.  9 planes of cap * 256 B (+ 6 with the pre buffers): 380 MB at k = 32,
        # b = 1 024 -- SKR_ADAM_BLOCK / the batch size set it
        cap = min(k * 5 * bsz, self.n_flat_blocks)
        self.refs = k * 5 * bsz
        if cap > self.cap or self.refs > getattr(self, "_refs_cap", 0):
            dev = self.opt.flat.device
            torch.cuda.synchronize(dev)               # nothing in flight refers to the buffers that are replaced
            self.cap, self._refs_cap = max(cap, self.cap), max(self.refs, getattr(self, "_refs_cap", 0))
            cap = self.cap
            # zero: the invariant between blocks.  One workspace unless a block's leftovers are written back beside the next block
            self.work = torch.zeros((2 if self.split_end else 1, 9 * cap * 64), dtype=torch.float32, device=dev)
            self.meta, self.slot_block, self.slot_fin = (torch.empty((2, self._refs_cap), dtype=torch.int32, device=dev) for _ in range(3))
            self.n_slots = torch.zeros((2, 1), dtype=torch.int32, device=dev)
            self.pre = torch.zeros((2, 3 * cap * 64), dtype=torch.float32, device=dev) if self.pre_advance else None
            self._used = [False, False]

    def _plan(self, q, pu, pi, pj, k, bsz, serial, prev=None, inline=False):
        """words, slot tables and hot-block tags of one block into set q, on the planning stream; ``prev`` = (tags, value) of
        the block before it: first namings of rows that block did not touch are marked as pre-advanced"""
        u0, i0, b0 = self.offsets
        # (``inline``: the first block of a call has nothing to be planned beside -- its words go out on the current stream,
        #  without the two stream hand-offs)
        ps, L = (torch.cuda.current_stream() if inline else self._plan_stream), _hip.lib()
        if self._used[q]:
            ps.wait_event(self._ev_done[q])       # the set's previous block is done with its tables and its workspace
        # the tags of set q were last read by the cold pass of the set's previous block (and by the write-backs of the
        # block before that one, which sit in front of it on the side stream)
        ps.wait_event(self.opt._ev_cold)
        rc = L.skr_bpr_fused_plan2(pu, pi, pj, bsz, k, u0, i0, b0, self.n_flat_blocks, self.scratch.data_ptr(),
                                   self.meta[q].data_ptr(), self.slot_block[q].data_ptr(), self.slot_fin[q].data_ptr(),
                                   self.n_slots[q].data_ptr(), prev[0].data_ptr() if prev else None, prev[1] if prev else 0,
                                   ps.cuda_stream)
        rc |= L.skr_adam_block_mark(self.slot_block[q].data_ptr(), k * 5 * bsz, 0, 64, self.tags[q].data_ptr(), serial, None,
                                    self.opt.t, ps.cuda_stream)
        if rc:
            _hip.check(rc)
        self._ev_plan[q].record(ps)

    def run_blocks(self, pu, pi, pj, n_blocks, k, bsz, ploss, loss_stride_bytes):
        """n_blocks * k batches of bsz triples each at the device addresses pu / pi / pj (int32, step-major); the loss sums
        of step s go to ploss + s * loss_stride_bytes (SKR_LOSS_SLOTS pairs of floats each)"""
        if n_blocks <= 0:
            return
        opt, L, st = self.opt, _hip.lib(), _hip.stream()
        self._size(k, bsz)
        cur = torch.cuda.current_stream()
        if opt._ensure_side():
            opt._ev_cold.record(cur)              # nothing to wait for yet
        side = opt._side
        blk_bytes = 4 * k * bsz
        u0, i0, b0 = self.offsets
        pp, pm, pv, n_par = opt.flat.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr(), opt.flat.numel()
        cap = self.cap
        lr, (b1, b2), eps, reg = opt.lr, opt.betas, opt.eps, self.reg
        self._plan_stream.wait_stream(cur)        # the columns were produced on the current stream
        side.wait_stream(cur)                     # ... and whatever wrote the tables before is on it too
        serial0 = self.serial + 1
        self.serial += n_blocks
        self._plan(0, pu, pi, pj, k, bsz, serial0, inline=True)
        self._plan_stream.wait_event(self._ev_plan[0])     # the planning kernels share one scratch: block 1's come after block 0's
        for blk in range(n_blocks):
            q, o = blk & 1, blk * blk_bytes
            more = blk + 1 < n_blocks
            pre_next = more and self.pre_advance
            if more:
                self._plan(q ^ 1, pu + o + blk_bytes, pi + o + blk_bytes, pj + o + blk_bytes, k, bsz, serial0 + blk + 1,
                           prev=(self.tags[q], serial0 + blk) if pre_next else None)
            cur.wait_event(self._ev_plan[q])      # this block's words and tags
            cur.wait_event(opt._ev_cold)          # the previous cold pass wrote rows this block may read
            pre_now = blk > 0 and self.pre_advance
            if pre_now:
                cur.wait_event(self._ev_pre[q])   # ... and the rows advanced ahead of this block are in its pre buffer
            if self._used[q]:
                cur.wait_event(self._ev_done[q])  # the set's workspace: its previous block's leftovers are written back
            side.wait_event(self._ev_plan[q])
            opt.launch_cold(self.tags[q], serial0 + blk, k)          # behind the previous block's write-back on that stream
            t0 = opt.t
            if pre_next:
                # behind this block's cold pass, beside its steps: the next block's first namings of rows this block does not touch
                side.wait_event(self._ev_plan[q ^ 1])
                rc_pre = L.skr_bpr_fused_pre(pp, pm, pv, n_par, self.pre[q ^ 1].data_ptr(), cap, self.slot_block[q ^ 1].data_ptr(),
                                             self.slot_fin[q ^ 1].data_ptr(), self.n_slots[q ^ 1].data_ptr(), lr, b1, b2, eps, t0 + k, k,
                                             self.tags[q].data_ptr(), serial0 + blk, side.cuda_stream)
                if rc_pre:
                    _hip.check(rc_pre)
                self._ev_pre[q ^ 1].record(side)
            ppre = self.pre[q].data_ptr() if pre_now else None
            split = more and self.split_end
            if split:
                cur.wait_event(self._ev_plan[q ^ 1])                 # the next block's tags decide what is written back where
            ptag = self.tags[q ^ 1].data_ptr() if split else None
            pw = self.work[q if self.split_end else 0].data_ptr()
            # measurement hook (bench.py): `block_timing` receives (start, steps done, end done, k) -- HIP events on the current
            # stream around the block's k step launches and around its end launch -- with event triples from `block_event_pool`
            # (made beforehand), after `block_timing_skip` unbracketed blocks; no pool or an empty one: no bracket.  A
            # bracketed block is issued launch by launch from here (same launches as skr_bpr_fused_block makes).
            bt = getattr(self, "block_timing", None)
            if bt is not None and getattr(self, "block_timing_skip", 0) > 0:
                self.block_timing_skip -= 1
                bt = None
            trip = self.block_event_pool.pop() if (bt is not None and getattr(self, "block_event_pool", None)) else None
            if trip is None:
                rc = L.skr_bpr_fused_block2(pp, pm, pv, n_par, pw, cap, pu + o, pi + o, pj + o, self.meta[q].data_ptr(), bsz, u0, i0, b0,
                                            lr, b1, b2, eps, t0, k, reg, ploss + blk * k * loss_stride_bytes, loss_stride_bytes // 4,
                                            self.slot_block[q].data_ptr(), self.slot_fin[q].data_ptr(), self.n_slots[q].data_ptr(),
                                            ptag, serial0 + blk + 1, ppre, st)
            else:
                rc, pmeta = 0, self.meta[q].data_ptr()
                trip[0].record(cur)
                for s_ in range(k):
                    os_ = o + 4 * s_ * bsz
                    rc |= L.skr_bpr_fused_step2(pp, pm, pv, n_par, pw, cap, pu + os_, pi + os_, pj + os_, pmeta + 20 * s_ * bsz, bsz, u0,
                                                i0, b0, lr, b1, b2, eps, t0, k, s_, reg, ploss + (blk * k + s_) * loss_stride_bytes,
                                                ppre, st)
                trip[1].record(cur)
                rc |= L.skr_bpr_fused_end(pp, pm, pv, n_par, pw, cap, self.slot_block[q].data_ptr(), self.slot_fin[q].data_ptr(),
                                          self.n_slots[q].data_ptr(), lr, b1, b2, eps, t0, k, ptag, serial0 + blk + 1,
                                          1 if ptag else 0, st)
                trip[2].record(cur)
                bt.append((trip[0], trip[1], trip[2], k))
            self._ev_block.record(cur)
            side.wait_event(self._ev_block)       # the next cold pass updates rows this block has just written back
            if split:
                # the rows the next block does not touch go back beside its steps, in front of its cold pass
                side.wait_event(self._ev_plan[q ^ 1])
                rc |= L.skr_bpr_fused_end(pp, pm, pv, n_par, pw, cap, self.slot_block[q].data_ptr(), self.slot_fin[q].data_ptr(),
                                          self.n_slots[q].data_ptr(), lr, b1, b2, eps, t0, k, ptag, serial0 + blk + 1, 2,
                                          side.cuda_stream)
                self._ev_done[q].record(side)
            else:
                self._ev_done[q].record(cur)
            self._used[q] = True
            self.last_q = q
            opt.t = t0 + k
            if rc:
                _hip.check(rc)
        if self.split_end and n_blocks > 1:
            # the last write-back on the side stream is in front of nothing: make the optimiser's event cover it
            opt._ev_cold.record(side)
