"""``AbstractRecommender`` (reference: skrec/recommender/base.py:20-74): dataset, logger, evaluator and
the activity groups are built here; subclasses provide ``fit`` / ``evaluate`` / ``predict``."""
import os
import platform
import time
from typing import List, Tuple, Union

import numpy as np

from ..io import Logger, RSDataset, group_users_by_interactions
from ..run_config import RunConfig
from ..utils.py import Config, MetricReport, RankingEvaluator, slugify
from ..version import __version__

__all__ = ["AbstractRecommender", "DenseAdam", "on_compute_stream"]


def on_compute_stream(fit):
    """Decorator of ``fit()``: the training loop and its evaluations run on a HIP stream of the library's own instead of
    the device's null stream, ordered behind the caller's stream on entry and in front of it on exit.  A launch on the null
    stream costs the host more than one on a created stream (the runtime orders it against every other stream it knows:
    this library keeps three to five), measured on an MI355X with ``bench.py``: BPRMF epoch 1.14 -> 1.07 s, the 20-step
    slice 35.0 -> 36.6 M interactions/s, batch 16 384: 132 -> 148 M/s; the graph models and GRU4RecPlus are unchanged.
    A caller that has already chosen a stream keeps it; ``SKR_COMPUTE_STREAM=0`` switches this off."""
    import functools

    @functools.wraps(fit)
    def wrapper(self, *args, **kwargs):
        import torch
        if os.environ.get("SKR_COMPUTE_STREAM", "1") == "0" or not torch.cuda.is_available():
            return fit(self, *args, **kwargs)
        cur = torch.cuda.current_stream()
        if cur != torch.cuda.default_stream():
            return fit(self, *args, **kwargs)
        own = getattr(self, "_compute_stream", None)
        if own is None:
            own = self._compute_stream = torch.cuda.Stream()
        own.wait_stream(cur)
        try:
            with torch.cuda.stream(own):
                return fit(self, *args, **kwargs)
        finally:
            cur.wait_stream(own)
    return wrapper


class AbstractRecommender(object):
    def __init__(self, run_config: RunConfig, model_config: Config):
        self.run_config = run_config
        self.dataset = RSDataset(run_config.data_dir, run_config.sep, run_config.file_column)
        self.logger: Logger = self._create_logger(self.dataset, model_config)
        self.dataset.set_logger(self.logger)
        self.evaluator = RankingEvaluator(self.dataset.train_data.to_user_dict(),
                                          self.dataset.test_data.to_user_dict(),
                                          metric=run_config.metric, top_k=run_config.top_k,
                                          batch_size=run_config.test_batch_size,
                                          num_thread=run_config.test_thread)
        self._user_groups = group_users_by_interactions(self.dataset)

    def _create_logger(self, dataset: RSDataset, config: Config) -> Logger:
        model_name = self.__class__.__name__
        run_id = slugify(f"{dataset.data_name}_{model_name}_{config.to_string('_')}", max_length=255 - 100)
        run_id = f"{run_id}_{time.time():.8f}"
        logger = Logger(os.path.join("log", dataset.data_dir.lstrip(os.sep), model_name, run_id + ".log"))
        logger.info(f"Server:\t{platform.node()}")
        logger.info(f"Workspace:\t{os.getcwd()}")
        logger.info(f"PID:\t{os.getpid()}")
        logger.info(f"skrec version:\tv{__version__}")
        logger.info(f"Model:\t{self.__class__.__module__}")
        logger.info(f"\n{dataset.statistic_info}")
        cfg = config.to_string("\n")
        logger.info(f"\nHyper-parameters:\n{cfg}\n")
        return logger

    def fit(self) -> MetricReport:
        raise NotImplementedError

    def evaluate(self, test_users=None) -> MetricReport:
        raise NotImplementedError

    def evaluate_group(self) -> List[Tuple[str, MetricReport]]:
        return [(g.label, self.evaluate(g.users)) for g in self._user_groups]

    def predict(self, users: Union[List[int], np.ndarray]) -> np.ndarray:
        raise NotImplementedError


class DenseAdam(object):
    """State of ``torch.optim.Adam(params, lr)`` for ONE flat fp32 buffer holding every parameter of the
    model, stepped by a single ``skr_adam_step`` launch per training step (betas 0.9/0.999, eps 1e-8,
    no weight decay: the reference's defaults, BPRMF.py:99).  Every element is updated every step,
    like the reference's dense Adam; with ``track_touch`` a byte per 64-float block lets the kernel
    skip READING gradients that are known to be zero (same result, 24 instead of 32 B/param)."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, track_touch=False, tf_epsilon=False):
        import torch
        assert flat.dim() == 1 and flat.is_contiguous()
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.touch = torch.zeros((flat.numel() + 63) // 64, dtype=torch.uint8, device=flat.device) if track_touch else None
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        # tf.train.AdamOptimizer puts epsilon outside the bias correction: p -= lr_t * m / (sqrt(v) + eps) with
        # lr_t = lr * sqrt(1-b2^t) / (1-b1^t)   (GRU4RecPlus.py:192): the skr_adam_*_tf entry points
        self.tf_epsilon = bool(tf_epsilon)
        self.t = 0

    # ---- temporally blocked stepping (bit-identical to step() after every batch) -----------------------------
    def begin_block(self, block_ids, k, per_step=None):
        """``block_ids``: int32 device tensor with the index (float offset / 64) of every 64-float block that any
        of the coming k steps touches, duplicates allowed.  Tags those blocks and gives every OTHER block its k
        zero-gradient updates in one pass; ``hot_step()`` must then be called once per step, k times.

        ``per_step``: the list is step-major with this many entries per step (k * per_step in all).  A hot step then
        visits only the rows of its own batch and of the next one (which is about to read them) and lets every other
        hot row catch up when its turn comes; the k-th step visits all of them.  Same updates, same order."""
        import torch
        from .. import _hip
        L, st = _hip.lib(), _hip.stream()
        if getattr(self, "_blk_tag", None) is None:
            nb = (self.flat.numel() + 63) // 64
            self._blk_tag = torch.zeros(nb, dtype=torch.int32, device=self.flat.device)
            self._blk_claim = torch.zeros(nb, dtype=torch.int32, device=self.flat.device)
            self._blk_serial = 0
        cur = torch.cuda.current_stream()
        if not self._ensure_side():
            cur.wait_event(self._ev_cold)    # the previous block's cold pass still reads the tags and writes cold rows
        assert per_step is None or block_ids.numel() == int(k) * int(per_step)
        self._blk_serial += 1
        self._blk_ids = block_ids           # kept alive until the block is done
        _hip.check(L.skr_adam_block_mark(_hip.ptr(block_ids), block_ids.numel(), 0, 64, _hip.ptr(self._blk_tag),
                                         self._blk_serial, _hip.ptr(self._blk_claim), self.t, st))
        # the cold pass touches no row the block's batches read or write, so it runs on a side stream underneath the
        # k small bpr / hot-step launches
        self._ev_marked.record(cur)
        self._side.wait_event(self._ev_marked)
        self.launch_cold(self._blk_tag, self._blk_serial, int(k))
        self._hot = (L.skr_adam_block_hot_tf if self.tf_epsilon else L.skr_adam_block_hot, self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                     self.flat.numel(), block_ids.data_ptr(), block_ids.numel(), self._blk_claim.data_ptr(), st,
                     self.t, int(k), None if per_step is None else int(per_step))

    def _ensure_side(self):
        """the side stream of the cold passes and its events; True when they were created by this call"""
        import os
        import torch
        if getattr(self, "_side", None) is not None:
            return False
        cur = torch.cuda.current_stream()
        self._side = torch.cuda.Stream(device=self.flat.device) if os.environ.get("SKR_ADAM_OVERLAP", "1") != "0" else cur
        self._ev_marked, self._ev_cold = torch.cuda.Event(), torch.cuda.Event()
        return True

    def launch_cold(self, tag, serial, k):
        """the cold pass of a k-step block that starts at step self.t, on the side stream (the caller has made that stream
        wait for whatever the pass depends on); records ``_ev_cold`` behind it"""
        import torch
        from .. import _hip
        timing = getattr(self, "cold_timing", None)     # measurement hook (bench.py): a list that receives (start, end, k)
        pool = getattr(self, "cold_event_pool", None)   # ... with event pairs made beforehand (no event creation in a timed region);
        skip = getattr(self, "cold_timing_skip", 0)     # ... after this many unbracketed passes
        if timing is not None and skip > 0:
            self.cold_timing_skip, timing = skip - 1, None
        if timing is not None and pool is not None and not pool:
            timing = None                               # when the pool is used up the passes are no longer bracketed
        if timing is not None:
            e0, e1 = pool.pop() if pool is not None else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record(self._side)
        L = _hip.lib()
        _hip.check((L.skr_adam_block_cold_tf if self.tf_epsilon else L.skr_adam_block_cold)(
            _hip.ptr(self.flat), _hip.ptr(self.m), _hip.ptr(self.v), self.flat.numel(), self.lr,
            self.betas[0], self.betas[1], self.eps, self.t, int(k), _hip.ptr(tag), int(serial), self._side.cuda_stream))
        if timing is not None:
            e1.record(self._side)
            timing.append((e0, e1, int(k)))
        self._ev_cold.record(self._side)

    def end_blocks(self):
        """join the side stream: call after the last block, before anything else reads the parameters"""
        import torch
        if getattr(self, "_side", None) is not None:
            torch.cuda.current_stream().wait_event(self._ev_cold)

    def hot_step(self):
        """this step's update of the hot blocks (see begin_block); their gradients are consumed"""
        fn, pp, pg, pm, pv, n, pids, nids, pclaim, st, t0, k, per = self._hot
        self.t += 1
        s = self.t - t0 - 1                      # step inside the block
        if per is not None and s < k - 1:        # rows of batch s and of batch s + 1
            pids, nids = pids + 4 * s * per, 2 * per
        rc = fn(pp, pg, pm, pv, n, self.lr, self.betas[0], self.betas[1], self.eps, t0, self.t, pids, nids, 0, 64, pclaim, st)
        if rc:
            from .. import _hip
            _hip.check(rc)

    def grad_view(self, start, shape):
        n = 1
        for d in shape:
            n *= d
        return self.grad[start:start + n].view(*shape)

    def step(self):
        from .. import _hip
        self.t += 1
        L = _hip.lib()
        _hip.check((L.skr_adam_step_tf if self.tf_epsilon else L.skr_adam_step)(
            _hip.ptr(self.flat), _hip.ptr(self.grad), _hip.ptr(self.m), _hip.ptr(self.v), self.flat.numel(), self.lr, self.betas[0],
            self.betas[1], self.eps, self.t, 1, _hip.ptr(self.touch), _hip.stream()))
