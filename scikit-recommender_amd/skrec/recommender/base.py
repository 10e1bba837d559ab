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

__all__ = ["AbstractRecommender", "DenseAdam"]


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
        # lr_t = lr * sqrt(1-b2^t) / (1-b1^t), i.e. torch's form with eps / sqrt(1-b2^t)   (GRU4RecPlus.py:192)
        self.tf_epsilon = bool(tf_epsilon)
        self.t = 0

    def grad_view(self, start, shape):
        n = 1
        for d in shape:
            n *= d
        return self.grad[start:start + n].view(*shape)

    def step(self):
        from .. import _hip
        self.t += 1
        eps = self.eps / (1.0 - self.betas[1] ** self.t) ** 0.5 if self.tf_epsilon else self.eps
        _hip.check(_hip.lib().skr_adam_step(_hip.ptr(self.flat), _hip.ptr(self.grad), _hip.ptr(self.m), _hip.ptr(self.v),
                                            self.flat.numel(), self.lr, self.betas[0], self.betas[1], eps, self.t, 1,
                                            _hip.ptr(self.touch), _hip.stream()))
