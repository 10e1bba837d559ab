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
    """State of ``torch.optim.Adam(params, lr)`` for dense fp32 tensors, stepped by ``skr_adam_step``
    (betas 0.9/0.999, eps 1e-8, no weight decay: the reference's defaults, BPRMF.py:99)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        import torch
        self.params = list(params)
        self.grads = [torch.zeros_like(p) for p in self.params]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.t = 0

    def step(self):
        from .. import _hip
        self.t += 1
        L, st = _hip.lib(), _hip.stream()
        for p, g, m, v in zip(self.params, self.grads, self.m, self.v):
            _hip.check(L.skr_adam_step(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), p.numel(), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.t, 1, st))
