"""skrec -- drop-in for scikit-recommender's hot path on AMD MI355X (gfx950).

Same import paths as the reference for everything ``run_skrec.py`` and the in-scope models touch
(``skrec.RunConfig``, ``skrec.ModelRegistry``, ``skrec.merge_config_with_cmd_args``, ``skrec.io``,
``skrec.utils.py``, ``skrec.recommender.{BPRMF,LightGCN,LayerGCN}``).  The heavy lifting happens in
``libskrec_hip.so`` (hand-written HIP kernels, C ABI in ``include/skrec_hip.h``).
"""
from .io import *  # noqa: F401,F403
from .io import RSDataset, ImplicitFeedback, PairwiseIterator, PointwiseIterator, InteractionIterator, Logger
from .utils.py import *  # noqa: F401,F403
from .utils.py import (BatchIterator, randint_choice, batch_randint_choice, RankingEvaluator, MetricReport,
                       EarlyStopping, Config, ModelConfig, merge_config_with_cmd_args)
from .utils import ModelRegistry
from .run_config import RunConfig
from .version import __version__
