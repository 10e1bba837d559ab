"""Host-side utilities of the drop-in package.  The names exported here are the ones the reference's
models import from ``skrec.utils.py`` (reference: skrec/utils/py/__init__.py:4-23)."""
from .batch_iterator import BatchIterator
from .config import Config, ModelConfig, merge_config_with_cmd_args
from .decorator import timer, typeassert
from .evaluator import EarlyStopping, MetricReport, RankingEvaluator
from .generic import OrderedDefaultDict, md5sum, pad_sequences, slugify
from .random import batch_randint_choice, randint_choice

__all__ = [
    "BatchIterator",
    "Config", "ModelConfig", "merge_config_with_cmd_args",
    "EarlyStopping", "MetricReport", "RankingEvaluator",
    "OrderedDefaultDict", "md5sum", "pad_sequences", "slugify",
    "batch_randint_choice", "randint_choice",
    "timer", "typeassert",
]
