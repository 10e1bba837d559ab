from .generic import OrderedDefaultDict
from .generic import pad_sequences
from .generic import md5sum
from .generic import slugify

from .batch_iterator import BatchIterator

from .decorator import timer
from .decorator import typeassert

from .random import randint_choice
from .random import batch_randint_choice

from .evaluator import RankingEvaluator
from .evaluator import MetricReport
from .evaluator import EarlyStopping

from .config import Config
from .config import ModelConfig
from .config import merge_config_with_cmd_args
