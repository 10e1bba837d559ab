from .dataset import RSDataset
from .dataset import ImplicitFeedback
from .dataset import KnowledgeGraph
from .dataset import UserGroup, group_users_by_interactions

from .data_iterator import PointwiseIterator
from .data_iterator import PairwiseIterator
from .data_iterator import InteractionIterator
from .data_iterator import PairwiseSampler, PointwiseSampler
from .data_iterator import SequentialPointwiseIterator
from .data_iterator import SequentialPairwiseIterator
from .data_iterator import UserVecIterator
from .data_iterator import ItemVecIterator
from .data_iterator import KGPairwiseIterator

from .logger import Logger
