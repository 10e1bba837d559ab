"""Data side of the drop-in package (reference: skrec/io/__init__.py:4-21).  ``PairwiseSampler`` /
``PointwiseSampler`` are the names BASELINE.json uses for the two hot-path iterators."""
from .data_iterator import (InteractionIterator, ItemVecIterator, KGPairwiseIterator, PairwiseIterator,
                            PairwiseSampler, PointwiseIterator, PointwiseSampler, SequentialPairwiseIterator,
                            SequentialPointwiseIterator, UserVecIterator)
from .dataset import ImplicitFeedback, KnowledgeGraph, RSDataset, UserGroup, group_users_by_interactions
from .logger import Logger

__all__ = [
    "ImplicitFeedback", "KnowledgeGraph", "RSDataset", "UserGroup", "group_users_by_interactions",
    "InteractionIterator", "PairwiseIterator", "PointwiseIterator", "PairwiseSampler", "PointwiseSampler",
    "SequentialPairwiseIterator", "SequentialPointwiseIterator", "UserVecIterator", "ItemVecIterator",
    "KGPairwiseIterator", "Logger",
]
