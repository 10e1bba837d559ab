"""Ordered config objects and the ``--key value`` command-line merge (reference: utils/py/config.py)."""
import ast
import copy
import sys
from argparse import Namespace
from collections import OrderedDict
from typing import Dict, List

import numpy as np

from ..common import PostInitMeta

__all__ = ["Config", "ModelConfig", "merge_config_with_cmd_args"]


class Config(Namespace, metaclass=PostInitMeta):
    """Namespace that remembers attribute creation order and validates itself after construction.

    Public attributes appear in ``items()`` / ``to_string()`` in the order they were first set
    (config.py:15-35); ``_validate()`` runs after the subclass ``__init__`` (config.py:41-45).
    """

    def __init__(self):
        self.__dict__["_order"] = []
        super().__init__()

    def __setattr__(self, key, value):
        self.__dict__[key] = value
        if key != "_order" and key not in self._order:
            self._order.append(key)

    def _get_kwargs(self):
        return [(k, self.__dict__[k]) for k in self._order]

    def items(self):
        yield from self._get_kwargs()

    def __post_init__(self):
        self._validate()

    def _validate(self):
        pass

    def to_string(self, sep: str = "\n"):
        return sep.join(f"{k}={v}" for k, v in self.items())


class ModelConfig(Config):
    @classmethod
    def param_space(cls) -> Dict[str, List]:
        return dict()

    @classmethod
    def num_combos(cls) -> int:
        return int(np.prod([len(v) for v in cls.param_space().values()]))


_LITERAL_TYPES = (str, int, float, list, tuple, bool, type(None))


def _parse_value(text: str):
    """'1e-3' -> 0.001, '[10,20]' -> [10, 20], 'true' -> True, anything else stays a string.
    (The reference eval()s the text, config.py:84; literal_eval accepts the same literals
    without executing code.)"""
    try:
        value = ast.literal_eval(text)
        return value if isinstance(value, _LITERAL_TYPES) else text
    except (ValueError, SyntaxError):
        low = text.lower()
        if low == "true":
            return True
        if low == "false":
            return False
        return text


def merge_config_with_cmd_args(config: Dict, inplace: bool = True) -> Dict:
    args = sys.argv[1:]
    if len(args) % 2 != 0:
        raise SyntaxError("The numbers of arguments and its values are not equal.")
    if not inplace:
        config = copy.deepcopy(config)
    overrides = OrderedDict()
    for name, value in zip(args[0::2], args[1::2]):
        if not name.startswith("--"):
            raise SyntaxError("Command arg must start with '--', but '%s' is not!" % name)
        overrides[name[2:]] = value
    for name, text in overrides.items():
        config[name] = _parse_value(text)
    return config
