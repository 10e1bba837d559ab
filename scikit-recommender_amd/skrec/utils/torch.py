"""torch helpers kept for API parity with the reference's utils/torch.py (used by foreign models and
as the CPU-side definition of the maths; the in-scope models run these ops inside HIP kernels)."""
from collections import OrderedDict
from functools import partial

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F
from torch import Tensor, nn

__all__ = ["inner_product", "bpr_loss", "l2_loss", "sp_mat_to_sp_tensor", "get_initializer",
           "sigmoid_cross_entropy", "square_loss", "euclidean_distance", "l2_distance"]


def inner_product(a: Tensor, b: Tensor, dim: int = -1) -> Tensor:
    return torch.sum(a * b, dim=dim)


def euclidean_distance(a: Tensor, b: Tensor, dim: int = -1) -> Tensor:
    return torch.norm(a - b, p=None, dim=dim)


l2_distance = euclidean_distance


def bpr_loss(y_pos: Tensor, y_neg: Tensor) -> Tensor:
    return -F.logsigmoid(y_pos - y_neg)


def l2_loss(*weights):
    return 0.5 * sum(torch.sum(torch.pow(w, 2)) for w in weights)


def sp_mat_to_sp_tensor(sp_mat: sp.spmatrix) -> Tensor:
    coo = sp_mat.tocoo().astype(np.float32)
    idx = torch.from_numpy(np.asarray([coo.row, coo.col]))
    return torch.sparse_coo_tensor(idx, coo.data, coo.shape).coalesce()


def sigmoid_cross_entropy(y_pre, y_true):
    return F.binary_cross_entropy_with_logits(input=y_pre, target=y_true, reduction="none")


def square_loss(y_pre, y_true):
    if isinstance(y_true, (float, int)):
        y_true = y_pre.new_full(y_pre.size(), y_true)
    return F.mse_loss(input=y_pre, target=y_true, reduction="none")


# name -> in-place initialiser, same table as the reference (utils/torch.py:88-104)
_initializers = OrderedDict([
    ("normal", partial(nn.init.normal_, mean=0.0, std=0.01)),
    ("uniform", partial(nn.init.uniform_, a=-0.05, b=0.05)),
    ("he_normal", nn.init.kaiming_normal_),
    ("he_uniform", nn.init.kaiming_uniform_),
    ("xavier_normal", nn.init.xavier_normal_),
    ("xavier_uniform", nn.init.xavier_uniform_),
    ("zeros", nn.init.zeros_),
    ("ones", nn.init.ones_),
])


def get_initializer(init_method: str):
    if init_method not in _initializers:
        names = ", ".join(_initializers)
        raise ValueError(f"'init_method' is invalid, and must be one of '{names}'")
    return _initializers[init_method]
