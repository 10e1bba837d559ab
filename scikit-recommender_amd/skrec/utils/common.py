"""Small shared helpers (reference: skrec/utils/common.py)."""
import os

import numpy as np
import scipy.sparse as sp

__all__ = ["normalize_adj_matrix", "PostInitMeta", "make_sure_dirs"]


def normalize_adj_matrix(sp_mat, norm_method="left"):
    """Degree-normalise a sparse adjacency matrix (reference: utils/common.py:11-40).

    ``left``      -> D^-1 A          ``symmetric`` -> D^-1/2 A D^-1/2
    Rows of degree zero get a zero scale (the reference replaces inf by 0, common.py:27,32).
    """
    deg = np.asarray(sp_mat.sum(axis=1)).reshape(-1)
    if norm_method == "left":
        with np.errstate(divide="ignore"):
            scale = np.power(deg, -1)
        scale[np.isinf(scale)] = 0.0
        return sp.diags(scale).dot(sp_mat)
    if norm_method == "symmetric":
        with np.errstate(divide="ignore"):
            scale = np.power(deg, -0.5)
        scale[np.isinf(scale)] = 0.0
        d = sp.diags(scale)
        return d.dot(sp_mat).dot(d)
    raise ValueError(f"'{norm_method}' is an invalid normalization method.")


class PostInitMeta(type):
    """Calls ``obj.__post_init__()`` once the whole ``__init__`` chain has run (common.py:43-48)."""

    def __call__(cls, *args, **kwargs):
        obj = super().__call__(*args, **kwargs)
        hook = getattr(obj, "__post_init__", None)
        if hook is not None:
            hook()
        return obj


def make_sure_dirs(dir_path):
    os.makedirs(dir_path, exist_ok=True)
