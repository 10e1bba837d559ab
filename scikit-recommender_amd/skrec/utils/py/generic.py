"""Generic helpers kept for API parity (reference: utils/py/generic.py)."""
import hashlib
import re
import unicodedata
from collections import OrderedDict

import numpy as np

__all__ = ["OrderedDefaultDict", "pad_sequences", "md5sum", "slugify"]


class OrderedDefaultDict(OrderedDict):
    def __init__(self, default_factory=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.default_factory = default_factory

    def __missing__(self, key):
        if self.default_factory is None:
            raise KeyError(key)
        self[key] = value = self.default_factory()
        return value


def pad_sequences(sequences, value=0, max_len=None, padding="post", truncating="post", dtype=np.int32):
    """Pad / truncate a list of sequences to a [n, max_len] array (generic.py:40-90)."""
    if max_len is None:
        max_len = max(len(s) for s in sequences)
    out = np.full([len(sequences), max_len], value, dtype=dtype)
    for row, seq in enumerate(sequences):
        if len(seq) == 0:
            continue
        if truncating == "pre":
            piece = seq[-max_len:]
        elif truncating == "post":
            piece = seq[:max_len]
        else:
            raise ValueError(f"Truncating type '{truncating}' not understood")
        if padding == "post":
            out[row, :len(piece)] = piece
        elif padding == "pre":
            out[row, max_len - len(piece):] = piece
        else:
            raise ValueError(f"Padding type '{padding}' not understood")
    return out


def md5sum(*args):
    digests = []
    for filename in args:
        with open(filename, "rb") as fin:
            digests.append(hashlib.md5(fin.read()).hexdigest())
    return digests[0] if len(args) == 1 else digests


def slugify(text, max_length=255):
    """File-name safe version of ``text`` (generic.py:110-128)."""
    text = unicodedata.normalize("NFKD", str(text)).encode("ascii", "ignore").decode("ascii")
    text = re.sub(r"[^\w\s\-.=]", "", text).strip()
    text = re.sub(r"[-\s]+", "-", text)
    return text[:max_length]
