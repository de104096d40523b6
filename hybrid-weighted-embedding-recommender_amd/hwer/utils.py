"""Host-side helpers of the hot path (reference: hwer/utils.py).  Tiny numpy / pure-Python functions that the
reference keeps on the host too (restated, not accelerated)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np


class NodeNotFoundException(Exception):
    """Raised for an anchor that was never registered (reference: hwer/utils.py:326-327)."""


def _apply_each(fn):
    # the reference wraps these two helpers so that several arrays can be passed at once (hwer/utils.py:286-293,312-313)
    def wrapper(*arrays, **kw):
        out = [fn(a, **kw) for a in arrays]
        return out[0] if len(out) == 1 else out
    return wrapper


@_apply_each
def unit_length(a, axis=0):
    """a / ||a||_2 along `axis`, no epsilon: a zero vector becomes NaN (reference: hwer/utils.py:43-44)."""
    a = np.asarray(a)
    with np.errstate(invalid="ignore", divide="ignore"):
        return a / np.expand_dims(np.linalg.norm(a, axis=axis), axis=axis)


@_apply_each
def unit_length_violations(a, axis=0, epsilon=1e-4):
    """(violations, mean |len-1|, too long, too short) — reference: hwer/utils.py:51-57."""
    lengths = np.expand_dims(np.linalg.norm(np.asarray(a), axis=axis), axis=axis)
    too_long = np.sum(lengths > 1 + epsilon)
    too_short = np.sum(lengths < 1 - epsilon)
    return too_long + too_short, np.mean(np.abs(lengths - 1)), too_long, too_short


def reciprocal_rank(y_true, y_pred):
    """1/rank of the first hit, 0.0 without one (reference: hwer/utils.py:71-78)."""
    wanted = set(y_true)
    for pos, e in enumerate(y_pred, start=1):
        if e in wanted:
            return 1.0 / pos
    return 0.0


def ndcg(y_true: Dict, y_pred: List):
    """DCG(pred) / (IDCG(truth cut to len(pred)) + 1e-8) with gains 2^rel - 1 (reference: hwer/utils.py:101-107)."""
    discounts = np.log2(np.arange(len(y_pred)) + 2)
    ideal = sorted(y_true.values(), reverse=True)[:len(y_pred)]
    idcg = np.sum((np.power(2, ideal) - 1) / np.log2(np.arange(len(ideal)) + 2))
    gains = [y_true.get(e, 0) for e in y_pred]
    dcg = np.sum((np.power(2, gains) - 1) / discounts)
    return dcg / (idcg + 1e-8)


def binary_ndcg(y_true: Dict, y_pred: List):
    """reference: hwer/utils.py:110-111"""
    return ndcg(dict.fromkeys(y_true, 1), y_pred)


def binary_ndcg_v2(y_true: List, y_pred: List):
    """reference: hwer/utils.py:114-115"""
    return ndcg(dict.fromkeys(y_true, 1), y_pred)


def recall(y_true, y_pred):
    """hits / max(min(|pred|, |true|), 1) (reference: hwer/utils.py:118-121)."""
    hits = sum(1 for e in y_pred if e in y_true)
    return hits / max(min(len(y_pred), len(y_true)), 1.0)
