"""Sampling, restated (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows ``mlx_parallm/utils.py:345-364`` (``sample`` closure of ``generate_step``) and
``mlx_parallm/sample_utils.py:3-38`` (``top_p_sampling``).

``mx.random.categorical`` draws from MLX's threefry stream, which cannot be reproduced
offline (App. A.5).  The build therefore DEFINES the draw as an inverse-CDF pick over the
candidates in the reference's own order (descending probability, ties by ascending token
id) driven by ONE uniform ``u`` in [0,1) per row, supplied by the caller; the *distribution*
is the reference's (softmax(logits/T) restricted to the top-p set and renormalised).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def log_softmax(logits: np.ndarray) -> np.ndarray:
    x = logits.astype(np.float64)
    m = x.max(axis=-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True)))


def descending_order(p: np.ndarray) -> np.ndarray:
    """argsort(-p) (sample_utils.py:18) with ties broken by ascending index (stable)."""
    return np.argsort(-p, kind="stable")


def top_p_candidates(logits_row: np.ndarray, top_p: float, temperature: float):
    """Returns (sorted_token_ids_kept, normalised_probs_kept) per sample_utils.py:15-31.

    ``mask = cumulative_probs <= top_p`` keeps the tokens whose INCLUSIVE cumulative sum is
    <= top_p (the crossing token is excluded).  If the set is empty (top token alone exceeds
    top_p) the reference divides 0/0 (quirk Q5); the build keeps the top-1 token instead."""
    x = logits_row.astype(np.float64) / float(temperature)
    x = x - x.max()
    p = np.exp(x)
    p = p / p.sum()
    order = descending_order(p)
    sp = p[order]
    c = np.cumsum(sp)
    keep = c <= float(top_p)
    n = int(keep.sum())
    if n == 0:
        n = 1
    sp = sp[:n]
    return order[:n], sp / sp.sum()


def inverse_cdf_pick(ids: np.ndarray, probs: np.ndarray, u: float) -> int:
    """Smallest j with cumsum(probs)[j] > u * 1.0 (clamped to the last candidate)."""
    c = np.cumsum(probs)
    j = int(np.searchsorted(c, float(u), side="right"))
    return int(ids[min(j, len(ids) - 1)])


def sample(logits: np.ndarray, temp: float = 0.0, top_p: float = 1.0,
           logit_bias: Optional[Dict[int, float]] = None,
           uniforms: Optional[np.ndarray] = None, logprobs_at_temperature: bool = False):
    """utils.py:345-364.  logits (B, V) float32.  Returns dict with
    tokens (B,1) int64, probs (B,1) = softmax(logits)[0, tokens] (row-0 quirk Q6),
    logprobs (B,) = log_softmax(logits)[b, token_b]."""
    logits = np.array(logits, dtype=np.float32, copy=True)
    B, V = logits.shape
    if logit_bias:                                                     # utils.py:346-349
        for k, v in logit_bias.items():
            logits[:, int(k)] += np.float32(v)
    lsm = log_softmax(logits)                                          # utils.py:350
    if temp == 0:
        tokens = np.argmax(logits, axis=-1)                            # lowest index among ties
    else:
        assert uniforms is not None and len(uniforms) == B
        tokens = np.zeros(B, dtype=np.int64)
        for b in range(B):
            tp = top_p if (0 < top_p < 1.0) else 1.0                   # utils.py:355-361
            if tp >= 1.0:
                # categorical(logits / temp): every token is a candidate
                x = logits[b].astype(np.float64) / float(temp)
                x = x - x.max()
                p = np.exp(x)
                p = p / p.sum()
                order = descending_order(p)
                ids, pr = order, p[order]
            else:
                ids, pr = top_p_candidates(logits[b], tp, temp)
            tokens[b] = inverse_cdf_pick(ids, pr, float(uniforms[b]))
    tokens = tokens.reshape(B, 1)
    probs = np.exp(lsm[0, tokens[:, 0]]).reshape(B, 1).astype(np.float32)   # utils.py:363 (Q6)
    lsm_out = lsm
    if logprobs_at_temperature and temp > 0:       # server/main.py:571-584: softmax(logits * (1 / T))
        lsm_out = log_softmax(logits * np.float32(1.0 / temp))
    logprobs = lsm_out[np.arange(B), tokens[:, 0]].astype(np.float32)
    return {"tokens": tokens, "probs": probs, "logprobs": logprobs, "log_softmax": lsm_out}
