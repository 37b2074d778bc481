"""MLX affine quantisation, restated (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Reference call sites: ``nn.quantize(model, group_size, bits)`` in
``scripts/build_tiny_model.py:150-151`` and ``mlx_parallm/utils.py:679-690`` (load-side
``class_predicate`` on ``<p>.scales``); on-disk keys ``<p>.weight`` (uint32), ``<p>.scales``,
``<p>.biases`` (``utils.py:888-908``).  The arithmetic is ``mx.quantize`` / ``mx.dequantize``
of the un-vendored ``mlx==0.25.2`` wheel; restated from its published behaviour
 -- unverifiable offline.
"""
from __future__ import annotations

import numpy as np

from .numerics import round_to


def quantize(w: np.ndarray, group_size: int = 64, bits: int = 4, dtype: str = "float32"):
    """w (N, K) float32 -> (packed uint32 (N, K*bits/32), scales (N, K/g), biases (N, K/g)).

    scales / biases are returned as float32 arrays holding values rounded to ``dtype``
    (the weight's dtype, as ``mx.quantize`` returns them)."""
    w = np.asarray(w, dtype=np.float32)
    n, k = w.shape
    assert k % group_size == 0, "last dim must be divisible by group_size"
    assert bits in (2, 4, 8)
    n_bins = float((1 << bits) - 1)
    eps = np.float32(1e-7)
    g = w.reshape(n, k // group_size, group_size)
    w_max = g.max(axis=-1, keepdims=True)
    w_min = g.min(axis=-1, keepdims=True)
    mask = np.abs(w_min) > np.abs(w_max)
    scales = np.maximum((w_max - w_min) / np.float32(n_bins), eps).astype(np.float32)
    scales = np.where(mask, scales, -scales)
    edge = np.where(mask, w_min, w_max)
    q0 = np.rint(edge / scales)
    with np.errstate(divide="ignore", invalid="ignore"):
        scales = np.where(q0 != 0, edge / q0, scales).astype(np.float32)
    biases = np.where(q0 == 0, np.float32(0), edge).astype(np.float32)
    # the integer codes are computed with the *stored* (dtype-rounded) scales/biases? No:
    # mx.quantize computes codes from the fp32 scales/biases of the input dtype; for a
    # float32 input they coincide.  We quantise in the input precision, then round.
    q = np.clip(np.rint((g - biases) / scales), 0, n_bins).astype(np.uint32)
    per_word = 32 // bits
    q = q.reshape(n, k // per_word, per_word)
    shifts = (np.arange(per_word, dtype=np.uint32) * bits).astype(np.uint32)
    packed = np.bitwise_or.reduce(q << shifts, axis=-1).astype(np.uint32)
    return (
        packed,
        round_to(scales.reshape(n, -1), dtype),
        round_to(biases.reshape(n, -1), dtype),
    )


def unpack(packed: np.ndarray, bits: int) -> np.ndarray:
    """(N, K*bits/32) uint32 -> (N, K) integer codes, element j at bits [bits*j, bits*(j+1))."""
    per_word = 32 // bits
    shifts = (np.arange(per_word, dtype=np.uint32) * bits).astype(np.uint32)
    q = (packed[..., None] >> shifts) & np.uint32((1 << bits) - 1)
    return q.reshape(packed.shape[0], -1)


def dequantize(packed, scales, biases, group_size: int = 64, bits: int = 4) -> np.ndarray:
    """w_hat = scale * q + bias, in float32 (no rounding of w_hat)."""
    q = unpack(packed, bits).astype(np.float32)
    n, k = q.shape
    q = q.reshape(n, k // group_size, group_size)
    w = q * scales[..., None].astype(np.float32) + biases[..., None].astype(np.float32)
    return w.reshape(n, k).astype(np.float32)
