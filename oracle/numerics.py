"""Rounding helpers for the oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

All oracle tensors are float32 NumPy arrays whose *values* are representable in the
tensor's logical dtype ("float32" | "bfloat16" | "float16").  ``round_to`` is the only
place where a value is rounded; it is round-to-nearest-even, like an MLX ``astype``.
"""
from __future__ import annotations

import numpy as np

DTYPES = ("float32", "bfloat16", "float16")


def round_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16 -> float32 (NaN kept NaN).  uint32 arithmetic: the add can only wrap for
    NaN bit patterns, which are restored afterwards."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = (u >> np.uint32(16)) & np.uint32(1)
    r += np.uint32(0x7FFF)
    r += u
    r &= np.uint32(0xFFFF0000)
    out = r.view(np.float32).reshape(x.shape)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    if nan.any():
        out[nan.reshape(x.shape)] = x[nan.reshape(x.shape)]
    return out


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    if dtype == "float32":
        return x
    if dtype == "bfloat16":
        return round_bf16(x)
    if dtype == "float16":
        with np.errstate(over="ignore"):
            return x.astype(np.float16).astype(np.float32)
    raise ValueError(f"unknown dtype {dtype}")


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    return (round_bf16(x).view(np.uint32) >> 16).astype(np.uint16)


def matmul_nt(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """x (..., K) @ w(N, K).T accumulated in float64, returned as float32.

    MLX's matmul / quantized_matmul accumulate in fp32; the oracle uses the exactly
    rounded value (fp64 accumulate, one rounding to fp32) so that it is independent of
    any summation order."""
    return (np.asarray(x, dtype=np.float64) @ np.asarray(w, dtype=np.float64).T).astype(np.float32)
