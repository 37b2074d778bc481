"""MLX affine quantisation on torch tensors (CPU or GPU) -- the ``nn.quantize(model, group_size,
bits)`` step of ``scripts/build_tiny_model.py:150-151`` / ``utils.quantize_model`` (utils.py:888-908),
producing the on-disk triple ``<p>.weight`` (uint32) / ``<p>.scales`` / ``<p>.biases``.

Format (mlx 0.25.2 ``mx.quantize``): per group of ``group_size`` consecutive
input features, ``w ~ scale * q + bias`` with ``q`` an unsigned ``bits``-bit code; 32/bits codes
per uint32, code j at bits [bits*j, bits*(j+1)).
"""
from __future__ import annotations

from typing import Tuple

import torch


def quantize(w: torch.Tensor, group_size: int = 64, bits: int = 4) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """w (N, K) float -> (packed int32-viewed-as-uint32 (N, K*bits/32), scales (N, K/g), biases (N, K/g)).
    scales/biases come back in ``w.dtype``."""
    if w.ndim != 2 or w.shape[1] % group_size != 0:
        raise ValueError("quantize expects (N, K) with K divisible by group_size")
    if bits not in (2, 4, 8):
        raise ValueError("bits must be 2, 4 or 8")
    dt = w.dtype
    n, k = w.shape
    g = w.to(torch.float32).reshape(n, k // group_size, group_size)
    n_bins = float((1 << bits) - 1)
    w_max = g.amax(dim=-1, keepdim=True)
    w_min = g.amin(dim=-1, keepdim=True)
    mask = w_min.abs() > w_max.abs()
    scales = torch.clamp((w_max - w_min) / n_bins, min=1e-7)
    scales = torch.where(mask, scales, -scales)
    edge = torch.where(mask, w_min, w_max)
    q0 = torch.round(edge / scales)
    scales = torch.where(q0 != 0, edge / q0, scales)
    biases = torch.where(q0 == 0, torch.zeros_like(edge), edge)
    q = torch.clamp(torch.round((g - biases) / scales), 0, n_bins).to(torch.int64)
    per = 32 // bits
    q = q.reshape(n, k // per, per)
    shifts = (torch.arange(per, device=w.device, dtype=torch.int64) * bits)
    packed = (q << shifts).sum(dim=-1)                       # < 2^32
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
    return packed, scales.reshape(n, -1).to(dt), biases.reshape(n, -1).to(dt)


def dequantize(packed: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor, group_size: int = 64,
               bits: int = 4) -> torch.Tensor:
    per = 32 // bits
    p = packed.to(torch.int64) & 0xFFFFFFFF
    shifts = (torch.arange(per, device=packed.device, dtype=torch.int64) * bits)
    q = ((p[..., None] >> shifts) & ((1 << bits) - 1)).reshape(packed.shape[0], -1).to(torch.float32)
    n, k = q.shape
    q = q.reshape(n, k // group_size, group_size)
    w = q * scales.to(torch.float32)[..., None] + biases.to(torch.float32)[..., None]
    return w.reshape(n, k)
