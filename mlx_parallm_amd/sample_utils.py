"""``top_p_sampling`` (mirror of ``mlx_parallm/sample_utils.py:3-38``) on the MI355X sampler kernel.

Inside ``generate_step`` sampling is fused behind the lm_head on the device; this standalone
entry point runs the same kernel on a logits array the caller supplies.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L


def _sample_device(logits, temperature: float, top_p: float, uniforms=None, top_logprobs: int = 0):
    import torch

    t = torch.as_tensor(np.asarray(logits, dtype=np.float32)).to("cuda").contiguous().clone()
    if t.ndim == 1:
        t = t[None]
    B, V = t.shape
    if uniforms is None:
        uniforms = np.random.default_rng().random(B)
    u = torch.as_tensor(np.asarray(uniforms, dtype=np.float32)).to("cuda").contiguous()
    toks = torch.empty(B, dtype=torch.int32, device="cuda")
    lp = torch.empty(B, dtype=torch.float32, device="cuda")
    p0 = torch.empty(B, dtype=torch.float32, device="cuda")
    k = max(int(top_logprobs), 1)
    tk_i = torch.empty((B, k), dtype=torch.int32, device="cuda")
    tk_l = torch.empty((B, k), dtype=torch.float32, device="cuda")
    stats = torch.empty((B, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_sample(C.c_void_p(t.data_ptr()), B, V, float(temperature), float(top_p),
                                 C.c_void_p(u.data_ptr()), int(top_logprobs), C.c_void_p(toks.data_ptr()),
                                 C.c_void_p(lp.data_ptr()), C.c_void_p(p0.data_ptr()), C.c_void_p(tk_i.data_ptr()),
                                 C.c_void_p(tk_l.data_ptr()), C.c_void_p(stats.data_ptr())))
    out = {"tokens": toks.cpu().numpy(), "logprobs": lp.cpu().numpy(), "probs_row0": p0.cpu().numpy()}
    if top_logprobs > 0:
        out["top_ids"], out["top_logprobs"] = tk_i.cpu().numpy(), tk_l.cpu().numpy()
    return out


def top_p_sampling(logits, top_p: float, temperature: float, axis: int = -1, *, uniforms: Optional[np.ndarray] = None):
    """softmax(logits / temperature) -> keep the descending-probability prefix whose cumulative
    probability is <= top_p -> renormalise -> draw.  Returns tokens (B, 1) like the reference."""
    if axis not in (-1, 1):
        raise ValueError("top_p_sampling works over the last axis")
    res = _sample_device(logits, temperature, top_p, uniforms)
    return res["tokens"].reshape(-1, 1)
