"""Helpers for the -m gpu tests: move oracle-side NumPy data to the device and call the
kernel-level C ABI (include/mi355_ops.h) through ctypes."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from mlx_parallm_amd import _lib as L

TDT = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}
MIDT = {"float32": L.MI_F32, "bfloat16": L.MI_BF16, "float16": L.MI_F16}
ULP = {"float32": 2.0 ** -20, "bfloat16": 2.0 ** -7, "float16": 2.0 ** -10}     # 2 ulp of the dtype


def dev(a: np.ndarray, dtype: str = "float32") -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TDT[dtype]).cuda().contiguous()


def dev_u32(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda().contiguous()


def dev_i32(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda().contiguous()


def host(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def wk_code(kind: str) -> int:
    return L.WK[kind]


def op_linear(kind: str, N: int, K: int, w, scales=None, biases=None, group: int = 64) -> L.OpLinear:
    ol = L.OpLinear()
    ol.wk, ol.N, ol.K, ol.group = wk_code(kind), N, K, group
    ol.w, ol.scales, ol.biases = w.data_ptr(), (scales.data_ptr() if scales is not None else 0), \
        (biases.data_ptr() if biases is not None else 0)
    return ol


def to_tiled(ol: L.OpLinear, keep: list) -> bool:
    """Repack a row-major op_linear into the tile-major layout in place (the tiled buffer is appended
    to `keep`).  Returns False when the matrix is not eligible (then it stays row-major)."""
    nbytes = int(L.lib().mi_op_tiled_bytes(C.byref(ol)))
    if nbytes == 0:
        return False
    dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_repack_tiled(C.byref(ol), C.c_void_p(dst.data_ptr())))
    keep.append(dst)
    ol.w, ol.scales, ol.biases, ol.layout = dst.data_ptr(), 0, 0, 1
    return True


def gemv(ol: L.OpLinear, x, M, act, *, rnd=0, pro=0, norm_w=None, eps=0.0, epi=0, out=None, ldo=0, resid=None,
         pair_offset=0, force_generic=0, ldx=None):
    a = L.OpGemvArgs()
    a.x = x.data_ptr(); a.ldx = ldx if ldx is not None else x.shape[-1]; a.M = M
    a.act = MIDT[act]; a.rnd = rnd; a.pro = pro; a.epi = epi
    a.norm_w = norm_w.data_ptr() if norm_w is not None else 0
    a.eps = eps; a.ldo = ldo
    a.out = out.data_ptr() if out is not None else 0
    a.resid = resid.data_ptr() if resid is not None else 0
    a.pair_offset = pair_offset; a.force_generic = force_generic
    torch.cuda.synchronize()
    used_mfma = L.lib().mi_op_gemv_uses_mfma(C.byref(ol), C.byref(a))
    L.check(L.lib().mi_op_gemv(C.byref(ol), C.byref(a)))
    return bool(used_mfma)


def gemv_args(x, M, act, *, pro=0, norm_w=None, eps=0.0, epi=0, out=None, ldo=0, resid=None, pair_offset=0, ldx=None) -> L.OpGemvArgs:
    a = L.OpGemvArgs()
    a.x = x.data_ptr(); a.ldx = ldx if ldx is not None else x.shape[-1]; a.M = M
    a.act = MIDT[act]; a.rnd = 0; a.pro = pro; a.epi = epi
    a.norm_w = norm_w.data_ptr() if norm_w is not None else 0
    a.eps = eps; a.ldo = ldo
    a.out = out.data_ptr() if out is not None else 0
    a.resid = resid.data_ptr() if resid is not None else 0
    a.pair_offset = pair_offset; a.force_generic = 0
    return a


def attn_shape(B, L_, Hq, Hkv, D, act, kv, rnd, cap) -> L.OpAttnShape:
    s = L.OpAttnShape()
    s.B, s.L, s.Hq, s.Hkv, s.D, s.act, s.kv, s.rnd, s.cap = B, L_, Hq, Hkv, D, MIDT[act], MIDT[kv], rnd, cap
    return s


def close_frac(got: np.ndarray, want: np.ndarray, dtype: str, atol: float = 0.0) -> float:
    """Fraction of elements further than 2 ulp(dtype) (relative) + atol from the oracle."""
    tol = ULP[dtype] * np.maximum(np.abs(want), np.abs(got)) + atol
    return float(np.mean(np.abs(got - want) > tol))


def q4_force(mt=0, tw=0, kw=0, ksplit=0, ns=0) -> int:
    """`ksplit` argument of gemm_skinny() that forces gemm_q4.hip's plan (include/mi355_ops.h): row tiles per workgroup, tile
    units x K lanes (= compute waves), K slices over workgroups, staging waves; 0 = the cost model's choice."""
    return -(mt | tw << 3 | kw << 7 | ksplit << 11 | ns << 15)


def gemm_skinny(ol: L.OpLinear, x, M, act, *, epi=0, out=None, ldo=0, resid=None, pair_offset=0, ksplit=0, iters=0, ldx=None,
                rnd=0, norm_w=None, eps=0.0):
    """gemm_skinny.hip / gemm_q4.hip on its own (17..128 rows) -> (ksplit used, mean launch ms or None).  norm_w: RMSNorm in
    front (int4 weights above 16 rows only: gemm_q4.hip's preparation pass applies it)."""
    a = L.OpGemvArgs()
    a.x = x.data_ptr(); a.ldx = ldx if ldx is not None else x.shape[-1]; a.M = M
    a.act = MIDT[act]; a.rnd = rnd; a.pro = 1 if norm_w is not None else 0; a.epi = epi
    a.norm_w = norm_w.data_ptr() if norm_w is not None else 0; a.eps = eps; a.ldo = ldo
    a.out = out.data_ptr() if out is not None else 0
    a.resid = resid.data_ptr() if resid is not None else 0
    a.pair_offset = pair_offset; a.force_generic = 0
    torch.cuda.synchronize()
    used, ms = C.c_int(0), C.c_float(0.0)
    L.check(L.lib().mi_op_gemm_skinny(C.byref(ol), C.byref(a), int(ksplit), C.byref(used), int(iters), C.byref(ms)))
    return used.value, (ms.value if iters >= 1 else None)


def gemm_prefill(ol: L.OpLinear, x, M, act, *, epi=0, out=None, ldo=0, resid=None, pair_offset=0, iters=0, ldx=None):
    """gemm_prefill.hip on its own (the tile GEMM of the prefill call) -> mean launch ms or None."""
    a = gemv_args(x, M, act, epi=epi, out=out, ldo=ldo, resid=resid, pair_offset=pair_offset, ldx=ldx)
    torch.cuda.synchronize()
    ms = C.c_float(0.0)
    L.check(L.lib().mi_op_gemm_prefill(C.byref(ol), C.byref(a), int(iters), C.byref(ms)))
    return ms.value if iters >= 1 else None
