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


# ---------------------------------------------------------------------------- accumulation envelope
# The oracle proper ("exact") rounds the exactly summed value once.  MLX accumulates in float32 in an order it does not
# document; so does every GPU kernel.  The two modes below redo every accumulation of the model (linears, RMSNorm
# statistics, attention scores, softmax sum, P.V) with FLOAT32 accumulators in two different orders: partial sums of 32
# consecutive terms (exact, rounded to float32 -- one matrix-core step) combined sequentially ("f32_seq32": a K loop)
# or as a balanced tree ("f32_pairwise": split-K / tree reductions).  tests/golden/make_golden_wide.py runs both next
# to the exact oracle and commits the spread; THAT spread -- evidence that involves no HIP kernel -- is the tolerance
# of the production-width parity tests (tests/test_gpu_golden_wide.py).
ACCUM_MODES = ("exact", "f32_seq32", "f32_pairwise")
ACCUM = "exact"
CHUNK = 32
# A third variant (round 4), orthogonal to the order: SDPA_P16 rounds the softmax numerators exp(s - max) to the KV dtype
# before P.V while the denominator sums the unrounded float32 values -- what a 16-bit matrix-core attention kernel does
# when it feeds P to the MFMA as 16-bit operands (the reference's kernel keeps P in float32, App. A.4).  Used to attribute
# the device's excess over the two float32 orders in the model-dtype KV mode.
SDPA_P16 = False


def set_sdpa_p16(on: bool) -> None:
    global SDPA_P16
    SDPA_P16 = bool(on)


# A fourth variant (round 4, an experiment recorded in DESIGN 8): X_SPLIT2 rounds FLOAT32 activations entering a 16-bit or
# quantised linear to two 16-bit terms (hi = T(x), lo = T(x - hi): ~17 mantissa bits) when the call has more than 16 rows --
# what a two-pass matrix-core prefill GEMM would compute instead of the three-term (exact) split.
X_SPLIT2 = False


def set_x_split2(on: bool) -> None:
    global X_SPLIT2, SDPA_SPLIT2
    X_SPLIT2 = bool(on)
    SDPA_SPLIT2 = bool(on)


# ... and with it the float32 PREFILL attention (more than one query position) on two-term operands: q, K, P and V each as
# hi + lo of bfloat16 (what three matrix-core products per pair -- hi.hi + hi.lo + lo.hi -- see; the lo.lo product, 2^-16
# relative, is not modelled separately: rounding each operand to 16+ bits is the larger effect)
SDPA_SPLIT2 = False
SDPA_SPLIT2_DECODE = False     # ... the decode steps' attention too (an experiment: the device's split decode form lost on speed)


def split2(x: np.ndarray, dtype: str) -> np.ndarray:
    hi = round_to(x.astype(np.float32), dtype)
    lo = round_to((x.astype(np.float32) - hi).astype(np.float32), dtype)
    return (hi + lo).astype(np.float32)


def set_accum(mode: str) -> None:
    global ACCUM
    if mode not in ACCUM_MODES:
        raise ValueError(f"unknown accumulation mode {mode}")
    ACCUM = mode


class Accum:
    """Float32 running combination of partial sums handed over in order: sequential, or a balanced tree built with a
    binary-counter stack (equal-sized blocks are merged; leftovers smallest first) -- the same orders as
    oracle/c/accum_gemm.c."""

    def __init__(self, mode: str):
        self.seq = mode == "f32_seq32"
        self.acc = None
        self.stack = []
        self.n = 0

    def add(self, part: np.ndarray) -> None:
        part = np.asarray(part, dtype=np.float32)
        if self.seq:
            self.acc = part if self.acc is None else (self.acc + part).astype(np.float32)
        else:
            t = self.n
            while t & 1:
                part = (self.stack.pop() + part).astype(np.float32)
                t >>= 1
            self.stack.append(part)
        self.n += 1

    def result(self) -> np.ndarray:
        if self.seq:
            return self.acc
        acc = self.stack[-1]
        for blk in reversed(self.stack[:-1]):
            acc = (blk + acc).astype(np.float32)
        return acc


def sum_last_f32(x: np.ndarray, mode: str) -> np.ndarray:
    """Sum over the last axis with float32 accumulation in `mode` order (chunks of 32 summed exactly)."""
    a = Accum(mode)
    n = x.shape[-1]
    for k0 in range(0, n, CHUNK):
        a.add(x[..., k0:k0 + CHUNK].astype(np.float64).sum(axis=-1).astype(np.float32))
    return a.result()


_accum_lib = None


def _accum_gemm_lib():
    """oracle/c/accum_gemm.c (built by oracle/c/Makefile); None when it is not built -- the NumPy form below is the
    same arithmetic, only slow at production widths."""
    global _accum_lib
    if _accum_lib is None:
        import ctypes as C
        from pathlib import Path

        path = Path(__file__).resolve().parent / "_build" / "libaccum_gemm.so"
        if not path.exists():
            _accum_lib = False
        else:
            lib = C.CDLL(str(path))
            lib.accum_gemm_nt.restype = None
            lib.accum_gemm_nt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int]
            _accum_lib = lib
    return _accum_lib or None


_exact_lib = None


def _exact_gemm_lib():
    """oracle/c/exact_gemm.c (built by oracle/c/Makefile): exactly-summed matmuls straight from compact weights -- what
    makes a 32- / 40-block production-width model fit this container (tests/golden/make_golden_wide.py, the *_full_* cases)."""
    global _exact_lib
    if _exact_lib is None:
        import ctypes as C
        from pathlib import Path

        path = Path(__file__).resolve().parent / "_build" / "libexact_gemm.so"
        if not path.exists():
            _exact_lib = False
        else:
            lib = C.CDLL(str(path))
            lib.exact_gemm_nt_w16.restype = None
            lib.exact_gemm_nt_w16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_long]
            lib.exact_qgemm_nt.restype = None
            lib.exact_qgemm_nt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_long, C.c_long, C.c_long]
            lib.dequant_q_f32.restype = None
            lib.dequant_q_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long]
            lib.widen_w16_f32.restype = None
            lib.widen_w16_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
            _exact_lib = lib
    return _exact_lib or None


W16_KIND = {"bfloat16": 0, "float16": 1}


def widen_w16(bits: np.ndarray, dtype: str) -> np.ndarray:
    """16-bit patterns (uint16) of `dtype` -> float32 values."""
    bits = np.ascontiguousarray(bits, dtype=np.uint16)
    lib = _exact_gemm_lib()
    if lib is not None:
        out = np.empty(bits.shape, np.float32)
        lib.widen_w16_f32(bits.ctypes.data, W16_KIND[dtype], out.ctypes.data, bits.size)
        return out
    if dtype == "bfloat16":
        return bf16_bits_to_f32(bits)
    return bits.view(np.float16).astype(np.float32)


def matmul_nt_w16(x: np.ndarray, bits: np.ndarray, dtype: str) -> np.ndarray:
    """matmul_nt against a 16-bit weight matrix kept as its bit patterns (N, K): double products, double sum, one rounding."""
    lib = _exact_gemm_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if lib is None or x2.shape[0] > 32:              # many rows (a prefill): BLAS on a transient float64 copy is faster
        return matmul_nt(x, widen_w16(bits, dtype))
    bits = np.ascontiguousarray(bits, dtype=np.uint16)
    y = np.empty((x2.shape[0], bits.shape[0]), np.float32)
    lib.exact_gemm_nt_w16(x2.ctypes.data, bits.ctypes.data, W16_KIND[dtype], y.ctypes.data, x2.shape[0], bits.shape[0], K)
    return y.reshape(*x.shape[:-1], bits.shape[0])


def dequant_f32(packed, scales, biases, group: int, bits: int):
    """oracle/ref_quant.py:dequantize through C when it is built (None otherwise): the same float32 w_hat, two roundings."""
    lib = _exact_gemm_lib()
    if lib is None:
        return None
    packed = np.ascontiguousarray(packed, dtype=np.uint32)
    scales = np.ascontiguousarray(scales, dtype=np.float32)
    biases = np.ascontiguousarray(biases, dtype=np.float32)
    N, K = packed.shape[0], packed.shape[1] * 32 // bits
    out = np.empty((N, K), np.float32)
    lib.dequant_q_f32(packed.ctypes.data, scales.ctypes.data, biases.ctypes.data, bits, group, out.ctypes.data, N, K)
    return out


def matmul_nt_q(x: np.ndarray, packed, scales, biases, group: int, bits: int):
    """matmul_nt against MLX-affine quantised weights, w_hat formed on the fly (None when the C library is not built or the
    call has many rows: the caller then dequantises and uses matmul_nt)."""
    lib = _exact_gemm_lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if lib is None or x2.shape[0] > 32:
        return None
    packed = np.ascontiguousarray(packed, dtype=np.uint32)
    scales = np.ascontiguousarray(scales, dtype=np.float32)
    biases = np.ascontiguousarray(biases, dtype=np.float32)
    N = packed.shape[0]
    y = np.empty((x2.shape[0], N), np.float32)
    lib.exact_qgemm_nt(x2.ctypes.data, packed.ctypes.data, scales.ctypes.data, biases.ctypes.data, bits, group, y.ctypes.data,
                       x2.shape[0], N, K)
    return y.reshape(*x.shape[:-1], N)


def matmul_nt_f32(x: np.ndarray, w: np.ndarray, mode: str, use_c: bool = True) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    K = x.shape[-1]
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K)
    lib = _accum_gemm_lib() if use_c else None
    if lib is not None:
        y = np.empty((x2.shape[0], w.shape[0]), np.float32)
        lib.accum_gemm_nt(x2.ctypes.data, w.ctypes.data, y.ctypes.data, x2.shape[0], w.shape[0], K,
                          0 if mode == "f32_seq32" else 1)
        return y.reshape(*lead, w.shape[0])
    a = Accum(mode)
    for k0 in range(0, K, CHUNK):
        a.add((x2[:, k0:k0 + CHUNK].astype(np.float64) @ w[:, k0:k0 + CHUNK].astype(np.float64).T).astype(np.float32))
    return a.result().reshape(*lead, w.shape[0])


def matmul_nt(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """x (..., K) @ w(N, K).T accumulated in float64, returned as float32.

    MLX's matmul / quantized_matmul accumulate in fp32; the oracle uses the exactly
    rounded value (fp64 accumulate, one rounding to fp32) so that it is independent of
    any summation order.  (Under set_accum("f32_...") -- the accumulation envelope, see above -- float32 accumulators.)"""
    if ACCUM != "exact":
        return matmul_nt_f32(x, w, ACCUM)
    return (np.asarray(x, dtype=np.float64) @ np.asarray(w, dtype=np.float64).T).astype(np.float32)
