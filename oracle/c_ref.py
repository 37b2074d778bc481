"""ctypes access to oracle/c/ref_decode.c (TEST INFRASTRUCTURE -- see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libref_decode.so"


class RefLayer(C.Structure):
    _fields_ = [("H", C.c_int), ("Hq", C.c_int), ("Hkv", C.c_int), ("D", C.c_int), ("I", C.c_int),
                ("eps", C.c_float), ("rope_theta", C.c_float), ("qk_norm", C.c_int)] + \
               [(n, C.c_void_p) for n in ("wq", "wk", "wv", "wo", "wg", "wu", "wd", "in_norm", "post_norm", "q_norm", "k_norm")]


def build() -> Path:
    res = subprocess.run(["make", "-C", str(HERE / "c")], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building oracle/c failed:\n" + res.stdout + res.stderr)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        _lib = C.CDLL(str(LIB))
        _lib.ref_layer_step.restype = None
        _lib.ref_layer_step.argtypes = [C.POINTER(RefLayer), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.ref_layer_scratch_floats.restype = C.c_size_t
        _lib.ref_layer_scratch_floats.argtypes = [C.POINTER(RefLayer), C.c_int]
        _lib.ref_head.restype = None
        _lib.ref_head.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ref_bench_decode.restype = C.c_double
        _lib.ref_bench_decode.argtypes = [C.c_int] * 10 + [C.POINTER(C.c_int), C.POINTER(C.c_double)]
    return _lib


def bench_decode(H, Hq, Hkv, D, I, V, nl, B, ctx, steps):
    """-> (total seconds for `steps` decode steps over `nl` blocks + lm_head, of which lm_head seconds,
    threads used)."""
    nth = C.c_int(0)
    th = C.c_double(0.0)
    sec = lib().ref_bench_decode(H, Hq, Hkv, D, I, V, nl, B, ctx, steps, C.byref(nth), C.byref(th))
    return float(sec), float(th.value), int(nth.value)


def bf16_bits(a: np.ndarray) -> np.ndarray:
    from .numerics import f32_to_bf16_bits

    return np.ascontiguousarray(f32_to_bf16_bits(np.asarray(a, np.float32)))
