"""ctypes binding of libmi355_decode.so (include/mi355_decode.h, include/mi355_ops.h).

There is no fallback: if the shared library is missing or cannot be loaded, importing the
symbols raises.  ``build()`` compiles it in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(os.environ.get("MLX_PARALLM_AMD_LIB") or (CSRC / "libmi355_decode.so"))   # (env: A/B another build)

# dtype / enum constants (mirror of the headers)
MI_F32, MI_BF16, MI_F16, MI_U32 = 0, 1, 2, 3
MI_KV_MODEL = -1
MI_ARCH_LLAMA, MI_ARCH_QWEN3 = 0, 1
MI_MAX_TOP_LOGPROBS = 20
MI_ABI_VERSION = 2
WK = {"f32": 0, "bf16": 1, "f16": 2, "q4_f32": 3, "q4_bf16": 4, "q4_f16": 5, "q8_f32": 6, "q8_bf16": 7, "q8_f16": 8}
RND_NONE, RND_BF16, RND_F16 = 0, 1, 2
PRO_NONE, PRO_NORM = 0, 1
EPI_STORE, EPI_STORE_F32, EPI_RESID, EPI_SWIGLU = 0, 1, 2, 3

_ERRORS = {-1: ValueError, -2: FileNotFoundError, -3: NotImplementedError, -4: RuntimeError}


class ModelDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("arch", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
        ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("intermediate_size", C.c_int32),
        ("vocab_size", C.c_int32), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
        ("rope_scale", C.c_float), ("tie_word_embeddings", C.c_int32), ("act_dtype", C.c_int32),
        ("quant_bits", C.c_int32), ("quant_group_size", C.c_int32), ("max_positions", C.c_int32),
    ]


class SampleParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("temperature", C.c_float), ("top_p", C.c_float), ("n_logit_bias", C.c_int32),
        ("logit_bias_ids", C.POINTER(C.c_int32)), ("logit_bias_values", C.POINTER(C.c_float)),
        ("uniforms", C.POINTER(C.c_float)), ("seed", C.c_uint64), ("top_logprobs", C.c_int32),
        ("logprobs_at_temperature", C.c_int32),
        ("row_temperature", C.POINTER(C.c_float)), ("row_top_p", C.POINTER(C.c_float)),
        ("stream_position", C.c_int64),
    ]


class OpLinear(C.Structure):
    _fields_ = [("wk", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("group", C.c_int32),
                ("w", C.c_void_p), ("scales", C.c_void_p), ("biases", C.c_void_p), ("layout", C.c_int32)]


class OpGemvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int32), ("M", C.c_int32), ("act", C.c_int32), ("rnd", C.c_int32),
                ("pro", C.c_int32), ("epi", C.c_int32), ("norm_w", C.c_void_p), ("eps", C.c_float),
                ("ldo", C.c_int32), ("out", C.c_void_p), ("resid", C.c_void_p), ("pair_offset", C.c_int32),
                ("force_generic", C.c_int32)]


class OpAttnShape(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32),
                ("act", C.c_int32), ("kv", C.c_int32), ("rnd", C.c_int32), ("cap", C.c_int32)]


# name -> (restype, argtypes); every symbol the two headers declare
_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_F32P = C.POINTER(C.c_float)
SIGNATURES = {
    # mi355_decode.h
    "mi_engine_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_P)]),
    "mi_engine_destroy": (None, [_P]),
    "mi_engine_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int]),
    "mi_engine_set_lora": (C.c_int, [_P, C.c_int, C.c_char_p, _P, _P, C.c_int, C.c_float, C.c_int, C.c_int]),
    "mi_engine_finalize": (C.c_int, [_P]),
    "mi_kv_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "mi_kv_create_paged": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "mi_kv_prefix_attach": (C.c_int, [_P, C.c_int, _I32P, C.c_int, C.POINTER(C.c_int)]),
    "mi_kv_prefix_publish": (C.c_int, [_P, C.c_int, _I32P, C.c_int]),
    "mi_kv_prefix_clear": (C.c_int, [_P]),
    "mi_kv_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int]),
    "mi_kv_destroy": (None, [_P]),
    "mi_kv_reset": (C.c_int, [_P, C.c_int]),
    "mi_kv_reserve": (C.c_int, [_P, C.c_int]),
    "mi_kv_offsets": (C.c_int, [_P, _I32P]),
    "mi_kv_capacity": (C.c_int, [_P]),
    "mi_forward": (C.c_int, [_P, _P, _I32P, C.c_int, C.c_int, _F32P, C.c_int]),
    "mi_decode_sample": (C.c_int, [_P, _P, _I32P, C.c_int, C.c_int, C.POINTER(SampleParams), _I32P, _F32P, _F32P, _I32P, _F32P]),
    "mi_score_tokens": (C.c_int, [_P, _P, _I32P, _I32P, C.c_int, C.c_int, C.POINTER(SampleParams), _F32P, _I32P, _F32P]),
    "mi_step_enqueue": (C.c_int, [_P, _P, _I32P, C.c_int, C.c_int, C.POINTER(SampleParams), C.POINTER(C.c_int64)]),
    "mi_step_enqueue_rows": (C.c_int, [_P, _P, _I32P, C.c_int, _I32P, C.c_int, C.POINTER(SampleParams), C.POINTER(C.c_int64)]),
    "mi_step_enqueue_mixed": (C.c_int, [_P, _P, _I32P, _I32P, _I32P, C.c_int, _I32P, C.POINTER(SampleParams), C.POINTER(C.c_int64)]),
    "mi_kv_reset_row": (C.c_int, [_P, C.c_int]),
    "mi_step_wait": (C.c_int, [_P, C.c_int64, _I32P, _F32P, _F32P, _I32P, _F32P]),
    "mi_profile_select": (C.c_int, [_P, C.c_char_p]),
    "mi_profile_read": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "mi_engine_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "mi_engine_sync": (C.c_int, [_P]),
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_char_p, []),
    "mi_abi_version": (C.c_int, []),
    # mi355_ops.h
    "mi_op_gemv": (C.c_int, [C.POINTER(OpLinear), C.POINTER(OpGemvArgs)]),
    "mi_op_gemv_uses_mfma": (C.c_int, [C.POINTER(OpLinear), C.POINTER(OpGemvArgs)]),
    "mi_op_gemv_bench": (C.c_int, [C.POINTER(OpLinear), C.POINTER(OpGemvArgs), C.c_int, C.POINTER(C.c_float)]),
    "mi_op_gemm_skinny": (C.c_int, [C.POINTER(OpLinear), C.POINTER(OpGemvArgs), C.c_int, C.POINTER(C.c_int), C.c_int,
                                    C.POINTER(C.c_float)]),
    "mi_op_gemm_prefill": (C.c_int, [C.POINTER(OpLinear), C.POINTER(OpGemvArgs), C.c_int, C.POINTER(C.c_float)]),
    "mi_op_tiled_bytes": (C.c_uint64, [C.POINTER(OpLinear)]),
    "mi_op_repack_tiled": (C.c_int, [C.POINTER(OpLinear), _P]),
    "mi_op_embed": (C.c_int, [C.POINTER(OpLinear), _P, C.c_int, C.c_int, C.c_int, _P]),
    "mi_op_rope_tables": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float]),
    "mi_op_rope_append": (C.c_int, [C.POINTER(OpAttnShape), _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_int]),
    "mi_op_attention": (C.c_int, [C.POINTER(OpAttnShape), _P, _P, _P, _P, _P, C.c_float, C.c_int, _P]),
    "mi_op_attention_decode": (C.c_int, [C.POINTER(OpAttnShape), _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, C.c_float,
                                         C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "mi_op_sample": (C.c_int, [_P, C.c_int, C.c_int, C.c_float, C.c_float, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def build(verbose: bool = False) -> Path:
    """Compile the HIP sources in-tree (``make`` in mlx_parallm_amd/csrc) for gfx950."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = str(min(8, os.cpu_count() or 2))
    res = subprocess.run(["make", "-C", str(CSRC), "-j", jobs], env=env, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise RuntimeError("building libmi355_decode.so failed")
    return LIB_PATH


def lib() -> C.CDLL:
    """Load the shared library (once) and attach the signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  mlx_parallm_amd has no CPU or PyTorch fallback.")
    try:
        # PyTorch ships its own libamdhip64; whichever HIP runtime is mapped first serves the whole process.
        # Map torch's before ours so that tensors handed to mi_engine_set_tensor and the engine share one
        # runtime (the other order leaves torch without devices: "No HIP GPUs are available").
        import torch  # noqa: F401
    except Exception:          # torch is plumbing, not a requirement of the library itself
        pass
    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if handle.mi_abi_version() != MI_ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {handle.mi_abi_version()}, this binding expects {MI_ABI_VERSION}: "
                          "rebuild the library (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = lib().mi_last_error().decode("utf-8", "replace")
    raise _ERRORS.get(rc, RuntimeError)(msg)
