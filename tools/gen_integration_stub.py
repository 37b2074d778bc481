#!/usr/bin/env python3
"""Prints the ctypes struct block of INTEGRATION.md section B from mlx_parallm_amd/_lib.py, so the stub a reference
maintainer copies cannot drift from the binding the tests exercise (tests/test_abi_symbols.py compares them).

    python tools/gen_integration_stub.py            # print the block
    python tools/gen_integration_stub.py --write    # splice it into INTEGRATION.md between the markers
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

BEGIN, END = "# --- BEGIN generated structs (tools/gen_integration_stub.py) ---", "# --- END generated structs ---"

_NAMES = {C.c_int32: "C.c_int32", C.c_uint32: "C.c_uint32", C.c_float: "C.c_float", C.c_uint64: "C.c_uint64",
          C.c_int64: "C.c_int64", C.c_void_p: "C.c_void_p", C.POINTER(C.c_int32): "C.POINTER(C.c_int32)",
          C.POINTER(C.c_float): "C.POINTER(C.c_float)"}


def struct_src(cls, comment: str) -> str:
    items = [f'("{n}", {_NAMES[t]})' for n, t in cls._fields_]
    lines, cur = [], "    _fields_ = ["
    for i, it in enumerate(items):
        piece = it + ("," if i + 1 < len(items) else "]")
        if len(cur) + len(piece) + 1 > 116:
            lines.append(cur.rstrip())
            cur = " " * 16
        cur += piece + " "
    lines.append(cur.rstrip())
    return f"class {cls.__name__}(C.Structure):   # {comment}\n" + "\n".join(lines) + "\n"


def block() -> str:
    from mlx_parallm_amd import _lib

    out = [BEGIN,
           f"MI_ABI_VERSION = {_lib.MI_ABI_VERSION}      # lib.mi_abi_version() must return this",
           struct_src(_lib.ModelDesc, "mi_model_desc  <- ModelArgs (llama.py:15-46) + config[\"quantization\"]"),
           struct_src(_lib.SampleParams, "mi_sample_params  <- the `sample` closure's free variables (utils.py:345-364)"),
           "# both structs: set .struct_size = C.sizeof(<class>) before every call (ABI guard, mi355_decode.h)",
           END]
    return "\n".join(out)


if __name__ == "__main__":
    b = block()
    if "--write" in sys.argv:
        p = ROOT / "INTEGRATION.md"
        s = p.read_text()
        i, j = s.index(BEGIN), s.index(END) + len(END)
        p.write_text(s[:i] + b + s[j:])
    else:
        print(b)
