"""The C-ABI library must load on a GPU-less host and export every symbol that include/*.h declares
(no compute calls here).  Also: the product fails loudly without a device -- no CPU fallback."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = set()
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_headers_declare_the_expected_surface():
    syms = declared_symbols()
    for required in ("mi_engine_create", "mi_engine_set_tensor", "mi_engine_set_lora", "mi_engine_finalize",
                     "mi_kv_create", "mi_kv_reset", "mi_kv_reserve", "mi_forward", "mi_decode_sample",
                     "mi_step_enqueue", "mi_step_wait", "mi_last_error", "mi_op_gemv", "mi_op_attention_decode"):
        assert required in syms


def test_library_exports_every_declared_symbol():
    from mlx_parallm_amd import _lib

    lib = _lib.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported by {_lib.LIB_PATH.name}"
    # and the ctypes signature table covers exactly the declared surface
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_no_cpu_backend_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mlx_parallm_amd import _lib
    from mlx_parallm_amd.engine import Engine

    cfg = {"model_type": "llama", "hidden_size": 64, "num_hidden_layers": 1, "intermediate_size": 128,
           "num_attention_heads": 4, "rms_norm_eps": 1e-6, "vocab_size": 128}
    with pytest.raises(RuntimeError, match="no HIP device"):
        Engine(cfg)
    ol, a = _lib.OpLinear(), _lib.OpGemvArgs()
    assert _lib.lib().mi_op_gemv(C.byref(ol), C.byref(a)) == -4          # MI_ERR_RUNTIME, not a silent fallback


def test_product_never_imports_the_oracle():
    for py in (ROOT / "mlx_parallm_amd").rglob("*.py"):
        text = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{py} imports the oracle"
    assert "oracle" not in (ROOT / "mlx_parallm_amd" / "csrc" / "Makefile").read_text()


def test_struct_size_guard_refuses_a_binding_from_another_header_version():
    """mi_model_desc / mi_sample_params start with struct_size (ABI guard, mi355_decode.h): a short or stale struct is
    refused with MI_ERR_INVALID before anything behind it is read -- checked before the device is even looked for."""
    from mlx_parallm_amd import _lib

    lib = _lib.lib()
    assert lib.mi_abi_version() == _lib.MI_ABI_VERSION
    d = _lib.ModelDesc()
    d.struct_size = C.sizeof(_lib.ModelDesc) - 4
    h = C.c_void_p()
    assert lib.mi_engine_create(C.byref(d), 0, C.byref(h)) == -1 and b"struct_size" in lib.mi_last_error()
    sp = _lib.SampleParams()                      # struct_size left 0: a round-1 binding
    t = C.c_int64(-1)
    assert lib.mi_step_enqueue(None, None, None, 1, 1, C.byref(sp), C.byref(t)) == -1
    assert b"mi_sample_params.struct_size" in lib.mi_last_error()
    rows = (C.c_int32 * 1)(0)
    assert lib.mi_step_enqueue_rows(h, h, rows, 1, None, 1, C.byref(sp), C.byref(t)) == -1
    # the header and the ctypes mirror agree on both layouts (compiled probe)
    import subprocess
    import tempfile

    src = '#include <stdio.h>\n#include <stddef.h>\n#include "mi355_decode.h"\nint main(void){printf("%zu %zu %zu %zu\\n", ' \
          'sizeof(mi_model_desc), sizeof(mi_sample_params), offsetof(mi_sample_params, row_top_p), ' \
          'offsetof(mi_sample_params, seed));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "p.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(Path(td) / "p.c"), "-o", str(Path(td) / "p")], check=True)
        out = subprocess.run([str(Path(td) / "p")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(_lib.ModelDesc), C.sizeof(_lib.SampleParams),
                                     _lib.SampleParams.row_top_p.offset, _lib.SampleParams.seed.offset]


def test_integration_stub_is_the_generated_one():
    """INTEGRATION.md section B shows the reference-side ctypes stub; its struct block is generated from _lib.py."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_stub", ROOT / "tools" / "gen_integration_stub.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.block() in (ROOT / "INTEGRATION.md").read_text(), "run tools/gen_integration_stub.py --write"


def test_no_launched_kernel_spills_registers():
    """The build keeps the compiler's resource report per source (csrc/*.res, Makefile); a kernel the engine launches
    must not spill to scratch -- in a weight-streaming loop a spill costs more than any register it frees."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("kernel_resources", ROOT / "tools" / "kernel_resources.py")
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.collect()
    if not rows:
        pytest.skip("library built without the resource report (no csrc/*.res)")
    names = {r["kernel"] for r in rows}
    assert any(n.startswith("skinny_kernel<bf16, 0, 1, false, false>") for n in names), "demangling failed"
    assert len(rows) > 500
    bad = kr.spilling(rows)
    assert not bad, "register spills in: " + ", ".join(f'{r["kernel"]} ({r["scratch"]} B/lane)' for r in bad)
