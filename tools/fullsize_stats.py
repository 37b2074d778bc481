#!/usr/bin/env python3
"""Calibration helper for tests/test_gpu_fullsize.py: noise between the MFMA path and the exact VALU path at
Mistral-7B size (prints RMS / max error, cosine, argmax agreement, rank of one path's argmax in the other)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np

import bench
from mlx_parallm_amd.engine import Engine
from test_gpu_fullsize import _greedy

for wl in ("mistral-7b-bf16", "mistral-7b-int4"):
    family, prec = wl.rsplit("-", 1)
    quant = 4 if prec == "int4" else 0
    cfg = dict(bench.SHAPES[family])
    if quant:
        cfg["quantization"] = {"group_size": 64, "bits": quant}
    eng = Engine(cfg, device=0, max_positions=2048, act_dtype="bfloat16")
    bench.load_synthetic(eng, cfg, 0, quant, 0, 1, None)
    rng = np.random.default_rng(3)
    p = rng.integers(0, cfg["vocab_size"], size=(8, 96)).astype(np.int32)
    fast, lf = _greedy(eng, p, 6)
    exact, le = _greedy(eng, p, 6, force_generic_gemv=1, fused_decode_attention=0, prefill_gemm=0, decode_attention_mfma=0)
    err = lf - le
    cos = [float(np.dot(lf[b], le[b]) / np.linalg.norm(lf[b]) / np.linalg.norm(le[b])) for b in range(8)]
    ranks = [int((lf[b] > lf[b, np.argmax(le[b])]).sum()) for b in range(8)]
    margins = [float(np.sort(le[b])[-1] - np.sort(le[b])[-2]) for b in range(8)]
    print(wl, "logit std", lf.std(), "rms err", np.sqrt((err ** 2).mean()), "max err", np.abs(err).max(), "cos min", min(cos))
    print("  first tokens equal", (fast[0] == exact[0]).mean(), "rank of exact argmax in fast", ranks, "margins", np.round(margins, 3))
    eng.close()
