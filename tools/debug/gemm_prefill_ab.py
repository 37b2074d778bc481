"""The prefill tile GEMM on its own (mi_op_gemm_prefill): LDS-DMA tile vs the register-staged tile, interleaved rounds in
one process, Mistral-7B linears over 8 x 1024 rows, random bf16 operands."""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from mlx_parallm_amd import _lib as L
from gpu_helpers import gemm_prefill, op_linear, to_tiled

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shapes = [("qkv", 6144, 4096, L.EPI_STORE), ("o", 4096, 4096, L.EPI_RESID), ("gate_up", 28672, 4096, L.EPI_SWIGLU), ("down", 4096, 14336, L.EPI_RESID)]
g = torch.Generator(device="cuda"); g.manual_seed(1)
for name, N, K, epi in shapes:
    w = (torch.rand((N, K), device="cuda", generator=g) * 0.1 - 0.05).to(torch.bfloat16)
    ol, keep = op_linear("bf16", N, K, w), [w]
    assert to_tiled(ol, keep)
    x = (torch.rand((M, K), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    no = N // 2 if epi == L.EPI_SWIGLU else N
    out = torch.zeros((M, no), dtype=torch.bfloat16, device="cuda")
    kw = dict(epi=epi, out=out, ldo=no, resid=out if epi == L.EPI_RESID else None, pair_offset=N // 2 if epi == L.EPI_SWIGLU else 0)
    res = {k: [] for k in "01"}
    for r in range(rounds):
        for v in "01":
            os.environ["MI_GEMM_DMA"] = v
            res[v].append(gemm_prefill(ol, x, M, "bfloat16", iters=10, **kw))
    fl = 2.0 * M * N * K
    for v, lab in (("0", "register-staged"), ("1", "LDS-DMA ping-pong")):
        ms = np.array(res[v])
        print(f"{name:8s} {lab:20s} median {np.median(ms)*1e3:8.1f} us  min {ms.min()*1e3:8.1f} us   {fl/np.median(ms)/1e9:7.1f} TFLOP/s (median)", flush=True)
