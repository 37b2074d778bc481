#!/usr/bin/env python3
"""Micro-benchmark of the fused decode-attention kernel (mi_op_attention_decode): Mistral-7B /
Qwen3-14B head geometry, batch 8, KV length S, sweeping the split count.  Rotates over `--layers`
KV buffers so that re-reads are not served by the 256 MiB Infinity Cache."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from mlx_parallm_amd import _lib as L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=1024)
    ap.add_argument("--Hq", type=int, default=32)
    ap.add_argument("--Hkv", type=int, default=8)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--splits", default="1,2,4,8,16")
    ap.add_argument("--variant", type=int, default=0, help="0: MFMA kernel, 1: VALU kernel")
    ap.add_argument("--no-combine", type=int, default=0, help="timing experiment: skip the ticket/combine tail")
    a = ap.parse_args()
    B, S, Hq, Hkv, D = a.B, a.S, a.Hq, a.Hkv, a.D
    cap = S + 64
    dev = "cuda"
    kcs = [torch.randn((B, Hkv, cap, D), device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(a.layers)]
    vcs = [torch.randn((B, Hkv, cap, D), device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(a.layers)]
    qkv = torch.randn((B, (Hq + 2 * Hkv) * D), device=dev, dtype=torch.float32).to(torch.bfloat16)
    out = torch.zeros((B, Hq * D), device=dev, dtype=torch.bfloat16)
    offs = torch.full((B,), S, device=dev, dtype=torch.int32)
    cos = torch.zeros((cap + 1, D // 2), device=dev, dtype=torch.float32)
    sin = torch.zeros_like(cos)
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_rope_tables(C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), cap + 1, D, 1e4, 1.0))
    sh = L.OpAttnShape()
    sh.B, sh.L, sh.Hq, sh.Hkv, sh.D, sh.act, sh.kv, sh.rnd, sh.cap = B, 1, Hq, Hkv, D, L.MI_BF16, L.MI_BF16, 0, cap
    kv_bytes = B * Hkv * (S + 1) * D * 2 * 2
    for ns in [int(x) for x in a.splits.split(",")]:
        part = torch.zeros((B * Hq * ns * (D + 2),), device=dev, dtype=torch.float32)
        ctr = torch.zeros((B * Hkv,), device=dev, dtype=torch.int32)
        torch.cuda.synchronize()
        tot = 0.0
        for i in range(a.layers):
            ms = C.c_float(0)
            L.check(L.lib().mi_op_attention_decode(
                C.byref(sh), C.c_void_p(qkv.data_ptr()), C.c_void_p(kcs[i].data_ptr()), C.c_void_p(vcs[i].data_ptr()),
                C.c_void_p(offs.data_ptr()), None, None, 1e-6, C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()),
                C.c_void_p(out.data_ptr()), float(D ** -0.5), 0, ns, C.c_void_p(part.data_ptr()),
                None if a.no_combine else C.c_void_p(ctr.data_ptr()), a.variant, a.iters if a.layers == 1 else 2, C.byref(ms)))
            tot += ms.value
        avg = tot / a.layers
        print(f"nsplit={ns:3d}  {avg*1e3:8.1f} us   {kv_bytes/avg/1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
