#!/usr/bin/env python3
"""Kernel micro-benchmark: the decode GEMV launches of one Mistral-7B / Qwen3-14B block, timed back to
back with HIP events (mi_op_gemv_bench).  Prints algorithmic GB/s per kernel.  Rotates over several
weight copies so that the 256 MiB Infinity Cache cannot serve re-reads."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from mlx_parallm_amd import _lib as L
from mlx_parallm_amd.quant import quantize


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--quant", type=int, default=0)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--generic", type=int, default=0)
    ap.add_argument("--tiled", type=int, default=1)
    ap.add_argument("--skinny", action="store_true", help="time gemm_skinny also for batch <= 16 (set MI_SKINNY_MIN_ROWS=1 for batch <= 8)")
    ap.add_argument("--ksplit", default="0", help="batch > 16 (gemm_skinny): comma-separated K splits to time, 0 = cost model")
    ap.add_argument("--lib", default=None, help="A/B: load this build of libmi355_decode.so instead")
    args = ap.parse_args()
    if args.lib:
        L.LIB_PATH = Path(args.lib)
        old_sig = dict(L.SIGNATURES)
        probe = C.CDLL(args.lib)
        L.SIGNATURES = {k: v for k, v in old_sig.items() if hasattr(probe, k)}
        if "mi_op_tiled_bytes" not in L.SIGNATURES:
            args.tiled = 0
    H, I, QD, KVD, V = {"mistral-7b": (4096, 14336, 4096, 1024, 32000), "qwen3-14b": (5120, 17408, 5120, 1024, 151936)}[args.model]
    B = args.batch
    dev = "cuda"
    shapes = [("qkv", QD + 2 * KVD, H, L.PRO_NORM, L.EPI_STORE), ("o", H, QD, L.PRO_NONE, L.EPI_RESID),
              ("gate_up", 2 * I, H, L.PRO_NORM, L.EPI_SWIGLU), ("down", H, I, L.PRO_NONE, L.EPI_RESID),
              ("head", V, H, L.PRO_NORM, L.EPI_STORE_F32)]
    for name, N, K, pro, epi in shapes:
        w = (torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        ol = L.OpLinear()
        keep = [w]
        if args.quant:
            packed, scales, biases = quantize(w, 64, args.quant)
            keep = [packed, scales, biases]
            ol.wk = L.WK["q4_bf16" if args.quant == 4 else "q8_bf16"]
            ol.w, ol.scales, ol.biases = packed.data_ptr(), scales.data_ptr(), biases.data_ptr()
            wbytes = packed.numel() * 4 + scales.numel() * 4
        else:
            ol.wk = L.WK["bf16"]
            ol.w = w.data_ptr()
            wbytes = w.numel() * 2
        ol.N, ol.K, ol.group = N, K, 64
        if args.tiled:
            nbytes = int(L.lib().mi_op_tiled_bytes(C.byref(ol)))
            dst = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            L.check(L.lib().mi_op_repack_tiled(C.byref(ol), C.c_void_p(dst.data_ptr())))
            keep.append(dst)
            ol.w, ol.scales, ol.biases, ol.layout = dst.data_ptr(), 0, 0, 1
        x = torch.randn((B, K), device=dev, dtype=torch.float32).to(torch.bfloat16)
        nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
        n_out = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.zeros((B, n_out), device=dev, dtype=torch.float32 if epi == L.EPI_STORE_F32 else torch.bfloat16)
        a = L.OpGemvArgs()
        a.x, a.ldx, a.M, a.act, a.rnd, a.pro, a.epi = x.data_ptr(), K, B, L.MI_BF16, 0, pro, epi
        a.norm_w, a.eps, a.ldo, a.out, a.resid = nw.data_ptr(), 1e-5, n_out, out.data_ptr(), out.data_ptr()
        a.pair_offset, a.force_generic = (N // 2 if epi == L.EPI_SWIGLU else 0), args.generic
        torch.cuda.synchronize()
        ms = C.c_float(0)
        if B > 16 or args.skinny:                     # the split-K streaming GEMM (the engine normalises first)
            a.pro = L.PRO_NONE
            for ks in [int(v) for v in args.ksplit.split(",")]:
                used = C.c_int(0)
                L.check(L.lib().mi_op_gemm_skinny(C.byref(ol), C.byref(a), ks, C.byref(used), args.iters, C.byref(ms)))
                print(f"{name:8s} N={N:6d} K={K:6d}  {ms.value*1e3:8.1f} us  {wbytes/ms.value/1e6:8.1f} GB/s  (skinny, ksplit={used.value})", flush=True)
            del keep, w
            continue
        L.check(L.lib().mi_op_gemv_bench(C.byref(ol), C.byref(a), args.iters, C.byref(ms)))
        print(f"{name:8s} N={N:6d} K={K:6d}  {ms.value*1e3:8.1f} us  {wbytes/ms.value/1e6:8.1f} GB/s  (mfma={L.lib().mi_op_gemv_uses_mfma(C.byref(ol), C.byref(a))})", flush=True)
        del keep, w


if __name__ == "__main__":
    main()
