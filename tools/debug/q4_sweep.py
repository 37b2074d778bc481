#!/usr/bin/env python3
"""Times gemm_q4.hip's plans on the linears of one decoder block (int4 g64, bf16 activations) next to the split-K kernel it
replaces (gemm_skinny.hip with its cost model: run the same command with MI_Q4=0 for that column).

    python tools/debug/q4_sweep.py [--model qwen3-14b] [--rows 64] [--iters 30] [--plans auto|all]
"""
import argparse
import ctypes as C
import itertools
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import torch  # noqa: E402

from mlx_parallm_amd import _lib as L  # noqa: E402
from mlx_parallm_amd.quant import quantize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--plans", default="auto")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    H, I, QD, KVD, V = {"mistral-7b": (4096, 14336, 4096, 1024, 32000), "qwen3-14b": (5120, 17408, 5120, 1024, 151936)}[args.model]
    B = args.rows
    shapes = [("qkv", QD + 2 * KVD, H, 1, L.EPI_STORE), ("o", H, QD, 0, L.EPI_RESID), ("gate_up", 2 * I, H, 1, L.EPI_SWIGLU),
              ("down", H, I, 0, L.EPI_RESID), ("head", V, H, 1, L.EPI_STORE_F32)]
    plans = [(0, 0, 0, 0, 0)]
    if args.plans == "all":
        plans += [(mt, tw, kw, ks, 0) for mt, (tw, kw), ks in itertools.product((2, 4), ((1, 8), (2, 4), (4, 2), (8, 1), (5, 2), (3, 2), (2, 2), (2, 5)), (1, 2, 4))]
    elif args.plans != "auto":
        plans = [tuple(int(v) for v in pl.split(":")) for pl in args.plans.split(",")]
    for name, N, K, pro, epi in shapes:
        if args.only and name not in args.only.split(","):
            continue
        w = (torch.randn((N, K), device="cuda", dtype=torch.float32) * 0.02).to(torch.bfloat16)
        packed, scales, biases = quantize(w, 64, 4)
        del w
        ol = L.OpLinear()
        ol.wk = L.WK["q4_bf16"]
        ol.w, ol.scales, ol.biases = packed.data_ptr(), scales.data_ptr(), biases.data_ptr()
        ol.N, ol.K, ol.group = N, K, 64
        wbytes = packed.numel() * 4 + scales.numel() * 4
        nbytes = int(L.lib().mi_op_tiled_bytes(C.byref(ol)))
        dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        L.check(L.lib().mi_op_repack_tiled(C.byref(ol), C.c_void_p(dst.data_ptr())))
        ol.w, ol.scales, ol.biases, ol.layout = dst.data_ptr(), 0, 0, 1
        x = torch.randn((B, K), device="cuda", dtype=torch.float32).to(torch.bfloat16)
        nw = torch.ones(K, device="cuda", dtype=torch.bfloat16)
        n_out = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.zeros((B, n_out), device="cuda", dtype=torch.float32 if epi == L.EPI_STORE_F32 else torch.bfloat16)
        a = L.OpGemvArgs()
        a.x, a.ldx, a.M, a.act, a.rnd, a.pro, a.epi = x.data_ptr(), K, B, L.MI_BF16, 0, 0, epi
        a.norm_w, a.eps, a.ldo, a.out, a.resid = nw.data_ptr(), 1e-5, n_out, out.data_ptr(), out.data_ptr()
        a.pair_offset, a.force_generic = (N // 2 if epi == L.EPI_SWIGLU else 0), 0
        for mt, tw, kw, ks, ns in plans:
            if mt and ((K // 128) // max(ks, 1) < kw or kw * 4 * mt + 1 > 33 or mt * 16 > max(B, 16) + 15):
                continue
            code = -(mt | tw << 3 | kw << 7 | ks << 11 | ns << 15)
            used, ms = C.c_int(0), C.c_float(0)
            torch.cuda.synchronize()
            try:
                L.check(L.lib().mi_op_gemm_skinny(C.byref(ol), C.byref(a), code, C.byref(used), args.iters, C.byref(ms)))
            except Exception as e:  # noqa: BLE001
                print(f"{name:8s} plan mt={mt} tw={tw} kw={kw} ks={ks} ns={ns}: {e}", flush=True)
                continue
            print(f"{name:8s} N={N:6d} K={K:6d} rows={B:3d} plan mt={mt} tw={tw:2d} kw={kw} ks={ks} ns={ns}  {ms.value * 1e3:8.1f} us  "
                  f"{wbytes / ms.value / 1e6:8.1f} GB/s (ksplit used {used.value})", flush=True)
        del dst, packed, scales, biases


if __name__ == "__main__":
    main()
