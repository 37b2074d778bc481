"""DEBUG BUILD ONLY (not collected by the test suite: the chain kernel is not in libmi355_decode.so).  Run on a GPU box:
    bash tools/debug/build_chain_lib.sh          (in the build container)
    MLX_PARALLM_AMD_LIB=$PWD/mlx_parallm_amd/csrc/alt/libmi355_chain.so python -m pytest tools/debug/chain/test_chain.py -m gpu -q
chain.hip -- the linear layers of one decoder block's decode step (o_proj + residual, RMSNorm + gate|up + SwiGLU,
down_proj + residual, RMSNorm + the next block's q|k|v) as ONE persistent launch with a weight-loader wave per CU --
against the same four linears run as the engine's single launches (mi_op_gemv, themselves oracle-checked in
test_gpu_kernels.py) and against the oracle's arithmetic, at the Mistral-7B and Qwen3-14B layer shapes and at a small
shape with ragged tile counts.  Same rounding points, another float32 summation order: equal to a few 16-bit ulps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ctypes as C  # noqa: E402
import sys  # noqa: E402
from pathlib import Path  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[3] / "tests"))
from gpu_helpers import dev, gemv, gemv_args, host, op_linear, to_tiled  # noqa: E402
from mlx_parallm_amd import _lib as L  # noqa: E402


def chain(linears, args, wait_prev, iters=0):
    """chain.hip on its own (mi_op_chain): -> (kernel give-up code, mean launch ms or None)."""
    n = len(linears)
    lp = (C.POINTER(L.OpLinear) * n)(*[C.pointer(ol) for ol in linears])
    aa = (L.OpGemvArgs * n)(*args)
    wp = (C.c_int32 * n)(*[int(w) for w in wait_prev])
    ms, err = C.c_float(0.0), C.c_int32(0)
    torch.cuda.synchronize()
    fn = L.lib().mi_op_chain           # (only the debug library exports it)
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.POINTER(L.OpLinear)), C.POINTER(L.OpGemvArgs), C.POINTER(C.c_int32), C.c_int, C.c_int,
                   C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.check(fn(lp, aa, wp, n, int(iters), C.byref(ms), C.byref(err)))
    return int(err.value), (ms.value if iters >= 1 else None)



from oracle.numerics import matmul_nt, round_to  # noqa: E402

RNG = np.random.default_rng(11)


def _dense(N, K, dt, std=0.02):
    w = torch.randn((N, K), device="cuda", dtype=torch.float32) * std
    wd = w.to({"bfloat16": torch.bfloat16, "float16": torch.float16}[dt]).contiguous()
    kind = {"bfloat16": "bf16", "float16": "f16"}[dt]
    ol, keep = op_linear(kind, N, K, wd), [wd]
    assert to_tiled(ol, keep)
    return ol, wd, keep


def _block(H, I, NQ, NQKV, M, dt):
    """-> (linears, buffers) of one decoder block's decode-step linears at hidden H, intermediate I, attention width NQ."""
    tdt = {"bfloat16": torch.bfloat16, "float16": torch.float16}[dt]
    o, wo, k1 = _dense(H, NQ, dt)
    gu, wgu, k2 = _dense(2 * I, H, dt)
    dn, wdn, k3 = _dense(H, I, dt)
    qkv, wqkv, k4 = _dense(NQKV, H, dt)
    g = torch.Generator(device="cuda").manual_seed(5)
    bufs = dict(
        attn=(torch.randn((M, NQ), device="cuda", generator=g) * 0.5).to(tdt),
        h=(torch.randn((M, H), device="cuda", generator=g)).to(tdt),
        post=(1.0 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(tdt),
        inn=(1.0 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(tdt),
    )
    return (o, gu, dn, qkv), (wo, wgu, wdn, wqkv), bufs, [k1, k2, k3, k4]


def _run(linears, bufs, M, H, I, NQKV, dt, chained, iters=0):
    o, gu, dn, qkv = linears
    tdt = bufs["h"].dtype
    h = bufs["h"].clone()
    act = torch.zeros((M, I), device="cuda", dtype=tdt)
    out = torch.zeros((M, NQKV), device="cuda", dtype=tdt)
    eps = 1e-5
    args = [
        gemv_args(bufs["attn"], M, dt, epi=L.EPI_RESID, resid=h, ldo=H),
        gemv_args(h, M, dt, pro=L.PRO_NORM, norm_w=bufs["post"], eps=eps, epi=L.EPI_SWIGLU, out=act, ldo=I, pair_offset=I),
        gemv_args(act, M, dt, epi=L.EPI_RESID, resid=h, ldo=H),
        gemv_args(h, M, dt, pro=L.PRO_NORM, norm_w=bufs["inn"], eps=eps, epi=L.EPI_STORE, out=out, ldo=NQKV),
    ]
    ms = None
    if chained:
        err, ms = chain([o, gu, dn, qkv], args, [0, 1, 1, 1], iters=0)
        assert err == 0, f"chain kernel gave up waiting (code {err})"
        res = (host(h), host(act), host(out))
        if iters:
            err, ms = chain([o, gu, dn, qkv], args, [0, 1, 1, 1], iters=iters)       # (h keeps accumulating: timing only)
            assert err == 0
        return res, ms
    import ctypes as C
    for ol, a in zip((o, gu, dn, qkv), args):
        torch.cuda.synchronize()
        L.check(L.lib().mi_op_gemv(C.byref(ol), C.byref(a)))
    res = (host(h), host(act), host(out))
    if iters:
        tot = 0.0
        for ol, a in zip((o, gu, dn, qkv), args):
            t = C.c_float(0.0)
            L.check(L.lib().mi_op_gemv_bench(C.byref(ol), C.byref(a), iters, C.byref(t)))
            tot += t.value
        ms = tot
    return res, ms


def _reference(ws, bufs, I, dt):
    """float64-exact restatement of the four linears with the model's rounding points (the oracle's arithmetic for
    llama.py:143,165,186-190), computed on the host from the same weights and inputs."""
    wo, wgu, wdn, wqkv = [host(w) for w in ws]
    attn, h0, post, inn = host(bufs["attn"]), host(bufs["h"]), host(bufs["post"]), host(bufs["inn"])

    def norm(x, w):
        x64 = x.astype(np.float64)
        rs = 1.0 / np.sqrt((x64 * x64).mean(-1, keepdims=True) + 1e-5)
        return round_to(round_to((x64 * rs).astype(np.float32), dt) * w, dt)

    y = round_to(matmul_nt(attn, wo), dt)
    h1 = round_to(h0 + y, dt)
    xn = norm(h1, post)
    g, u = round_to(matmul_nt(xn, wgu[:I]), dt), round_to(matmul_nt(xn, wgu[I:]), dt)
    sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), dt)
    act = round_to(round_to(g * sig, dt) * u, dt)
    h2 = round_to(h1 + round_to(matmul_nt(act, wdn), dt), dt)
    q = round_to(matmul_nt(norm(h2, inn), wqkv), dt)
    return h2, act, q


def _flip_stats(got, want, scale):
    """A 16-bit rounding of a float32 sum taken in another order flips by one ulp now and then, and a flipped value feeds
    everything behind it: count elements further than 2 bf16 ulps OF THE VALUES THAT WERE ADDED (scale) from the reference."""
    d = np.abs(got - want)
    tol = 2.0 ** -7 * 2 * np.maximum(np.abs(want), scale) + 1e-3
    return float(d.max()), float((d > tol).mean())


@pytest.mark.parametrize("name,H,I,NQ,NQKV", [("mistral-7b", 4096, 14336, 4096, 6144), ("qwen3-14b", 5120, 17408, 5120, 7168),
                                              ("small-ragged", 1024, 2560, 512, 1536)])
@pytest.mark.parametrize("M", [8, 3])
def test_chain_equals_the_single_launches(name, H, I, NQ, NQKV, M):
    """Both launch structures against the float64-exact reference: the chain must be as close to it as the single launches are
    (same rounding points, another float32 summation order), at the production shapes and with ragged unit counts."""
    dt = "bfloat16"
    linears, ws, bufs, keep = _block(H, I, NQ, NQKV, M, dt)
    iters = 20 if M == 8 else 0
    single, t1 = _run(linears, bufs, M, H, I, NQKV, dt, chained=False, iters=iters)
    chained, t2 = _run(linears, bufs, M, H, I, NQKV, dt, chained=True, iters=iters)
    want = _reference(ws, bufs, I, dt)
    for nm, w_, s_, c_, scale in zip(("h", "act", "qkv"), want, single, chained, (1.0, 0.05, 1.0)):
        assert np.isfinite(c_).all(), nm
        ms, fs = _flip_stats(s_, w_, scale)
        mc, fc = _flip_stats(c_, w_, scale)
        print(f"{name} M={M} {nm}: vs float64 reference -- single launches max {ms:.3e} / beyond 2 ulp {fs:.2e}; chain max {mc:.3e} / {fc:.2e}")
        assert fc <= max(2.0 * fs, 2e-3) and mc <= max(2.0 * ms, 0.1), (name, nm, (ms, fs), (mc, fc))
    if iters:
        print(f"{name}: four single launches {t1 * 1e3:.1f} us, one chain launch {t2 * 1e3:.1f} us")


def test_chain_against_the_oracle_arithmetic():
    """Small block, ragged unit counts, 5 rows: the chain alone against the float64-exact reference."""
    dt, M, H, I, NQ, NQKV = "bfloat16", 5, 1024, 1536, 1024, 1536
    linears, ws, bufs, keep = _block(H, I, NQ, NQKV, M, dt)
    got, _ = _run(linears, bufs, M, H, I, NQKV, dt, chained=True)
    for nm, want, g_, scale in zip(("h", "act", "qkv"), _reference(ws, bufs, I, dt), got, (1.0, 0.05, 1.0)):
        mx, frac = _flip_stats(g_, want, scale)
        assert frac <= 5e-3, (nm, mx, frac)
