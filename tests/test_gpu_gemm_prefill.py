"""Kernel-level parity (-m gpu) of gemm_prefill.hip's 256 x 256 tile -- the GEMM of the prefill call (generate_step's
first model call, utils.py:243-262 -> every nn.Linear over B x L rows, llama.py:64-67,93,143,160-165) -- through
mi_op_gemm_prefill (include/mi355_ops.h): against the oracle's matmul on sampled rows, and the LDS-DMA kernel bit for bit
against the register-staged kernel it replaced (same MFMA chain per accumulator, so the float32 sums must be identical)."""
import os

import numpy as np
import pytest
import torch

from oracle.numerics import matmul_nt, round_to

pytestmark = pytest.mark.gpu

from mlx_parallm_amd import _lib as L  # noqa: E402
from gpu_helpers import dev, gemm_prefill, host, op_linear, to_tiled  # noqa: E402
from test_gpu_kernels import _assert_close  # noqa: E402

RNG = np.random.default_rng(777)
MODES = (1,)      # MI_GEMM_DMA: 1 = the LDS-DMA tile (default), 0 = the register-staged tile


def _weight(kind, N, K):
    dt = {"bf16": "bfloat16", "f16": "float16"}[kind]
    w = round_to(RNG.standard_normal((N, K)).astype(np.float32) * 0.05, dt)
    wd = dev(w, dt)
    ol, keep = op_linear(kind, N, K, wd), [wd]
    assert to_tiled(ol, keep)
    return ol, w, keep


def _run(ol, xd, M, act, dma, **kw):
    old = os.environ.get("MI_GEMM_DMA")
    os.environ["MI_GEMM_DMA"] = str(int(dma))
    try:
        gemm_prefill(ol, xd, M, act, **kw)
        torch.cuda.synchronize()
    finally:
        if old is None:
            del os.environ["MI_GEMM_DMA"]
        else:
            os.environ["MI_GEMM_DMA"] = old


def _rows(M):
    """the ragged last block whole, the first rows, and a few from the middle"""
    pick = set(range(min(M, 20))) | set(range(max(0, (M - 1) // 256 * 256), M)) | set(RNG.integers(0, M, 24).tolist())
    return np.array(sorted(pick))


@pytest.mark.parametrize("act,kind", [("bfloat16", "bf16"), ("float16", "f16")])
@pytest.mark.parametrize("M,N,K", [
    (2085, 6144, 512),       # 9 x 24 blocks, ragged last row block (37 rows)
    (2048, 6400, 128),       # two K tiles: prologue and tail only
    (2304, 5632, 192),       # odd number of K tiles
    (4096, 4096, 1024),      # 16 x 16 blocks, 16 K tiles
])
def test_store_and_residual_match_oracle_and_the_register_staged_tile(act, kind, M, N, K):
    ol, w, keep = _weight(kind, N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    h0 = round_to(RNG.standard_normal((M, N)).astype(np.float32), act)
    xd = dev(x, act)
    rows = _rows(M)
    want = round_to(matmul_nt(x[rows], w), act)
    got = {}
    for dma in MODES + (0,):
        out = torch.full((M + 3, N), 7.0, dtype=xd.dtype, device="cuda")
        _run(ol, xd, M, act, dma, epi=L.EPI_STORE, out=out, ldo=N)
        o = host(out)
        assert np.all(o[M:] == 7.0), "rows past M were written"
        got[dma] = o[:M]
    _assert_close(got[1][rows], want, act)
    for m in MODES:
        assert np.array_equal(got[m], got[0]), f"LDS-DMA tile (MI_GEMM_DMA={m}) differs from the register-staged tile"
    # residual epilogue: h += y, in place
    hres = {}
    for dma in MODES + (0,):
        h = dev(h0, act)
        _run(ol, xd, M, act, dma, epi=L.EPI_RESID, resid=h, out=h, ldo=N)
        hres[dma] = host(h)
    y = round_to(matmul_nt(x[rows], w), act)
    _assert_close(hres[1][rows], round_to(h0[rows] + y, act), act)
    for m in MODES:
        assert np.array_equal(hres[m], hres[0]), f"MI_GEMM_DMA={m}"


@pytest.mark.parametrize("act,kind", [("bfloat16", "bf16"), ("float16", "f16")])
@pytest.mark.parametrize("M,I,K", [(2085, 3072, 512), (2048, 3200, 256)])
def test_swiglu_matches_oracle_and_the_register_staged_tile(act, kind, M, I, K):
    ol, w, keep = _weight(kind, 2 * I, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    xd = dev(x, act)
    rows = _rows(M)
    y = round_to(matmul_nt(x[rows], w), act)
    gt, up = y[:, :I].astype(np.float64), y[:, I:]
    sig = round_to((1.0 / (1.0 + np.exp(-gt))).astype(np.float32), act)
    sl = round_to(gt.astype(np.float32) * sig, act)
    want = round_to(sl * up, act)
    got = {}
    for dma in MODES + (0,):
        out = torch.full((M + 1, I), 7.0, dtype=xd.dtype, device="cuda")
        _run(ol, xd, M, act, dma, epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I)
        o = host(out)
        assert np.all(o[M:] == 7.0)
        got[dma] = o[:M]
    _assert_close(got[1][rows], want, act, scale=float(np.abs(y).max()))
    for m in MODES:
        assert np.array_equal(got[m], got[0]), f"LDS-DMA tile (MI_GEMM_DMA={m}) differs from the register-staged tile"


@pytest.mark.parametrize("act,kind", [("bfloat16", "bf16"), ("float16", "f16")])
@pytest.mark.parametrize("M,N,K", [
    (1024, 4096, 4096),      # 4 x 16 tiles -> K split 4 ways, 16 K tiles per slice
    (300, 2048, 2048),       # 2 x 8 tiles, ragged rows -> 4 slices of 8 K tiles
    (2085, 4096, 1536),      # 9 x 16 tiles -> 2 slices of 12 K tiles
])
def test_k_split_of_the_lds_dma_tile(act, kind, M, N, K):
    """One prompt of a few hundred rows: too few 256 x 256 tiles for the chip, so K is split over workgroups (float32
    partial tiles + the ordered reduce).  Against the oracle on sampled rows; deterministic; the residual epilogue too."""
    ol, w, keep = _weight(kind, N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    h0 = round_to(RNG.standard_normal((M, N)).astype(np.float32), act)
    xd = dev(x, act)
    rows = _rows(M)
    y = round_to(matmul_nt(x[rows], w), act)
    outs = []
    for _ in range(2):
        out = torch.full((M + 3, N), 7.0, dtype=xd.dtype, device="cuda")
        _run(ol, xd, M, act, 1, epi=L.EPI_STORE, out=out, ldo=N)
        o = host(out)
        assert np.all(o[M:] == 7.0), "rows past M were written"
        outs.append(o[:M])
    _assert_close(outs[0][rows], y, act)
    assert np.array_equal(outs[0], outs[1])
    # the split really happened: the unsplit 128 x 128 path sums in another order (some values differ in the last place)
    os.environ["MI_GEMM_DMA_SPLITK"] = "0"
    try:
        out = torch.full((M, N), 7.0, dtype=xd.dtype, device="cuda")
        _run(ol, xd, M, act, 1, epi=L.EPI_STORE, out=out, ldo=N)
    finally:
        del os.environ["MI_GEMM_DMA_SPLITK"]
    _assert_close(host(out)[rows], y, act)
    h = dev(h0, act)
    _run(ol, xd, M, act, 1, epi=L.EPI_RESID, resid=h, out=h, ldo=N)
    _assert_close(host(h)[rows], round_to(h0[rows] + y, act), act)
