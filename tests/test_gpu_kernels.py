"""Kernel-level parity (-m gpu): every HIP kernel against the oracle's restatement of the MLX
op it replaces, through the C ABI of include/mi355_ops.h.

Tolerances.  float32 activations: |err| <= 2e-5 * scale (fp32 accumulation order).  16-bit
activations: results are rounded to the dtype at the same points as the oracle, so elements
either match exactly or differ by one rounding step where the fp32 sums straddle a rounding
boundary; we require >= 99 % of elements within 2 ulp AND every element within 4 ulp.
"""
import ctypes as C

import os

import numpy as np
import pytest
import torch

from oracle import numerics, ref_model, ref_quant, ref_sample
from oracle.numerics import matmul_nt, round_to

pytestmark = pytest.mark.gpu

from mlx_parallm_amd import _lib as L  # noqa: E402
from gpu_helpers import (MIDT, attn_shape, close_frac, dev, dev_i32, dev_u32, gemv, host, op_linear, ptr, to_tiled)  # noqa: E402

RNG = np.random.default_rng(1234)


def _assert_close(got, want, dtype, scale=None):
    """`scale` = magnitude of the largest intermediate that is rounded to `dtype` on the way to the
    result (default: the RMS of the result).  A rounding step may land one unit off where the fp32 sums
    of kernel and oracle straddle a rounding boundary, so: every element within 2 units of
    max(|want|, scale), and at least 90 % within half a unit."""
    if scale is None:
        scale = float(np.sqrt(np.mean(np.square(want)))) + 1e-12
    if dtype == "float32":
        assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * scale), np.abs(got - want).max()
        return
    unit = {"bfloat16": 2.0 ** -7, "float16": 2.0 ** -10}[dtype] * np.maximum(np.abs(want), scale)
    err = np.abs(got - want)
    assert np.all(err <= 2 * unit), float((err / unit).max())
    assert np.mean(err > 0.5 * unit) <= 0.10, float(np.mean(err > 0.5 * unit))


TILED = True      # exercise the tile-major weight layout (what mi_engine_finalize produces) where eligible


def _make_weight(kind, N, K, act, tiled=None):
    """-> (op_linear, dense float32 view the oracle multiplies with, keepalive tensors)."""
    tiled = TILED if tiled is None else tiled
    if kind in ("f32", "bf16", "f16"):
        dt = {"f32": "float32", "bf16": "bfloat16", "f16": "float16"}[kind]
        w = round_to(RNG.standard_normal((N, K)).astype(np.float32) * 0.05, dt)
        wd = dev(w, dt)
        ol, keep = op_linear(kind, N, K, wd), [wd]
        if tiled:
            to_tiled(ol, keep)
        return ol, w, keep
    bits = 4 if kind.startswith("q4") else 8
    sdt = {"f32": "float32", "bf16": "bfloat16", "f16": "float16"}[kind.split("_")[1]]
    w = RNG.standard_normal((N, K)).astype(np.float32) * 0.05
    packed, scales, biases = ref_quant.quantize(round_to(w, sdt), 64, bits, sdt)
    pd, sd, bd = dev_u32(packed), dev(scales, sdt), dev(biases, sdt)
    ol, keep = op_linear(kind, N, K, pd, sd, bd), [pd, sd, bd]
    if tiled:
        to_tiled(ol, keep)
    return ol, ref_quant.dequantize(packed, scales, biases, 64, bits), keep


@pytest.mark.parametrize("kind,act", [("bf16", "bfloat16"), ("q4_bf16", "bfloat16"), ("f16", "float16"), ("q4_f16", "float16"),
                                      ("bf16", "float32"), ("q4_bf16", "float32")])
@pytest.mark.parametrize("force_generic", [0, 1])
def test_tiled_and_row_major_layouts_agree_bitwise(kind, act, force_generic):
    """Same matrix, both layouts, same kernel: the outputs must be IDENTICAL (only addresses change)."""
    M, N, K = 7, 112, 512
    outs = []
    state = RNG.bit_generator.state
    for tiled in (False, True):
        RNG.bit_generator.state = state
        ol, wdense, keep = _make_weight(kind, N, K, act, tiled=tiled)
        assert bool(ol.layout) == tiled
        x = round_to(np.random.default_rng(5).standard_normal((M, K)).astype(np.float32), act)
        xd = dev(x, act)
        out = torch.zeros((M, N), dtype=xd.dtype, device="cuda")
        gemv(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N, force_generic=force_generic)
        outs.append(host(out))
    if force_generic or act == "float32":
        assert np.array_equal(outs[0], outs[1])      # same (generic) kernel, only the addressing differs
    for o in outs:                                   # (force_generic=0: row-major -> generic, tiled -> MFMA)
        _assert_close(o, round_to(matmul_nt(x, wdense), act), act)


COMBOS = [
    # (activation dtype, weight kind)
    ("float32", "f32"), ("float32", "q4_f32"), ("float32", "q8_f32"), ("float32", "bf16"), ("float32", "q4_bf16"),
    ("bfloat16", "bf16"), ("bfloat16", "q4_bf16"), ("bfloat16", "q8_bf16"),
    ("float16", "f16"), ("float16", "q4_f16"), ("float16", "q8_f16"),
]


@pytest.mark.parametrize("force_generic", [1, 0])
@pytest.mark.parametrize("M", [1, 3, 8])
@pytest.mark.parametrize("act,kind", COMBOS)
def test_gemv_norm_store(act, kind, M, force_generic):
    """RMSNorm prologue + projection (llama.py:187 -> :93)."""
    N, K = 80, 256
    ol, wdense, keep = _make_weight(kind, N, K, act)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    nw = round_to(1.0 + 0.1 * RNG.standard_normal(K).astype(np.float32), act)
    xn, _ = ref_model.rms_norm(x, act, nw, act, 1e-5)
    want = round_to(matmul_nt(xn, wdense), act)
    xd, nwd = dev(x, act), dev(nw, act)
    out = torch.zeros((M, N), dtype=xd.dtype, device="cuda")
    used = gemv(ol, xd, M, act, pro=L.PRO_NORM, norm_w=nwd, eps=1e-5, epi=L.EPI_STORE, out=out, ldo=N,
                force_generic=force_generic)
    if force_generic:
        assert not used
    _assert_close(host(out), want, act)


@pytest.mark.parametrize("force_generic", [1, 0])
@pytest.mark.parametrize("act,kind", [("float32", "f32"), ("float32", "q4_f32"), ("bfloat16", "bf16"),
                                      ("bfloat16", "q4_bf16"), ("float16", "f16"), ("float16", "q4_f16")])
def test_gemv_epilogues(act, kind, force_generic):
    """residual add (llama.py:188,190), SwiGLU (llama.py:165), float32 logits store, K > one LDS chunk."""
    M = 5
    # --- residual: h = h + W x, K = 4608 spans several x chunks
    N, K = 64, 4608
    ol, wdense, keep = _make_weight(kind, N, K, act)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32) * 0.5, act)
    h = round_to(RNG.standard_normal((M, N)).astype(np.float32), act)
    r = round_to(matmul_nt(x, wdense), act)
    want = round_to(h + r, act)
    xd, hd = dev(x, act), dev(h, act)
    gemv(ol, xd, M, act, epi=L.EPI_RESID, resid=hd, ldo=N, force_generic=force_generic)
    _assert_close(host(hd), want, act, scale=4.0)
    # --- SwiGLU over a fused gate|up matrix
    I, K = 48, 256
    ol, wdense, keep = _make_weight(kind, 2 * I, K, act)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    g = round_to(matmul_nt(x, wdense[:I]), act)
    u = round_to(matmul_nt(x, wdense[I:]), act)
    sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), act)
    want = round_to(round_to(g * sig, act) * u, act)
    xd = dev(x, act)
    out = torch.zeros((M, I), dtype=xd.dtype, device="cuda")
    gemv(ol, xd, M, act, epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I, force_generic=force_generic)
    _assert_close(host(out), want, act)
    # --- float32 logits (lm_head), odd N for the generic path / multiple of 16 for MFMA
    N = 80 if not force_generic else 77
    ol, wdense, keep = _make_weight(kind, N, K, act)
    want = round_to(matmul_nt(x, wdense), act)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    gemv(ol, xd, M, act, epi=L.EPI_STORE_F32, out=out, ldo=N, force_generic=force_generic)
    _assert_close(host(out), want, act)


@pytest.mark.parametrize("M", [9, 16])
@pytest.mark.parametrize("act,kind", [("bfloat16", "bf16"), ("bfloat16", "q4_bf16"), ("float16", "q4_f16")])
def test_gemv_mfma_16_rows(act, kind, M):
    N, K = 96, 512
    ol, wdense, keep = _make_weight(kind, N, K, act)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    want = round_to(matmul_nt(x, wdense), act)
    xd = dev(x, act)
    out = torch.zeros((M, N), dtype=xd.dtype, device="cuda")
    used = gemv(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N)
    assert used
    _assert_close(host(out), want, act)


def test_gemv_runtime_rounding_mode():
    """float32 storage + bf16 logical rounding = the layer-0 half of the PagedKVCache quirk."""
    M, N, K = 4, 40, 256
    ol, wdense, keep = _make_weight("bf16", N, K, "float32")
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), "bfloat16")
    nw = round_to(1.0 + 0.1 * RNG.standard_normal(K).astype(np.float32), "bfloat16")
    xn, _ = ref_model.rms_norm(x, "bfloat16", nw, "bfloat16", 1e-6)
    want = round_to(matmul_nt(xn, wdense), "bfloat16")
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    gemv(ol, dev(x), M, "float32", rnd=L.RND_BF16, pro=L.PRO_NORM, norm_w=dev(nw), eps=1e-6, epi=L.EPI_STORE,
         out=out, ldo=N)
    _assert_close(host(out), want, "bfloat16")


@pytest.mark.parametrize("kind,act", [("f32", "float32"), ("bf16", "bfloat16"), ("q4_f32", "float32"),
                                      ("q4_bf16", "bfloat16"), ("q8_f16", "float16")])
def test_embed(kind, act):
    V, H = 200, 128
    ol, wdense, keep = _make_weight(kind, V, H, act)
    toks = RNG.integers(0, V, size=11)
    out = torch.zeros((11, H), dtype=dev(np.zeros(1), act).dtype, device="cuda")
    td = dev_i32(toks)
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_embed(C.byref(ol), ptr(td), 11, MIDT[act], 0, ptr(out)))
    assert np.array_equal(host(out), round_to(wdense[toks], act))


def _rope_setup(D, max_pos, base=10000.0, scale=1.0):
    cos = torch.zeros((max_pos, D // 2), dtype=torch.float32, device="cuda")
    sin = torch.zeros_like(cos)
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_rope_tables(ptr(cos), ptr(sin), max_pos, D, base, scale))
    c_ref, s_ref = ref_model.rope_tables(D, base, scale, max_pos)
    assert np.allclose(host(cos), c_ref, atol=1e-7) and np.allclose(host(sin), s_ref, atol=1e-7)
    return cos, sin, c_ref, s_ref


@pytest.mark.parametrize("act", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("Hq,Hkv,D,qk_norm", [(4, 4, 16, False), (8, 2, 64, True), (5, 1, 128, True)])
@pytest.mark.parametrize("L_", [1, 6, 70])
def test_rope_append_and_attention(act, Hq, Hkv, D, qk_norm, L_):
    """q/k norm + RoPE + append, then attention over the cache, prefill (L>1) and decode (L=1, with
    and without split-KV), heterogeneous per-row offsets."""
    B, cap, max_pos = 3, 96, 128
    offs = [17, 0, 40] if L_ == 1 else [5, 0, 9]
    cos, sin, c_ref, s_ref = _rope_setup(D, max_pos, 1e6 if qk_norm else 1e4)
    nqkv = (Hq + 2 * Hkv) * D
    # existing cache contents
    kc = round_to(RNG.standard_normal((B, Hkv, cap, D)).astype(np.float32), act)
    vc = round_to(RNG.standard_normal((B, Hkv, cap, D)).astype(np.float32), act)
    qkv = round_to(RNG.standard_normal((B, L_, nqkv)).astype(np.float32), act)
    qn = round_to(1 + 0.1 * RNG.standard_normal(D).astype(np.float32), act)
    kn = round_to(1 + 0.1 * RNG.standard_normal(D).astype(np.float32), act)
    # ---- oracle
    q = qkv[..., :Hq * D].reshape(B, L_, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].reshape(B, L_, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].reshape(B, L_, Hkv, D).transpose(0, 2, 1, 3)
    if qk_norm:
        q, _ = ref_model.rms_norm(q, act, qn, act, 1e-6)
        k, _ = ref_model.rms_norm(k, act, kn, act, 1e-6)
    pos = np.array([[o + t for t in range(L_)] for o in offs])
    q = ref_model.rope(q.transpose(0, 2, 1, 3), act, pos, c_ref, s_ref)
    k = ref_model.rope(k.transpose(0, 2, 1, 3), act, pos, c_ref, s_ref)
    kc_ref, vc_ref = kc.copy(), vc.copy()
    for b in range(B):
        kc_ref[b, :, offs[b]:offs[b] + L_] = k[b]
        vc_ref[b, :, offs[b]:offs[b] + L_] = v[b]
    want = np.zeros((B, L_, Hq * D), np.float32)
    for b in range(B):
        n = offs[b] + L_
        mask = ref_model.create_additive_causal_mask_variable(L_, [offs[b]], n)
        o, _ = ref_model.sdpa(q[b:b + 1], kc_ref[b:b + 1, :, :n], vc_ref[b:b + 1, :, :n], D ** -0.5, mask, act, act)
        want[b] = o[0].transpose(1, 0, 2).reshape(L_, Hq * D)
    # ---- device
    s = attn_shape(B, L_, Hq, Hkv, D, act, act, 0, cap)
    qkv_d, kc_d, vc_d = dev(qkv.reshape(B * L_, nqkv), act), dev(kc, act), dev(vc, act)
    q_d = torch.zeros((B * L_, Hq * D), dtype=qkv_d.dtype, device="cuda")
    off_d = dev_i32(offs)
    qn_d, kn_d = dev(qn, act), dev(kn, act)
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_rope_append(C.byref(s), ptr(qkv_d), ptr(q_d), ptr(kc_d), ptr(vc_d), ptr(off_d),
                                      ptr(qn_d) if qk_norm else None, ptr(kn_d) if qk_norm else None, 1e-6,
                                      ptr(cos), ptr(sin), max_pos))
    _assert_close(host(q_d).reshape(B, L_, Hq, D).transpose(0, 2, 1, 3), q, act, scale=2.0)
    _assert_close(host(kc_d), kc_ref, act, scale=2.0)
    assert np.array_equal(host(vc_d), vc_ref)
    # attention reads the device's own cache; compare against the oracle with ITS cache
    for nsplit in ([1] if L_ > 1 else [1, 3]):
        out = torch.zeros((B * L_, Hq * D), dtype=qkv_d.dtype, device="cuda")
        part = torch.zeros((B * L_ * Hq * nsplit * (D + 2),), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        L.check(L.lib().mi_op_attention(C.byref(s), ptr(q_d), ptr(kc_d), ptr(vc_d), ptr(off_d), ptr(out),
                                        float(D ** -0.5), nsplit, ptr(part)))
        got = host(out).reshape(B, L_, Hq * D)
        if act == "float32":
            assert np.allclose(got, want, rtol=1e-4, atol=2e-5), np.abs(got - want).max()
        else:
            assert close_frac(got, want, act, atol=1e-3) <= 0.02, close_frac(got, want, act, atol=1e-3)
        if L_ > 1 and (D >= 32 if act != "float32" else D % 64 == 0):
            # the LDS-DMA kernel (default) against the register-staged one it replaced: same images, same MFMA order
            out0 = torch.zeros_like(out)
            os.environ["MI_ATTN_PREFILL_DMA"] = "0"
            try:
                L.check(L.lib().mi_op_attention(C.byref(s), ptr(q_d), ptr(kc_d), ptr(vc_d), ptr(off_d), ptr(out0),
                                                float(D ** -0.5), nsplit, ptr(part)))
                torch.cuda.synchronize()
            finally:
                del os.environ["MI_ATTN_PREFILL_DMA"]
            assert torch.equal(out, out0), "LDS-DMA prefill attention differs from the register-staged kernel"


@pytest.mark.parametrize("act", ["bfloat16", "float16", "float32"])
@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (4, 4, 64), (5, 1, 32), (8, 1, 128)])
def test_prefill_attention_many_blocks(act, Hq, Hkv, D):
    """Prefill attention over several hundred keys (the K / V ring of three buffers goes round many times), ragged offsets,
    against the oracle on sampled queries and bit for bit against the register-staged kernel."""
    if act == "float32" and D % 64 != 0:
        pytest.skip("float32 caches: the matrix-core prefill kernels take head_dim 64 / 128")
    B, L_, cap = 3, 150, 512
    offs = [0, 333, 97]
    rng = np.random.default_rng(4242)        # (its own stream: the module's RNG feeds the tests below in file order)
    kc = round_to(rng.standard_normal((B, Hkv, cap, D)).astype(np.float32), act)
    vc = round_to(rng.standard_normal((B, Hkv, cap, D)).astype(np.float32), act)
    q = round_to(rng.standard_normal((B, L_, Hq, D)).astype(np.float32), act)
    s = attn_shape(B, L_, Hq, Hkv, D, act, act, 0, cap)
    q_d, kc_d, vc_d, off_d = dev(q.reshape(B * L_, Hq * D), act), dev(kc, act), dev(vc, act), dev_i32(offs)
    part = torch.zeros((16,), dtype=torch.float32, device="cuda")
    outs = {}
    for mode in ("1", "0"):
        out = torch.zeros((B * L_, Hq * D), dtype=q_d.dtype, device="cuda")
        os.environ["MI_ATTN_PREFILL_DMA"] = mode
        try:
            torch.cuda.synchronize()
            L.check(L.lib().mi_op_attention(C.byref(s), ptr(q_d), ptr(kc_d), ptr(vc_d), ptr(off_d), ptr(out),
                                            float(D ** -0.5), 1, ptr(part)))
            torch.cuda.synchronize()
        finally:
            del os.environ["MI_ATTN_PREFILL_DMA"]
        outs[mode] = out
    assert torch.equal(outs["1"], outs["0"]), "LDS-DMA prefill attention differs from the register-staged kernel"
    got = host(outs["1"]).reshape(B, L_, Hq, D)
    if act == "float32":
        # float32 caches: the default kernel multiplies two-term bf16 splits of q, K, P and V on the bf16 matrix core (three
        # MFMAs per product, 16+ mantissa bits per operand); MI_ATTN_PREFILL_F32_EXACT=1 is the exact-product kernel on
        # v_mfma_f32_16x16x4_f32.  The two must agree to the dropped lo.lo terms (2^-16 of a product) ...
        exact = torch.zeros_like(outs["1"])
        os.environ["MI_ATTN_PREFILL_F32_EXACT"] = "1"
        try:
            L.check(L.lib().mi_op_attention(C.byref(s), ptr(q_d), ptr(kc_d), ptr(vc_d), ptr(off_d), ptr(exact),
                                            float(D ** -0.5), 1, ptr(part)))
            torch.cuda.synchronize()
        finally:
            del os.environ["MI_ATTN_PREFILL_F32_EXACT"]
        d = (outs["1"] - exact).abs()
        assert 0.0 < float(d.max()) <= 2e-5 and float(d.mean()) <= 2e-6, (float(d.max()), float(d.mean()))
        # ... and the exact kernel holds a 10 x tighter bound against the oracle than the common one below
        ge = host(exact).reshape(B, L_, Hq, D)
        for b in range(B):
            for t in (0, 16, 77, L_ - 1):
                n = offs[b] + t + 1
                o, _ = ref_model.sdpa(q[b:b + 1, t:t + 1].transpose(0, 2, 1, 3), kc[b:b + 1, :, :n], vc[b:b + 1, :, :n], D ** -0.5, None, act, act)
                assert np.allclose(ge[b, t], o[0, :, 0], rtol=1e-5, atol=2e-6), (b, t, np.abs(ge[b, t] - o[0, :, 0]).max())
    for b in range(B):
        for t in (0, 1, 15, 16, 31, 32, 77, L_ - 1):
            n = offs[b] + t + 1
            o, _ = ref_model.sdpa(q[b:b + 1, t:t + 1].transpose(0, 2, 1, 3), kc[b:b + 1, :, :n], vc[b:b + 1, :, :n], D ** -0.5, None, act, act)
            if act == "float32":
                assert np.allclose(got[b, t], o[0, :, 0], rtol=1e-4, atol=2e-5), (b, t)
            else:
                assert close_frac(got[b, t], o[0, :, 0], act, atol=1e-3) <= 0.02, (b, t)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("act", ["bfloat16", "float16", "float32"])
@pytest.mark.parametrize("Hq,Hkv,D,qk_norm", [(8, 2, 128, False), (5, 1, 128, True), (4, 4, 64, True), (16, 2, 32, False)])
def test_fused_decode_attention(act, Hq, Hkv, D, qk_norm, variant):
    """mi_op_attention_decode: q/k norm + RoPE + KV append + attention + split combine in one launch,
    MFMA (variant 0, 16-bit caches) and VALU (variant 1) kernels: ragged per-row context lengths up to
    eight 256-key rounds, 1 / 3 / 4 / 8 splits; the cache must receive exactly the new K / V row."""
    # (float32 caches: variant 0 is the v_mfma_f32_16x16x4_f32 kernel for head_dim 64 / 128 -- exact float32 products, held to
    # a 10 x tighter bound below -- and the VALU kernel otherwise.  A form on two-term bf16 operands was built, passed this
    # test as variant 0 and measured slower: attn_decode.hip, -DMI_ATTN_DECODE_SPLIT_BUILD.)
    # 1023 / 1024 / 1100: the bench's regime (one full 4 x 256-key pass, then a second, nearly empty round per
    # workgroup); 2047: eight rounds
    B, cap, max_pos = 8, 2064, 2112
    offs = [700, 0, 37, 300, 1023, 1024, 1100, 2047]
    cos, sin, c_ref, s_ref = _rope_setup(D, max_pos, 1e6 if qk_norm else 1e4)
    nqkv = (Hq + 2 * Hkv) * D
    kc = round_to(RNG.standard_normal((B, Hkv, cap, D)).astype(np.float32), act)
    vc = round_to(RNG.standard_normal((B, Hkv, cap, D)).astype(np.float32), act)
    qkv = round_to(RNG.standard_normal((B, 1, nqkv)).astype(np.float32), act)
    qn = round_to(1 + 0.1 * RNG.standard_normal(D).astype(np.float32), act)
    kn = round_to(1 + 0.1 * RNG.standard_normal(D).astype(np.float32), act)
    q = qkv[..., :Hq * D].reshape(B, 1, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hkv) * D].reshape(B, 1, Hkv, D)
    v = qkv[..., (Hq + Hkv) * D:].reshape(B, 1, Hkv, D).transpose(0, 2, 1, 3)
    if qk_norm:
        q, _ = ref_model.rms_norm(q, act, qn, act, 1e-6)
        k, _ = ref_model.rms_norm(k, act, kn, act, 1e-6)
    pos = np.array([[o] for o in offs])
    q = ref_model.rope(q.transpose(0, 2, 1, 3), act, pos, c_ref, s_ref)
    k = ref_model.rope(k.transpose(0, 2, 1, 3), act, pos, c_ref, s_ref)
    kc_ref, vc_ref = kc.copy(), vc.copy()
    want = np.zeros((B, Hq * D), np.float32)
    for b in range(B):
        kc_ref[b, :, offs[b]:offs[b] + 1] = k[b]
        vc_ref[b, :, offs[b]:offs[b] + 1] = v[b]
        n = offs[b] + 1
        o, _ = ref_model.sdpa(q[b:b + 1], kc_ref[b:b + 1, :, :n], vc_ref[b:b + 1, :, :n], D ** -0.5, None, act, act)
        want[b] = o[0].transpose(1, 0, 2).reshape(Hq * D)
    s = attn_shape(B, 1, Hq, Hkv, D, act, act, 0, cap)
    qkv_d, off_d = dev(qkv.reshape(B, nqkv), act), dev_i32(offs)
    qn_d, kn_d = dev(qn, act), dev(kn, act)
    for nsplit in (1, 3, 4, 8):
        kc_d, vc_d = dev(kc, act), dev(vc, act)
        out = torch.zeros((B, Hq * D), dtype=qkv_d.dtype, device="cuda")
        part = torch.zeros((B * Hq * nsplit * (D + 2),), dtype=torch.float32, device="cuda")
        ctr = torch.zeros((B * Hkv,), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        L.check(L.lib().mi_op_attention_decode(C.byref(s), ptr(qkv_d), ptr(kc_d), ptr(vc_d), ptr(off_d),
                                               ptr(qn_d) if qk_norm else None, ptr(kn_d) if qk_norm else None, 1e-6,
                                               ptr(cos), ptr(sin), ptr(out), float(D ** -0.5), 0, nsplit, ptr(part),
                                               ptr(ctr), variant, 1, None))
        _assert_close(host(kc_d), kc_ref, act, scale=2.0)
        assert np.array_equal(host(vc_d), vc_ref)
        assert not ctr.cpu().numpy().any()                 # tickets are handed back for the next launch
        got = host(out)
        if act == "float32" and variant == 0 and D % 64 == 0:
            assert np.allclose(got, want, rtol=1e-5, atol=2e-6), np.abs(got - want).max()
        elif act == "float32":
            assert np.allclose(got, want, rtol=1e-4, atol=2e-5), np.abs(got - want).max()
        else:
            assert close_frac(got, want, act, atol=2e-3) <= 0.02, (nsplit, close_frac(got, want, act, atol=2e-3))


def _run_sampler(lg, temp, top_p, u, k=0):
    B, V = lg.shape
    t = dev(lg)
    ud = dev(np.asarray(u, np.float32))
    toks = torch.zeros(B, dtype=torch.int32, device="cuda")
    lp = torch.zeros(B, dtype=torch.float32, device="cuda")
    p0 = torch.zeros(B, dtype=torch.float32, device="cuda")
    ki = torch.zeros((B, max(k, 1)), dtype=torch.int32, device="cuda")
    kl = torch.zeros((B, max(k, 1)), dtype=torch.float32, device="cuda")
    st = torch.zeros((B, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_sample(ptr(t), B, V, temp, top_p, ptr(ud), k, ptr(toks), ptr(lp), ptr(p0), ptr(ki), ptr(kl), ptr(st)))
    return toks.cpu().numpy(), lp.cpu().numpy(), p0.cpu().numpy(), ki.cpu().numpy(), kl.cpu().numpy()


@pytest.mark.parametrize("V", [37, 4096, 151936])
def test_sampler_greedy_logprobs_topk(V):
    B = 4
    lg = RNG.standard_normal((B, V)).astype(np.float32) * 2
    lg = round_to(lg, "bfloat16")                         # many exact ties, like a bf16 lm_head
    lg[1, 5] = lg[1].max(); lg[1, 3] = lg[1].max()        # explicit tie: lowest index wins
    want = ref_sample.sample(lg, temp=0.0)
    toks, lp, p0, ki, kl = _run_sampler(lg, 0.0, 1.0, np.zeros(B), k=5)
    assert np.array_equal(toks, want["tokens"][:, 0])
    assert np.allclose(lp, want["logprobs"], atol=1e-4)
    assert np.allclose(p0, want["probs"][:, 0], rtol=1e-3, atol=1e-7)
    lsm = want["log_softmax"]
    for b in range(B):
        order = np.lexsort((np.arange(V), -lg[b]))[:5]     # descending logit, ascending id
        assert np.array_equal(ki[b], order)
        assert np.allclose(kl[b], lsm[b, order], atol=1e-4)


def _oracle_candidates(lg_row, temp, top_p):
    """The oracle's candidate list in draw order (descending probability, ties by ascending id) with float64 masses."""
    if top_p < 1.0:
        ids, pr = ref_sample.top_p_candidates(lg_row, top_p, temp)
        return np.asarray(ids), np.asarray(pr, dtype=np.float64)
    x = lg_row.astype(np.float64) / float(temp)
    p = np.exp(x - x.max())
    ids = np.lexsort((np.arange(len(p)), -p))
    return ids, p[ids]


def _nucleus_cut_slack(lg_row, temp, top_p, n):
    """The nucleus itself is cut at top_p * Z: if the full distribution's cumulative at the cut lies within the same
    arithmetic's worst-case error (+ the float32 top_p) of top_p, the device may keep one candidate more or fewer, which
    renormalises every edge by that candidate's share.  -> that share (0.0 when the cut is not that close)."""
    if top_p >= 1.0:
        return 0.0
    xs = lg_row.astype(np.float64) / float(temp)
    pf = np.exp(xs - xs.max())
    order = np.lexsort((np.arange(len(pf)), -pf))
    pf = pf[order] / pf.sum()
    cf = np.cumsum(pf)
    ef = (5.5 * np.abs(np.log(np.maximum(pf / pf[0], 1e-300))) + 2.0) * 2.0 ** -24
    slack = float(np.sum(pf * ef)) + abs(float(np.float32(top_p)) - top_p) + (len(pf) + 1) * 2.0 ** -40
    near = [k for k in (n - 2, n - 1) if 0 <= k < len(cf) and abs(cf[k] - top_p) <= slack]
    if not near:
        return 0.0
    return float(pf[min(n, len(pf) - 1)] / cf[n - 1]) + float(pf[n - 1] / cf[n - 1])


def _sampler_edge_bound(lg_row, temp, top_p, ids, pr, j, u):
    """How far the device's boundary between candidates j and j + 1 may lie from the oracle's cumulative c_j (DESIGN 2,
    "sampler boundary bound"), from the kernel's arithmetic (misc.hip mass_fx / sample_kernel):
      * candidate i has the mass floor(2^40 * __expf((l_i - max) * (1/T))), __expf(a) = v_exp_f32(a * log2(e)).  The
        float32 roundings on the ARGUMENT -- l - max, 1/T, the product, log2(e), that product; T itself is a float32 on the
        device and a double in the oracle -- are 5.5 half-ulps = 5.5 * 2^-24 relative, i.e. a RELATIVE mass error of
        |a_i| * 5.5 * 2^-24 (a_i in nats), plus one ulp (2^-23) of v_exp_f32: eps_i = (5.5 |a_i| + 2) 2^-24;
      * c'_j - c_j = ((1 - c_j) sum_{i<=j} p_i e_i - c_j sum_{i>j} p_i e_i) / (1 + sum p_i e_i), |e_i| <= eps_i: the
        worst case puts every rounding below the edge one way and every one above it the other way;
      * truncation to 2^-40 fixed point: at most one unit per candidate and one for u * Z, against Z >= 2^40 (the
        largest mass is exactly 2^40);
      * u reaches the device as a float32: half an ulp of u.
    A function of the candidates' masses, the cumulative position and their number -- ~1e-7 for these rows; the measured
    edges (tools/debug/sampler_edge_probe.py, 136 boundaries) lie within 3.6e-8, i.e. the float32 u alone."""
    x = lg_row.astype(np.float64)
    a = np.abs((x[ids] - x.max()) / float(temp))
    p = np.asarray(pr, np.float64) / np.sum(pr)
    eps = (5.5 * a + 2.0) * 2.0 ** -24
    c = float(np.cumsum(p)[j])
    below, above = float(np.sum((p * eps)[: j + 1])), float(np.sum((p * eps)[j + 1:]))
    mass = ((1.0 - c) * below + c * above) / (1.0 - below - above)
    bound = mass + (len(ids) + 1) * 2.0 ** -40 + 0.5 * float(np.spacing(np.float32(u)))
    share = _nucleus_cut_slack(lg_row, temp, top_p, len(ids))
    return bound + share
    return bound


@pytest.mark.parametrize("V", [64, 5000, 32000, 151936])
@pytest.mark.parametrize("temp,top_p", [(1.0, 0.9), (0.7, 0.5), (1.3, 1.0), (1.0, 0.05)])
def test_sampler_top_p_injected_uniforms(V, temp, top_p):
    """Same uniforms -> same tokens as the oracle's inverse-CDF pick over the reference's candidate
    order; and the picked token always lies inside the oracle's nucleus.  V = 32000 uses logits rounded to
    bfloat16 (what a 16-bit lm_head produces): hundreds of ids share one value, and the pick inside such a tie
    group goes by ascending id."""
    B = 8
    lg = (RNG.standard_normal((B, V)) * (1.3 if V == 32000 else 3)).astype(np.float32)
    if V == 32000:
        lg = round_to(lg, "bfloat16")
    u = RNG.random(B)
    want = ref_sample.sample(lg, temp=temp, top_p=top_p, uniforms=u)
    toks, lp, p0, _, _ = _run_sampler(lg, temp, top_p, u)
    mism = 0
    for b in range(B):
        if top_p < 1.0:
            ids, pr = ref_sample.top_p_candidates(lg[b], top_p, temp)
            assert toks[b] in set(ids.tolist())
        if toks[b] != want["tokens"][b, 0]:
            # The draw is u against a cumulative distribution; the device sums 2^-40 fixed-point masses of __expf, the
            # oracle float64 probabilities.  They may only disagree when u falls within that arithmetic's error of a
            # boundary between two neighbouring candidates: the DERIVED bound of _sampler_edge_bound.
            ids, pr = _oracle_candidates(lg[b], temp, top_p)
            cum = np.cumsum(pr / np.sum(pr))
            rw, rg = int(np.where(ids == want["tokens"][b, 0])[0][0]), int(np.where(ids == toks[b])[0][0])
            j = min(rw, rg)
            bound = _sampler_edge_bound(lg[b], temp, top_p, ids, pr, j, float(u[b]))
            assert abs(rw - rg) == 1 and abs(float(u[b]) - cum[j]) <= bound, (b, rw, rg, float(u[b]), cum[j], bound)
            mism += 1
    assert mism <= 1, (toks, want["tokens"][:, 0])
    assert np.allclose(lp, want["log_softmax"][np.arange(B), toks], atol=1e-4)


@pytest.mark.parametrize("V,std,bf", [(5000, 3.0, False), (32000, 1.3, True), (151936, 3.0, False)])
@pytest.mark.parametrize("temp,top_p", [(1.0, 0.9), (0.7, 0.5), (1.3, 1.0), (1.0, 0.05)])
def test_sampler_edges_within_derived_bound(V, std, bf, temp, top_p):
    """The derived boundary bound, exercised on every run: for boundaries spread over each row's candidate list, a
    uniform placed just beyond (before) the oracle's cumulative edge by the bound must draw the candidate after (before)
    the edge.  One launch: 8 boundaries x 2 sides x 4 logits rows."""
    rng = np.random.default_rng(977 + V + int(100 * temp))
    rows, us, wants = [], [], []
    for _ in range(4):
        for _try in range(50):                     # rows whose nucleus cut is ambiguous (see _nucleus_cut_slack) are redrawn
            lg = (rng.standard_normal(V) * std).astype(np.float32)
            if bf:
                lg = round_to(lg, "bfloat16")
            ids, pr = _oracle_candidates(lg, temp, top_p)
            if _nucleus_cut_slack(lg, temp, top_p, len(ids)) == 0.0:
                break
        else:
            pytest.fail("no row with an unambiguous nucleus cut in 50 draws")
        cum = np.cumsum(pr / np.sum(pr))
        n = len(ids)
        if n < 2:
            continue
        pn = pr / np.sum(pr)
        for q in (0.0, 0.05, 0.15, 0.3, 0.5, 0.7, 0.9, 0.98):         # cumulative positions (q = 0: the first boundary)
            j = min(int(np.searchsorted(cum, q)), n - 2)
            if j < 0:
                continue
            bound = _sampler_edge_bound(lg, temp, top_p, ids, pr, j, float(cum[j]))
            assert bound < 2e-6, (V, temp, top_p, j, bound)            # the derivation is not a licence: rows like these give ~1e-7
            if min(pn[j], pn[j + 1]) <= 2.0 * bound:                   # a candidate narrower than the bound cannot be targeted
                continue
            for side, want in ((-1.0, ids[j]), (+1.0, ids[j + 1])):
                uu = float(cum[j]) + side * bound
                if 0.0 <= uu < 1.0:
                    rows.append(lg), us.append(uu), wants.append(int(want))
    if not us:
        pytest.skip("every row's nucleus is a single candidate")
    toks, _lp, _p0, _, _ = _run_sampler(np.stack(rows), temp, top_p, np.asarray(us))
    bad = [(i, us[i], int(toks[i]), wants[i]) for i in range(len(us)) if int(toks[i]) != wants[i]]
    assert not bad, bad[:5]
