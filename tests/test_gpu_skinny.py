"""Kernel-level parity (-m gpu) of gemm_skinny.hip -- the split-K weight-streaming GEMM the engine uses for 17..64
activation rows (decode steps of large batches, BASELINE configs 4 / 5) -- against the oracle's matmul, through
mi_op_gemm_skinny (include/mi355_ops.h).  Tolerances as in test_gpu_kernels.py."""
import numpy as np
import pytest
import torch

from oracle import ref_quant
from oracle.numerics import matmul_nt, round_to

pytestmark = pytest.mark.gpu

from mlx_parallm_amd import _lib as L  # noqa: E402
from gpu_helpers import dev, dev_u32, gemm_skinny, gemv, host, op_linear, q4_force, to_tiled  # noqa: E402
from test_gpu_kernels import _assert_close  # noqa: E402

RNG = np.random.default_rng(4321)


def _weight(kind, N, K, rng=None):
    rng = RNG if rng is None else rng
    if kind in ("bf16", "f16"):
        dt = {"bf16": "bfloat16", "f16": "float16"}[kind]
        w = round_to(rng.standard_normal((N, K)).astype(np.float32) * 0.05, dt)
        wd = dev(w, dt)
        ol, keep = op_linear(kind, N, K, wd), [wd]
        assert to_tiled(ol, keep)
        return ol, w, keep
    sdt = {"bf16": "bfloat16", "f16": "float16"}[kind.split("_")[1]]
    bits = 8 if kind.startswith("q8") else 4
    w = rng.standard_normal((N, K)).astype(np.float32) * 0.05
    packed, scales, biases = ref_quant.quantize(round_to(w, sdt), 64, bits, sdt)
    pd, sd, bd = dev_u32(packed), dev(scales, sdt), dev(biases, sdt)
    ol, keep = op_linear(kind, N, K, pd, sd, bd), [pd, sd, bd]
    assert to_tiled(ol, keep)
    return ol, ref_quant.dequantize(packed, scales, biases, 64, bits), keep


KINDS = [("bfloat16", "bf16"), ("float16", "f16"), ("bfloat16", "q4_bf16"), ("float16", "q4_f16"),
         ("bfloat16", "q8_bf16"), ("float16", "q8_f16")]


@pytest.mark.parametrize("act,kind", KINDS)
@pytest.mark.parametrize("M,N,K,ksplit", [
    (17, 80, 128, 1),        # one partial chunk, 5 tiles (3 spare waves)
    (32, 256, 512, 2),       # two tile groups x two K slices
    (40, 144, 4608, 5),      # 18 chunks in unequal slices, 48-row fragments
    (64, 1040, 1024, 0),     # cost model's split, 65 tiles
    (33, 128, 384, 3),       # last chunk half full (K % 256 = 128)
    (80, 272, 1024, 2),      # 96-row fragments (6 tiles)
    (128, 144, 768, 0),      # 128-row fragments (8 tiles); int4: four 32-row slabs
])
def test_store_matches_oracle_and_is_deterministic(act, kind, M, N, K, ksplit):
    ol, wdense, keep = _weight(kind, N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    xd = dev(x, act)
    outs = []
    for _ in range(2):
        out = torch.full((M + 2, N), 7.0, dtype=xd.dtype, device="cuda")
        used, _ = gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N, ksplit=ksplit)
        assert used >= 1 and (ksplit == 0 or used == ksplit)
        outs.append(host(out))
    assert np.array_equal(outs[0], outs[1])                       # slice-order reduction: run-to-run identical
    assert np.all(outs[0][M:] == 7.0)                             # rows past M are never written
    _assert_close(outs[0][:M], round_to(matmul_nt(x, wdense), act), act)


@pytest.mark.parametrize("act,kind", KINDS)
def test_epilogues(act, kind):
    M = 48
    # residual add (llama.py:188,190), K split over 4 workgroups
    N, K = 192, 2048
    ol, wdense, keep = _weight(kind, N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32) * 0.5, act)
    h = round_to(RNG.standard_normal((M, N)).astype(np.float32), act)
    want = round_to(h + round_to(matmul_nt(x, wdense), act), act)
    xd, hd = dev(x, act), dev(h, act)
    gemm_skinny(ol, xd, M, act, epi=L.EPI_RESID, resid=hd, ldo=N, ksplit=4)
    _assert_close(host(hd), want, act, scale=4.0)
    # SwiGLU over a fused gate|up matrix (llama.py:165)
    I, K = 176, 768
    ol, wdense, keep = _weight(kind, 2 * I, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    g = round_to(matmul_nt(x, wdense[:I]), act)
    u = round_to(matmul_nt(x, wdense[I:]), act)
    sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), act)
    want = round_to(round_to(g * sig, act) * u, act)
    xd = dev(x, act)
    for ks in (1, 3):
        out = torch.zeros((M, I), dtype=xd.dtype, device="cuda")
        gemm_skinny(ol, xd, M, act, epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I, ksplit=ks)
        _assert_close(host(out), want, act)
    # float32 logits (lm_head)
    N = 208
    ol, wdense, keep = _weight(kind, N, K)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE_F32, out=out, ldo=N, ksplit=2)
    _assert_close(host(out), round_to(matmul_nt(x, wdense), act), act)


# ---- gemm_q4.hip (round 4): int4 weights above 16 rows -- x prepared once per launch (fragment-major + group sums, RMSNorm
# applied there), K split over the waves of a workgroup, two tiles per wave, row slabs, optional K slices over workgroups.
# Plans of every kind the host may choose are forced here: (row tiles per workgroup, tile units x K lanes = compute waves,
# K slices over workgroups, staging waves).
RNG_Q4 = np.random.default_rng(4322)       # (a stream of its own: the module stream keeps feeding the older tests the data they had)
Q4_PLANS = [(1, 1, 8, 1, 4), (1, 8, 1, 1, 2), (2, 2, 4, 1, 4), (2, 4, 2, 2, 2), (2, 8, 1, 1, 2), (3, 4, 2, 1, 4), (4, 4, 2, 1, 4),
            (4, 8, 1, 3, 2), (2, 1, 4, 1, 4), (2, 5, 2, 1, 2), (4, 2, 2, 2, 4), (2, 3, 2, 1, 2), (4, 10, 1, 1, 2)]


@pytest.mark.parametrize("act,kind", [("bfloat16", "q4_bf16"), ("float16", "q4_f16")])
@pytest.mark.parametrize("mt,tw,kw,ks,ns", Q4_PLANS)
def test_q4_plans_match_oracle(act, kind, mt, tw, kw, ks, ns):
    """(M, N, K): ragged rows (not a multiple of 16), an odd number of tiles, K blocks that do not divide by the K lanes,
    more than one tile group."""
    for M, N, K in [(17, 80, 640), (61, 272, 1152), (64, 2064, 1024), (100, 144, 896), (128, 96, 2560)]:
        if (K // 128) // ks < kw or mt > (M + 15) // 16 or mt > 2:
            continue                                             # (the host never makes such a plan: fewer blocks than K lanes; 3 / 4 row tiles are SwiGLU-only)
        ol, wdense, keep = _weight(kind, N, K, RNG_Q4)
        x = round_to(RNG_Q4.standard_normal((M, K)).astype(np.float32), act)
        xd = dev(x, act)
        outs = []
        for _ in range(2):
            out = torch.full((M + 2, N), 7.0, dtype=xd.dtype, device="cuda")
            gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N, ksplit=q4_force(mt, tw, kw, ks, ns))
            outs.append(host(out))
        assert np.array_equal(outs[0], outs[1]), (M, N, K)        # fixed reduction orders: run-to-run identical
        assert np.all(outs[0][M:] == 7.0)
        _assert_close(outs[0][:M], round_to(matmul_nt(x, wdense), act), act)


@pytest.mark.parametrize("act,kind", [("bfloat16", "q4_bf16"), ("float16", "q4_f16")])
@pytest.mark.parametrize("mt,tw,kw,ks,ns", [(2, 4, 2, 1, 4), (4, 4, 2, 1, 4), (2, 2, 4, 2, 4), (1, 8, 1, 1, 2), (3, 4, 2, 1, 4)])
def test_q4_norm_prologue_and_epilogues(act, kind, mt, tw, kw, ks, ns):
    M, eps = max(48, 16 * mt), 1e-5
    from oracle import ref_model

    def normed(x, nw):
        return ref_model.rms_norm(x, act, nw, act, eps)[0]

    # RMSNorm in front (llama.py:175-177) + residual add (llama.py:188,190)
    N, K = 192, 2048
    ol, wdense, keep = _weight(kind, N, K, RNG_Q4)
    x = round_to(RNG_Q4.standard_normal((M, K)).astype(np.float32) * 0.5, act)
    nw = round_to(1.0 + 0.1 * RNG_Q4.standard_normal(K).astype(np.float32), act)
    h = round_to(RNG_Q4.standard_normal((M, N)).astype(np.float32), act)
    want = round_to(h + round_to(matmul_nt(normed(x, nw), wdense), act), act)
    xd, hd, nwd = dev(x, act), dev(h, act), dev(nw, act)
    if mt <= 2:                                                  # (3 / 4 row tiles per workgroup: the SwiGLU instantiation only)
        gemm_skinny(ol, xd, M, act, epi=L.EPI_RESID, resid=hd, ldo=N, ksplit=q4_force(mt, tw, kw, ks, ns), norm_w=nwd, eps=eps)
        _assert_close(host(hd), want, act, scale=4.0)
    # SwiGLU over a fused gate|up matrix (llama.py:165), RMSNorm in front
    I, K = 176, 768
    ol, wdense, keep = _weight(kind, 2 * I, K, RNG_Q4)
    x = round_to(RNG_Q4.standard_normal((M, K)).astype(np.float32), act)
    nw = round_to(1.0 + 0.1 * RNG_Q4.standard_normal(K).astype(np.float32), act)
    xn = normed(x, nw)
    g = round_to(matmul_nt(xn, wdense[:I]), act)
    u = round_to(matmul_nt(xn, wdense[I:]), act)
    sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), act)
    want = round_to(round_to(g * sig, act) * u, act)
    xd, nwd = dev(x, act), dev(nw, act)
    out = torch.zeros((M, I), dtype=xd.dtype, device="cuda")
    if (K // 128) // ks >= kw:
        gemm_skinny(ol, xd, M, act, epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I, ksplit=q4_force(mt, tw, kw, ks, ns), norm_w=nwd, eps=eps)
        _assert_close(host(out), want, act)
    # float32 logits (lm_head), no norm
    N = 208
    ol, wdense, keep = _weight(kind, N, K, RNG_Q4)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    if (K // 128) // ks >= kw and mt <= 2:
        gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE_F32, out=out, ldo=N, ksplit=q4_force(mt, tw, kw, ks, ns))
        _assert_close(host(out), round_to(matmul_nt(x, wdense), act), act)


def test_rows_agree_with_the_16_row_kernel():
    """The same rows through gemv_mfma.hip (<= 16 rows per launch): equal up to the rounding of differently ordered
    float32 sums."""
    act, M, N, K = "bfloat16", 32, 512, 4096
    ol, wdense, keep = _weight("bf16", N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    xd = dev(x, act)
    a = torch.zeros((M, N), dtype=xd.dtype, device="cuda")
    b = torch.zeros((M, N), dtype=xd.dtype, device="cuda")
    gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE, out=a, ldo=N)
    for r in (0, 16):
        assert gemv(ol, xd[r:r + 16], 16, act, epi=L.EPI_STORE, out=b[r:r + 16], ldo=N)
    _assert_close(host(a), host(b), act)
    assert np.mean(host(a) == host(b)) > 0.97


# ---- through the engine: decode steps of more than 16 sequences (BASELINE configs 4 / 5: batches of 32 / 64)

from oracle import ref_generate  # noqa: E402
from mlx_parallm_amd import utils  # noqa: E402


def _prompts(cfg, B, L0, seed):
    rng = np.random.default_rng(seed)
    toks = rng.integers(3, cfg["vocab_size"], size=(B, L0))
    for b in range(B):
        toks[b, :int(rng.integers(0, L0 // 2))] = 1              # left padding (attended, quirk Q1)
    return toks.astype(np.int32)


@pytest.mark.parametrize("B", [11, 17, 40, 64, 100])
@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16"])
def test_large_batch_decode_logits_match_oracle(tiny_dirs, name, B):
    d, cfg = tiny_dirs[name]
    model = utils.load_model(d, max_positions=256)
    ref = ref_generate.load(d, max_pos=256)
    toks = _prompts(cfg, B, 6, seed=B)
    kv = model.engine.new_kv(B, capacity=16, kv_dtype="model")
    cache = ref.make_cache(B, paged=False)
    model.engine.forward(toks, kv)
    nxt = np.argmax(ref(toks, cache=cache)[:, -1], axis=-1)[:, None]
    steps = []
    for _ in range(3):
        got = model.engine.forward(nxt.astype(np.int32), kv)
        want = ref(nxt, cache=cache)[:, -1]
        assert np.abs(got - want).max() <= 0.08, np.abs(got - want).max()
        steps.append(got)
        nxt = np.argmax(want, axis=-1)[:, None]
    # the same steps through 16-row launches of the M <= 16 kernel: equal up to rounding noise, and not the same path
    model.engine.set_option("skinny_gemm", 0)
    kv2 = model.engine.new_kv(B, capacity=16, kv_dtype="model")
    cache2 = ref.make_cache(B, paged=False)
    model.engine.forward(toks, kv2)
    nxt = np.argmax(ref(toks, cache=cache2)[:, -1], axis=-1)[:, None]
    got2 = model.engine.forward(nxt.astype(np.int32), kv2)
    assert np.abs(got2 - steps[0]).max() <= 0.08
    model.engine.close()


def test_large_batch_decode_with_lora(tiny_dirs, tmp_path):
    """BASELINE config 5: int4 weights + LoRA on q / v, a decode batch above 16 rows."""
    import json

    from safetensors.torch import save_file

    d, cfg = tiny_dirs["llama_q4_bf16"]
    H, nh, nkv, D = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    rng = np.random.default_rng(11)
    w = {}
    i = cfg["num_hidden_layers"] - 1
    for key, n in (("self_attn.q_proj", nh * D), ("self_attn.v_proj", nkv * D)):
        w[f"model.layers.{i}.{key}.lora_a"] = torch.from_numpy((rng.uniform(-1, 1, (H, 16)) / np.sqrt(H)).astype(np.float32))
        w[f"model.layers.{i}.{key}.lora_b"] = torch.from_numpy(rng.standard_normal((16, n)).astype(np.float32) * 0.05)
    ad = tmp_path / "adapter"
    ad.mkdir()
    save_file(w, str(ad / "adapters.safetensors"))
    (ad / "adapter_config.json").write_text(json.dumps({
        "fine_tune_type": "lora", "num_layers": 1,
        "lora_parameters": {"rank": 16, "scale": 10.0, "dropout": 0.05, "keys": ["self_attn.q_proj", "self_attn.v_proj"]}}))
    model = utils.load_model(d, max_positions=256)
    utils.load_adapters(model, str(ad))
    ref = ref_generate.load(d, adapter_path=str(ad), max_pos=256)
    base = ref_generate.load(d, max_pos=256)
    B = 24
    toks = _prompts(cfg, B, 6, seed=3)
    kv = model.engine.new_kv(B, capacity=16, kv_dtype="model")
    cache, bcache = ref.make_cache(B, paged=False), base.make_cache(B, paged=False)
    # the prefill is 144 rows: the tile GEMM + the LoRA term added to its stored output (launch_lora_up_add)
    got0 = model.engine.forward(toks, kv)
    want0 = ref(toks, cache=cache)[:, -1]
    assert np.abs(got0 - want0).max() <= 0.08, np.abs(got0 - want0).max()
    model.engine.set_option("prefill_gemm", 0)                  # ... and the same through the 16-row launches
    kv0 = model.engine.new_kv(B, capacity=16, kv_dtype="model")
    assert np.abs(model.engine.forward(toks, kv0) - got0).max() <= 0.08
    model.engine.set_option("prefill_gemm", 1)
    nxt = np.argmax(want0, axis=-1)[:, None]
    base(toks, cache=bcache)
    got = model.engine.forward(nxt.astype(np.int32), kv)
    want = ref(nxt, cache=cache)[:, -1]
    plain = base(nxt, cache=bcache)[:, -1]
    assert np.abs(want - plain).max() > 0.5                     # the adapter really changes the logits
    assert np.abs(got - want).max() <= 0.08, np.abs(got - want).max()
    model.engine.close()


@pytest.mark.parametrize("act,kind", KINDS)
@pytest.mark.parametrize("M,N,K,ksplit", [(9, 80, 128, 1), (12, 256, 4608, 0), (13, 144, 384, 3), (16, 1040, 1024, 2)])
def test_16_row_instantiation(act, kind, M, N, K, ksplit):
    """9..16 rows: the engine hands over from gemv_mfma.hip at 9 rows."""
    ol, wdense, keep = _weight(kind, N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    xd = dev(x, act)
    out = torch.full((M + 1, N), 7.0, dtype=xd.dtype, device="cuda")
    gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N, ksplit=ksplit)
    got = host(out)
    assert np.all(got[M:] == 7.0)
    _assert_close(got[:M], round_to(matmul_nt(x, wdense), act), act)


@pytest.mark.parametrize("act,kind", [("bfloat16", "q8_bf16"), ("float16", "q8_f16")])
@pytest.mark.parametrize("M,N,K,ksplit", [(1, 80, 128, 1), (8, 256, 4608, 0), (13, 144, 384, 3), (16, 1040, 1024, 2)])
def test_int8_decode_rows(act, kind, M, N, K, ksplit):
    """int8 weights have no M <= 16 kernel of their own: every decode step runs through the 16-row instantiation."""
    ol, wdense, keep = _weight(kind, N, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    xd = dev(x, act)
    out = torch.full((M + 1, N), 7.0, dtype=xd.dtype, device="cuda")
    gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N, ksplit=ksplit)
    got = host(out)
    assert np.all(got[M:] == 7.0)
    _assert_close(got[:M], round_to(matmul_nt(x, wdense), act), act)
    # the generic kernel on the same tile-major int8 matrix (the PagedKVCache-mode path)
    out2 = torch.zeros((min(M, 8), N), dtype=xd.dtype, device="cuda")
    assert not gemv(ol, xd, min(M, 8), act, epi=L.EPI_STORE, out=out2, ldo=N)
    _assert_close(host(out2), round_to(matmul_nt(x[:8], wdense), act), act)


@pytest.mark.parametrize("B", [3, 12, 24, 40, 64])      # 16-, 32-row workgroups; 48 rows keep the norm launch (dense) / slabs of 32 (int4); 64 rows
@pytest.mark.parametrize("name", ["llama_q4_bf16", "llama_q8_f16", "llama_bf16_gqa", "qwen3_bf16"])
def test_norm_handover_between_launches(tiny_dirs, name, B):
    """RMSNorm statistics handed from the residual epilogue of one launch to the staging of the next (no rmsnorm
    launch): logits against the oracle, and against the same steps with the hand-over switched off."""
    d, cfg = tiny_dirs[name]
    model = utils.load_model(d, max_positions=256)
    ref = ref_generate.load(d, max_pos=256)
    toks = _prompts(cfg, B, 6, seed=100 + B)
    outs = {}
    for on in (1, 0):
        model.engine.set_option("norm_handover", on)
        kv = model.engine.new_kv(B, capacity=16, kv_dtype="model")
        cache = ref.make_cache(B, paged=False)
        model.engine.forward(toks, kv)
        nxt = np.argmax(ref(toks, cache=cache)[:, -1], axis=-1)[:, None]
        steps = []
        for _ in range(3):
            got = model.engine.forward(nxt.astype(np.int32), kv)
            want = ref(nxt, cache=cache)[:, -1]
            assert np.abs(got - want).max() <= 0.08, (on, np.abs(got - want).max())
            steps.append(got)
            nxt = np.argmax(want, axis=-1)[:, None]
        outs[on] = np.stack(steps)
    assert np.abs(outs[1] - outs[0]).max() <= 0.08
    model.engine.close()


@pytest.mark.parametrize("act,kind", KINDS)
@pytest.mark.parametrize("M", [72, 120])
def test_swiglu_above_64_rows(act, kind, M):
    I, K = 176, 768
    ol, wdense, keep = _weight(kind, 2 * I, K)
    x = round_to(RNG.standard_normal((M, K)).astype(np.float32), act)
    g = round_to(matmul_nt(x, wdense[:I]), act)
    u = round_to(matmul_nt(x, wdense[I:]), act)
    sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), act)
    want = round_to(round_to(g * sig, act) * u, act)
    xd = dev(x, act)
    out = torch.zeros((M, I), dtype=xd.dtype, device="cuda")
    gemm_skinny(ol, xd, M, act, epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I, ksplit=2)
    _assert_close(host(out), want, act)


@pytest.mark.parametrize("M,N,K,ksplit", [(1, 80, 128, 1), (8, 256, 4608, 0), (16, 144, 384, 3), (27, 256, 1024, 2)])
@pytest.mark.parametrize("rnd", [L.RND_NONE, L.RND_BF16])
def test_float32_activations_on_bf16_weights(M, N, K, ksplit, rnd):
    """PagedKVCache mode: float32 activations, 16-bit weights.  x = hi + mid + lo exactly, three MFMAs per
    weight fragment: a float32 dot product in another summation order; outputs float32 with the logical rounding."""
    ol, wdense, keep = _weight("bf16", N, K)
    x = RNG.standard_normal((M, K)).astype(np.float32)
    if rnd:
        x = round_to(x, "bfloat16")
    xd = dev(x, "float32")
    want = matmul_nt(x, wdense)
    if rnd:
        want = round_to(want, "bfloat16")
    out = torch.full((M + 1, N), 7.0, dtype=torch.float32, device="cuda")
    gemm_skinny(ol, xd, M, "float32", epi=L.EPI_STORE, out=out, ldo=N, ksplit=ksplit, rnd=rnd)
    got = host(out)
    assert np.all(got[M:] == 7.0)
    _assert_close(got[:M], want, "bfloat16" if rnd else "float32")
    # residual and SwiGLU epilogues in float32
    h = RNG.standard_normal((M, N)).astype(np.float32)
    hd = dev(h, "float32")
    gemm_skinny(ol, xd, M, "float32", epi=L.EPI_RESID, resid=hd, ldo=N, ksplit=ksplit, rnd=0)
    _assert_close(host(hd), h + matmul_nt(x, wdense), "float32", scale=4.0)
    if N % 32 == 0:
        I = N // 2
        g, u = matmul_nt(x, wdense[:I]), matmul_nt(x, wdense[I:])
        want = (g / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32) * u
        out = torch.zeros((M, I), dtype=torch.float32, device="cuda")
        gemm_skinny(ol, xd, M, "float32", epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I, ksplit=1, rnd=0)
        _assert_close(host(out), want, "float32")


@pytest.mark.parametrize("act,kind", [("bfloat16", "q4_bf16"), ("float16", "q4_f16")])
@pytest.mark.parametrize("M", [33, 64, 100, 128])
def test_q4_default_plan_on_wide_matrices(act, kind, M):
    """What the engine launches for a decode step of 33..128 sequences on int4 weights, with NO forced plan: the wide
    matrices (>= 1024 column tiles: gate|up, lm_head) go to gemm_q4.hip under its measured default plan (TW = 5, KW = 2,
    two staging waves; 4 row tiles per workgroup for SwiGLU, i.e. two row slabs above 64 rows; 2 otherwise), RMSNorm folded
    into the preparation pass.  Against the oracle; run-to-run identical."""
    from oracle import ref_model

    eps = 1e-6
    # SwiGLU over a fused gate|up matrix of 2 x 8192 rows (1024 tiles), RMSNorm in front
    I, K = 8192, 512
    ol, wdense, keep = _weight(kind, 2 * I, K, RNG_Q4)
    x = round_to(RNG_Q4.standard_normal((M, K)).astype(np.float32), act)
    nw = round_to(1.0 + 0.1 * RNG_Q4.standard_normal(K).astype(np.float32), act)
    xn = ref_model.rms_norm(x, act, nw, act, eps)[0]
    g = round_to(matmul_nt(xn, wdense[:I]), act)
    u = round_to(matmul_nt(xn, wdense[I:]), act)
    sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), act)
    want = round_to(round_to(g * sig, act) * u, act)
    xd, nwd = dev(x, act), dev(nw, act)
    outs = []
    for _ in range(2):
        out = torch.full((M + 1, I), 3.0, dtype=xd.dtype, device="cuda")
        used = gemm_skinny(ol, xd, M, act, epi=L.EPI_SWIGLU, out=out, ldo=I, pair_offset=I, norm_w=nwd, eps=eps)
        outs.append(host(out))
    assert np.array_equal(outs[0], outs[1])
    assert np.all(outs[0][M:] == 3.0)
    # A million outputs of a chain with TWO independently rounded inputs: g and u may each land one unit off where the
    # float32 sums of kernel and oracle straddle a rounding boundary (that is _assert_close's 2-unit bound for ONE rounded
    # sum), each moves silu(g) * u by one unit of the result, and the two roundings behind them (g * sig, then * u) can
    # turn that into one more half each: every element within 3 units, 90 % within half a unit.  Measured: 2.27 at worst.
    got = outs[0][:M]
    unit = {"bfloat16": 2.0 ** -7, "float16": 2.0 ** -10}[act] * np.maximum(np.abs(want), float(np.sqrt(np.mean(np.square(g * u)))))
    err = np.abs(got - want)
    assert np.all(err <= 3 * unit), float((err / unit).max())
    assert np.mean(err > 0.5 * unit) <= 0.10, float(np.mean(err > 0.5 * unit))
    # float32 logits over an odd number of tiles (1025), no norm
    N = 16400
    ol, wdense, keep = _weight(kind, N, K, RNG_Q4)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE_F32, out=out, ldo=N)
    _assert_close(host(out), round_to(matmul_nt(x, wdense), act), act)
