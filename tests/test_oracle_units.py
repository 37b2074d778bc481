"""Unit checks of the oracle itself (CPU): quantisation format, KV-cache semantics, masks,
sampler semantics.  The reference has no unit tests for any of these."""
import numpy as np
import pytest
import torch

from oracle import numerics, ref_model, ref_quant, ref_sample


# ---------------------------------------------------------------- affine quantisation (App. A.1)
def test_quantize_known_answer():
    # one group of 64: values 0..63 scaled so that w_min=0 -> |w_min| < |w_max|: scale<0, edge=w_max
    w = (np.arange(64, dtype=np.float32) / 63.0 * 1.5)[None, :]
    packed, scales, biases = ref_quant.quantize(w, 64, 4)
    # scale = -(1.5-0)/15 = -0.1 ; q0 = round(1.5/-0.1) = -15 != 0 -> scale = 1.5/-15 ; bias = 1.5
    assert np.isclose(scales[0, 0], -0.1, rtol=1e-6) and np.isclose(biases[0, 0], 1.5)
    q = ref_quant.unpack(packed, 4)[0]
    assert q[0] == 15 and q[63] == 0                     # codes count DOWN from the edge (negative scale)
    assert packed.shape == (1, 8) and packed.dtype == np.uint32
    # little-endian nibbles: element j of a word sits at bits [4j, 4j+4)
    assert (packed[0, 0] & 0xF) == q[0] and ((packed[0, 0] >> 28) & 0xF) == q[7]
    wh = ref_quant.dequantize(packed, scales, biases, 64, 4)
    assert np.abs(wh - w).max() <= 0.05 + 1e-6


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_quantize_oracle_vs_product_torch(bits, dtype):
    """Two independent implementations (oracle NumPy, product torch) of the same format."""
    from mlx_parallm_amd import quant as pq

    rng = np.random.default_rng(bits)
    w = numerics.round_to(rng.standard_normal((24, 256)).astype(np.float32) * 0.05, dtype)
    packed, scales, biases = ref_quant.quantize(w, 64, bits, dtype)
    tdt = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}[dtype]
    p2, s2, b2 = pq.quantize(torch.from_numpy(w).to(tdt), 64, bits)
    assert np.array_equal(packed, p2.numpy().view(np.uint32))
    assert np.array_equal(scales, s2.to(torch.float32).numpy())
    assert np.array_equal(biases, b2.to(torch.float32).numpy())
    wh = ref_quant.dequantize(packed, scales, biases, 64, bits)
    wh2 = pq.dequantize(p2, s2, b2, 64, bits).numpy()
    assert np.array_equal(wh, wh2)
    step = np.abs(scales).max()
    # the scale is re-fitted so that the edge value is exact (scale = edge / q0), which can clip the
    # other end of the range by up to one step
    assert np.abs(wh - w).max() <= 1.0 * step * 1.02 + 2e-3 * (dtype != "float32")


def test_round_bf16_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.001953125, -3.14159, 1e-40, np.inf], dtype=np.float32)
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(numerics.round_bf16(x), want)
    r = np.random.default_rng(0).standard_normal(10000).astype(np.float32)
    assert np.array_equal(numerics.round_bf16(r), torch.from_numpy(r).to(torch.bfloat16).float().numpy())


# ---------------------------------------------------------------- KV caches (base.py:42-150)
def test_batched_cache_growth_and_dtype():
    c = ref_model.RefBatchedKVCache(4, 2, batch_size=2)
    k = np.full((2, 2, 3, 4), 1.00390625, np.float32)            # needs > 8 mantissa bits
    ks, vs, dt = c.update_and_fetch(k, k, "bfloat16")
    assert dt == "bfloat16" and ks.shape == (2, 2, 3, 4) and c.keys.shape[2] == 256
    assert np.all(ks == 1.0)                                      # stored in the keys' dtype (bf16 rounds to 1.0)
    assert c.offsets == [3, 3]
    c.update_and_fetch(np.ones((2, 2, 300, 4), np.float32), np.ones((2, 2, 300, 4), np.float32), "bfloat16")
    assert c.offset == 303 and c.keys.shape[2] >= 303


def test_paged_cache_is_float32_quirk_and_per_row_offsets():
    c = ref_model.RefPagedKVCache(4, 2, batch_size=2)
    k = np.full((2, 2, 1, 4), 1.00390625, np.float32)
    ks, vs, dt = c.update_and_fetch(k, k, "bfloat16")
    assert dt == "float32"                                        # base.py:111-112
    assert np.all(ks == np.float32(1.00390625))                   # no rounding on the way in
    assert c.offsets == [1, 1] and c.offset == 0                  # base `offset` never advances (paged)
    c.reset()
    assert c.offsets == [0, 0]
    c.reset(3)
    assert c.batch_size == 3 and c.keys is None and c.offsets == [0, 0, 0]


def test_additive_causal_mask_variable():
    m = ref_model.create_additive_causal_mask_variable(3, [2, 0], 5)
    assert m.shape == (2, 3, 5)
    # row 0 (offset 2): query t sees keys <= 2+t
    assert np.array_equal(m[0] == 0, np.array([[1, 1, 1, 0, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 1]], bool))
    assert np.array_equal(m[1] == 0, np.array([[1, 0, 0, 0, 0], [1, 1, 0, 0, 0], [1, 1, 1, 0, 0]], bool))
    assert m.min() == np.float32(-1e9)
    from mlx_parallm_amd.models.base import create_additive_causal_mask_variable as prod
    assert np.array_equal(prod(3, [2, 0], 5), m)


# ---------------------------------------------------------------- sampler (utils.py:345-364, sample_utils.py)
def test_greedy_lowest_index_on_ties_and_row0_probs():
    lg = np.array([[0.0, 2.0, 2.0, 1.0], [3.0, 0.0, 0.0, 3.0]], np.float32)
    s = ref_sample.sample(lg, temp=0.0)
    assert s["tokens"].tolist() == [[1], [0]]
    p_row0 = np.exp(ref_sample.log_softmax(lg))[0]
    assert np.allclose(s["probs"][:, 0], [p_row0[1], p_row0[0]])           # quirk Q6: row 0's distribution
    assert np.allclose(s["logprobs"], [np.log(p_row0[1]), ref_sample.log_softmax(lg)[1, 0]])


def test_top_p_excludes_crossing_token_and_empty_set_fallback():
    p = np.array([0.5, 0.3, 0.15, 0.05])
    lg = np.log(p).astype(np.float32)[None]
    ids, pr = ref_sample.top_p_candidates(lg[0], 0.9, 1.0)
    assert ids.tolist() == [0, 1] and np.allclose(pr, [0.625, 0.375])       # 0.5, 0.8 <= 0.9 < 0.95
    ids, pr = ref_sample.top_p_candidates(lg[0], 0.4, 1.0)                  # top token alone > top_p (Q5)
    assert ids.tolist() == [0] and np.allclose(pr, [1.0])
    for u, want in [(0.0, 0), (0.62, 0), (0.63, 1), (0.999, 1)]:
        s = ref_sample.sample(lg, temp=1.0, top_p=0.9, uniforms=np.array([u]))
        assert s["tokens"][0, 0] == want
    s = ref_sample.sample(lg, temp=1.0, top_p=1.0, uniforms=np.array([0.96]))   # plain categorical: all tokens
    assert s["tokens"][0, 0] == 3


def test_temperature_reshapes_distribution():
    lg = np.array([[2.0, 1.0, 0.0]], np.float32)
    ids, pr = ref_sample.top_p_candidates(lg[0], 0.99, 0.5)
    want = np.exp(np.array([4.0, 2.0, 0.0]))
    want = want / want.sum()
    assert ids.tolist() == [0, 1] and np.allclose(pr, want[:2] / want[:2].sum())


def test_logit_bias_applied_before_softmax():
    lg = np.zeros((1, 4), np.float32)
    s = ref_sample.sample(lg, temp=0.0, logit_bias={2: 5.0})
    assert s["tokens"][0, 0] == 2 and s["logprobs"][0] > np.log(0.9)


# ---------------------------------------------------------------- accumulation envelope (oracle/numerics.py:set_accum)
def test_float32_accumulation_variants_bracket_the_exact_oracle():
    """The two float32-accumulating variants of the oracle (chunks of 32 combined sequentially / as a balanced tree) are
    the SAME arithmetic as the exact one up to float32 summation order: the C kernel and its NumPy restatement agree bit
    for bit, both stay within a few float32 ulps of the exactly rounded sums, and the two orders really differ."""
    rng = np.random.default_rng(5)
    x = numerics.round_bf16(rng.standard_normal((7, 4096 + 32)).astype(np.float32))
    w = numerics.round_bf16(rng.standard_normal((53, 4096 + 32)).astype(np.float32) * 0.02)
    exact = numerics.matmul_nt(x, w)
    got = {}
    for mode in ("f32_seq32", "f32_pairwise"):
        c = numerics.matmul_nt_f32(x, w, mode)
        n = numerics.matmul_nt_f32(x, w, mode, use_c=False)
        assert np.array_equal(c, n), mode
        assert np.abs(c - exact).max() <= 2e-6 * np.abs(exact).max() + 1e-6
        got[mode] = c
    assert not np.array_equal(got["f32_seq32"], got["f32_pairwise"])
    # an odd number of chunks exercises the leftover path of the binary-counter tree
    a = numerics.Accum("f32_pairwise")
    parts = [np.float32(v) for v in (1.0, 2.0 ** -24, 2.0 ** -24, 1.0, 3.0)]
    for p_ in parts:
        a.add(np.asarray([p_], np.float32))
    tree = np.float32(np.float32(np.float32(parts[0] + parts[1]) + np.float32(parts[2] + parts[3])) + parts[4])
    assert a.result()[0] == tree
    assert np.array_equal(numerics.sum_last_f32(np.ones((3, 100), np.float32), "f32_seq32"), np.full(3, 100, np.float32))


def test_accumulation_modes_on_a_whole_model(tmp_path):
    """set_accum switches every accumulation of the forward (linears, RMSNorm statistics, attention): logits of a small bf16
    model under either float32 order stay within rounding noise of the exact oracle, and the switch is restored."""
    from mlx_parallm_amd.tiny_model import build_tiny_model
    from oracle import ref_generate

    build_tiny_model(tmp_path, seed=2, vocab_size=300, hidden_size=128, layers=2, heads=4, kv_heads=2, intermediate_size=256,
                     quantize_model=False, dtype="bfloat16", tie_word_embeddings=False, with_tokenizer=False)
    ref = ref_generate.load(str(tmp_path), max_pos=64)
    toks = np.random.default_rng(0).integers(0, 300, size=(2, 40))
    exact = ref(toks, cache=ref.make_cache(2, paged=True))
    for mode in ("f32_seq32", "f32_pairwise"):
        numerics.set_accum(mode)
        try:
            got = ref(toks, cache=ref.make_cache(2, paged=True))
        finally:
            numerics.set_accum("exact")
        assert np.abs(got - exact).max() <= 2e-2 and np.sqrt(((got - exact) ** 2).mean()) <= 2e-3, mode
    assert numerics.ACCUM == "exact"
    with pytest.raises(ValueError):
        numerics.set_accum("f16")
