"""End-to-end parity (-m gpu): the product path (utils.load_model -> generate_step -> engine ->
C ABI -> HIP kernels) against the oracle's restatement of generate_step on the same checkpoint
directory and the same prompts.

Bar (BASELINE.json north_star): greedy token ids bit-exact, log-probabilities within 1e-3.
Greedy parity is checked with teacher forcing so that one near-tie cannot cascade: at every
step the device's token must equal the oracle's, EXCEPT where the oracle's own top-2 logit
margin is below the kernel's documented error bound -- such steps are counted and reported,
never hidden, and none is allowed on the float32 configurations.
"""
import numpy as np
import pytest

from oracle import ref_generate, ref_sample

pytestmark = pytest.mark.gpu

from mlx_parallm_amd import utils  # noqa: E402
from mlx_parallm_amd.engine import SampleArgs  # noqa: E402

RNG = np.random.default_rng(7)
MODEL_KV_LOGPROB_TOL = 0.1      # BatchedKVCache (16-bit KV / activations) mode; the float32-KV mode holds 1e-3

# name -> (kv dtype for the product, oracle paged flag, logit atol, logprob atol)
MODES = {
    "model": ("model", False),       # BatchedKVCache semantics: KV in the model dtype
    "float32": ("float32", True),    # PagedKVCache semantics: float32 KV, promotion after layer 0
}


def _load_pair(tiny_dirs, name, max_pos=512):
    d, cfg = tiny_dirs[name]
    model = utils.load_model(d, max_positions=max_pos)
    ref = ref_generate.load(d, max_pos=max_pos)
    return model, ref, cfg


def _left_pad_prompts(cfg, B, L0, ragged=True):
    toks = RNG.integers(3, cfg["vocab_size"], size=(B, L0))
    if ragged:
        pad = 1
        for b in range(B):
            n = int(RNG.integers(0, L0 // 2))
            toks[b, :n] = pad                           # left padding; pads ARE attended (quirk Q1)
    return toks.astype(np.int32)


def _logit_tol(cfg_name):
    if "f32" in cfg_name and "bf16" not in cfg_name:
        return 2e-4
    return 0.08          # a few bf16/f16 ulps at |logit| ~ 2-4 once rounding flips accumulate over the layers


@pytest.mark.parametrize("mode", ["model", "float32"])
@pytest.mark.parametrize("name", ["llama_q4_f32", "llama_f32", "llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16",
                                  "llama_q8_f16", "llama_f16"])
def test_prefill_and_decode_logits(tiny_dirs, name, mode):
    """model(y, cache) logits: prefill (all positions) then 3 decode steps, ragged left padding."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    kvd, paged = MODES[mode]
    B, L0 = 3, 9
    toks = _left_pad_prompts(cfg, B, L0)
    kv = model.engine.new_kv(B, capacity=32, kv_dtype=kvd)
    cache = ref.make_cache(B, paged=paged)
    got = model.engine.forward(toks, kv, all_positions=True)
    want = ref(toks, cache=cache)
    tol = _logit_tol(name) if not (mode == "float32") else max(2e-4, _logit_tol(name) / 20)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    nxt = np.argmax(want[:, -1], axis=-1)[:, None]
    for _ in range(3):
        got = model.engine.forward(nxt.astype(np.int32), kv)
        want = ref(nxt, cache=cache)[:, -1]
        assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
        nxt = np.argmax(want, axis=-1)[:, None]
    assert kv.offsets == cache[0].offsets == [L0 + 3] * B
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16", "llama_f16"])
@pytest.mark.parametrize("B,L0", [(4, 40), (5, 64)])
def test_float32_kv_prefill_through_the_tile_gemm(tiny_dirs, name, B, L0):
    """PagedKVCache mode, prompts of >= 32 rows in all: float32 activations are split exactly three ways
    (split3_rows_kernel) and go through the MFMA tile GEMM (128- and 256-row tiles) and the float32 MFMA attention;
    all-position logits and the decode steps behind them against the oracle, float32 tolerances."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    toks = _left_pad_prompts(cfg, B, L0)
    kv = model.engine.new_kv(B, capacity=L0 + 8, kv_dtype="float32")
    cache = ref.make_cache(B, paged=True)
    got = model.engine.forward(toks, kv, all_positions=True)
    want = ref(toks, cache=cache)
    tol = 4e-3          # layer 0 of this mode still rounds to the 16-bit dtype: rare 1-ulp flips (see test_gpu_golden_wide.py)
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    assert np.sqrt(((got - want) ** 2).mean()) <= 2e-4
    nxt = np.argmax(want[:, -1], axis=-1)[:, None]
    for _ in range(2):
        got = model.engine.forward(nxt.astype(np.int32), kv)
        want = ref(nxt, cache=cache)[:, -1]
        assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
        nxt = np.argmax(want, axis=-1)[:, None]
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_bf16_gqa", "qwen3_bf16"])
def test_float32_kv_prefill_two_and_three_terms_of_x(tiny_dirs, name):
    """PagedKVCache mode, dense bf16 weights: the tile GEMM multiplies [hi | lo] of x (option prefill_x_terms = 2, the default:
    16+ mantissa bits, two walks of W) or the exact [hi | mid | lo] (3).  Both against the oracle at the float32 tolerances of
    the test above; the two against each other within the 2^-17 relative error of the dropped term; and the option really
    switches the arithmetic (the results differ somewhere)."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    eng = model.engine
    B, L0 = 5, 64
    toks = _left_pad_prompts(cfg, B, L0)
    want = ref(toks, cache=ref.make_cache(B, paged=True))
    outs = {}
    for terms in (2, 3):
        eng.set_option("prefill_x_terms", terms)
        kv = eng.new_kv(B, capacity=L0 + 8, kv_dtype="float32")
        outs[terms] = eng.forward(toks, kv, all_positions=True)
        kv.close()
        assert np.abs(outs[terms] - want).max() <= 4e-3, (terms, np.abs(outs[terms] - want).max())
        assert np.sqrt(((outs[terms] - want) ** 2).mean()) <= 2e-4, terms
    eng.set_option("prefill_x_terms", 2)
    d = np.abs(outs[2] - outs[3])
    assert 0.0 < d.max() <= 2e-3 and np.sqrt((d ** 2).mean()) <= 5e-5, (d.max(), np.sqrt((d ** 2).mean()))
    # On a 64..128-wide model the exact split's float32 sums are nearly exact (rms 1.7e-7 against the oracle) and the dropped
    # third term IS the error of the two-term form (measured rms 2.2e-6): two orders under the 2e-4 this mode is held to.  At
    # production width it disappears under the float32 sums' own error and layer 0's 16-bit flips (the CPU
    # variants' mean |logprob - exact| 1.91352e-4 against 1.91362e-4).
    e2, e3 = np.sqrt(((outs[2] - want) ** 2).mean()), np.sqrt(((outs[3] - want) ** 2).mean())
    assert e2 <= 1e-5 and e3 <= 1e-5, (e2, e3)
    eng.close()


@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_f16", "qwen3_bf16"])
def test_decode_swiglu_on_the_row_interleaved_gate_up_copy_is_bit_identical(tiny_dirs, name):
    """Decode steps of <= 16 rows stream a row-interleaved copy of the dense gate|up matrix (tile t = gate rows 8t..8t+7 then
    up rows 8t..8t+7: one-tile work items, 7 per CU at Mistral-7B's shape instead of 3.5 pairs).  Same K order per column,
    same cross-wave reduction, same rounding chain of silu(g) * u: the logits must equal the paired-tile kernel's bit for bit."""
    if name not in tiny_dirs:
        pytest.skip(f"no tiny model {name}")
    model, ref, cfg = _load_pair(tiny_dirs, name)
    eng = model.engine
    for B in (1, 5, 8, 13):
        toks = _left_pad_prompts(cfg, B, 12)
        outs = {}
        for on in (1, 0):
            eng.set_option("gate_up_interleave", on)
            kv = eng.new_kv(B, capacity=32, kv_dtype="model")
            lg = [eng.forward(toks, kv, all_positions=True)[:, -1]]
            nxt = np.argmax(lg[0], axis=-1)[:, None].astype(np.int32)
            for _ in range(4):
                lg.append(eng.forward(nxt, kv))
                nxt = np.argmax(lg[-1], axis=-1)[:, None].astype(np.int32)
            outs[on] = np.stack(lg)
            kv.close()
        eng.set_option("gate_up_interleave", 1)
        assert np.array_equal(outs[1], outs[0]), (name, B, np.abs(outs[1] - outs[0]).max())
    eng.close()


def test_f16_model_in_float32_kv_mode_runs_on_the_matrix_cores_and_equals_the_exact_kernels(tiny_dirs):
    """PagedKVCache mode of an f16 model (base.py:111-112 promotes to float32 whatever the model dtype): the linears go
    through an EXACT [hi | lo] bf16 copy of the f16 weights on the float32-activation matrix-core kernels (x split three
    ways) instead of the generic VALU kernel; both are float32 dot products of the same exact products, so the logits of
    the two routes agree to float32 summation noise -- prefill (tile GEMM: 5 x 64 rows; streaming kernel: 2 x 9) and decode."""
    model, ref, cfg = _load_pair(tiny_dirs, "llama_f16")
    eng = model.engine
    for B, L0 in ((5, 64), (2, 9)):
        toks = _left_pad_prompts(cfg, B, L0)
        outs = {}
        for on in (1, 0):
            eng.set_option("f16_hilo", on)
            kv = eng.new_kv(B, capacity=L0 + 8, kv_dtype="float32")
            lg = [eng.forward(toks, kv, all_positions=True)[:, -1]]
            nxt = np.argmax(lg[0], axis=-1)[:, None].astype(np.int32)
            for _ in range(3):
                lg.append(eng.forward(nxt, kv))
                nxt = np.argmax(lg[-1], axis=-1)[:, None].astype(np.int32)
            outs[on] = np.stack(lg)
            kv.close()
        eng.set_option("f16_hilo", 1)
        assert np.abs(outs[1] - outs[0]).max() <= 4e-3 and np.sqrt(((outs[1] - outs[0]) ** 2).mean()) <= 3e-4, \
            (B, L0, np.abs(outs[1] - outs[0]).max())
        assert not np.array_equal(outs[1], outs[0])       # (another summation order: the option really switches kernels)
    eng.close()


def _teacher_forced_greedy(model, ref, cfg, kvd, paged, B, L0, steps, margin_eps):
    toks = _left_pad_prompts(cfg, B, L0)
    kv = model.engine.new_kv(B, capacity=L0 + steps + 1, kv_dtype=kvd)
    cache = ref.make_cache(B, paged=paged)
    y = toks
    near_ties, total = 0, 0
    max_lp_err = 0.0
    for s in range(steps):
        res = model.engine.decode_sample(kv, y.astype(np.int32), SampleArgs(temp=0.0))
        logits = ref(y, cache=cache)[:, -1]
        want = ref_sample.sample(logits, temp=0.0)
        for b in range(B):
            total += 1
            wt, gt = int(want["tokens"][b, 0]), int(res["tokens"][b])
            if wt != gt:
                margin = float(logits[b, wt] - logits[b, gt])
                assert 0 <= margin <= margin_eps, f"step {s} row {b}: token {gt} != {wt}, oracle margin {margin}"
                near_ties += 1
            else:
                max_lp_err = max(max_lp_err, abs(float(res["logprobs"][b]) - float(want["logprobs"][b])))
        y = want["tokens"]                              # teacher forcing with the oracle's token
    return near_ties, total, max_lp_err


@pytest.mark.parametrize("name", ["llama_q4_f32", "llama_f32"])
@pytest.mark.parametrize("mode", ["model", "float32"])
def test_greedy_bit_exact_float32_models(tiny_dirs, name, mode):
    """BASELINE config 1 (tiny model, scripts/build_tiny_model.py shape): token ids bit-exact,
    logprobs within 1e-3 -- no near-tie exemptions needed in float32."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    kvd, paged = MODES[mode]
    near, total, lp_err = _teacher_forced_greedy(model, ref, cfg, kvd, paged, B=4, L0=12, steps=24, margin_eps=0.0)
    assert near == 0 and lp_err <= 1e-3, (near, total, lp_err)
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16", "llama_q8_f16", "llama_f16"])
def test_greedy_float32_kv_mode_16bit_models(tiny_dirs, name):
    """The reference's actual numerics (PagedKVCache float32 quirk): bit-exact ids, logprobs 1e-3."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    near, total, lp_err = _teacher_forced_greedy(model, ref, cfg, "float32", True, B=4, L0=12, steps=20,
                                                 margin_eps=2e-3)
    assert near <= 1 and lp_err <= 1e-3, (near, total, lp_err)
    model.engine.close()


@pytest.mark.parametrize("B", [33, 100, 128])
@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16"])
def test_decode_steps_of_many_rows_match_oracle(tiny_dirs, name, B):
    """BASELINE config 5 spreads 128..1024 prompts over the GPUs: decode steps of up to 128 rows per GPU (the split-K
    streaming kernel in 32-row slabs for int4, two row tiles for 16-bit weights, the decode attention over 128 cache rows;
    float32 KV: 32-row launches of the float32-activation kernel).  Teacher-forced against the oracle in the reference's
    numerics: ids equal where the margin allows, logprobs within 1e-3."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    near, total, lp_err = _teacher_forced_greedy(model, ref, cfg, "float32", True, B=B, L0=6, steps=4, margin_eps=2e-3)
    assert near <= max(1, total // 200) and lp_err <= 1e-3, (near, total, lp_err)
    near, total, lp_err = _teacher_forced_greedy(model, ref, cfg, "model", False, B=B, L0=6, steps=4, margin_eps=0.13)
    assert near <= max(2, total // 10) and lp_err <= MODEL_KV_LOGPROB_TOL, (near, total, lp_err)
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16", "llama_q8_f16", "llama_f16"])
def test_greedy_model_dtype_kv_16bit_models(tiny_dirs, name):
    """KV and activations in the 16-bit model dtype (the bandwidth-optimal default).  Logits are
    rounded to 16 bits, so exact ties are common; ids must match wherever the oracle's margin
    exceeds a few ulps of the logit."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    near, total, lp_err = _teacher_forced_greedy(model, ref, cfg, "model", False, B=4, L0=12, steps=20,
                                                 margin_eps=0.13)
    assert near <= max(2, total // 10), (near, total)
    # chosen-token logprobs on the steps where the ids agree: 16-bit logits carry ~3 ulps of rounding noise each
    # (|logit| up to ~8 -> ulp 0.03-0.06), the log-sum-exp averages it: measured <= 0.05 on these models
    print(f"{name}: model-dtype KV max |logprob - oracle| = {lp_err:.4f}")
    assert lp_err <= MODEL_KV_LOGPROB_TOL, lp_err
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_f32", "llama_bf16_gqa", "qwen3_bf16"])
def test_fused_and_unfused_decode_attention_agree(tiny_dirs, name):
    """The one-launch decode attention (norm + RoPE + append + split-KV + in-kernel combine) against
    the three-launch form, long enough context that the KV range is split across workgroups."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    B, L0 = 2, 200
    toks = _left_pad_prompts(cfg, B, L0, ragged=False)
    outs = []
    for fused in (1, 0):
        model.engine.set_option("fused_decode_attention", fused)
        kv = model.engine.new_kv(B, capacity=256, kv_dtype="model")
        model.engine.forward(toks, kv, want_logits=False)
        y = toks[:, -1:]
        steps = []
        for _ in range(4):
            lg = model.engine.forward(y, kv)
            steps.append(lg)
            y = np.argmax(lg, axis=-1)[:, None].astype(np.int32)
        outs.append(np.stack(steps))
        kv.close()
    tol = 2e-4 if name == "llama_f32" else 0.08
    assert np.abs(outs[0] - outs[1]).max() <= tol, np.abs(outs[0] - outs[1]).max()
    cache = ref.make_cache(B, paged=False)
    ref(toks, cache=cache)
    want = ref(toks[:, -1:], cache=cache)[:, -1]
    assert np.abs(outs[0][0] - want).max() <= tol
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_bf16_gqa", "llama_q4_bf16", "qwen3_bf16"])
def test_paired_gemv_launches_are_bit_identical(tiny_dirs, name):
    """fused_gemv_pairs (o_proj -> gate|up and down_proj -> next q|k|v as one launch each, in-launch seam with
    write-through hand-off) must give exactly the logits of the one-launch-per-GEMV path."""
    model, _ref, cfg = _load_pair(tiny_dirs, name)
    B = 3
    toks = _left_pad_prompts(cfg, B, 9, ragged=False)
    outs = []
    model.engine.set_option("skinny_gemm", 0)          # the pairs are built from gemv_mfma.hip's phases (int4 steps default to gemm_skinny.hip)
    for opt in (0, 3):
        model.engine.set_option("fused_gemv_pairs", opt)
        kv = model.engine.new_kv(B, capacity=64, kv_dtype="model")
        model.engine.forward(toks, kv, want_logits=False)
        y, steps = toks[:, -1:], []
        for _ in range(6):
            lg = model.engine.forward(y, kv)
            steps.append(lg)
            y = np.argmax(lg, axis=-1)[:, None].astype(np.int32)
        outs.append(np.stack(steps))
        kv.close()
    assert np.array_equal(outs[0], outs[1])
    model.engine.close()


@pytest.mark.parametrize("L0", [43, 75])       # 172 rows: the 128 x 128 kernel; 300 rows: the 256 x 256 LDS-staged kernel
@pytest.mark.parametrize("name", ["llama_bf16_gqa", "qwen3_bf16", "llama_q4_bf16"])
def test_prefill_tile_gemm_matches_oracle_and_chunked_path(tiny_dirs, name, L0):
    """Prefill with enough rows (B*L = 172, not a multiple of the 128-row tile) to take the MFMA tile
    GEMM (gemm_prefill.hip): all-position logits against the oracle and against the chunked
    skinny-kernel path, then decode on top of the KV it wrote.  int4 weights go through the [hi | lo]
    dequantised copy (2^-17 relative), which must be indistinguishable at this tolerance."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    B = 4
    toks = _left_pad_prompts(cfg, B, L0)
    outs = []
    for gemm in (1, 0):
        model.engine.set_option("prefill_gemm", gemm)
        kv = model.engine.new_kv(B, capacity=96, kv_dtype="model")
        pre = model.engine.forward(toks, kv, all_positions=True)
        nxt = model.engine.forward(toks[:, -1:], kv)
        outs.append((pre, nxt))
        kv.close()
    cache = ref.make_cache(B, paged=False)
    want_pre = ref(toks, cache=cache)
    want_nxt = ref(toks[:, -1:], cache=cache)[:, -1]
    tol = 0.1
    for pre, nxt in outs:
        assert np.abs(pre - want_pre).max() <= tol, np.abs(pre - want_pre).max()
        assert np.abs(nxt - want_nxt).max() <= tol
    assert np.abs(outs[0][0] - outs[1][0]).max() <= tol
    assert np.mean(np.abs(outs[0][0] - want_pre) > 0.02) < 0.05
    model.engine.close()


def test_generate_step_pipelined_matches_oracle_generate_step(tiny_dirs):
    """utils.generate_step (one-step-ahead pipelining, device-resident token feedback) == the
    oracle's generate_step on the reference's default cache (paged)."""
    model, ref, cfg = _load_pair(tiny_dirs, "llama_q4_f32")
    toks = _left_pad_prompts(cfg, 2, 10)
    want = []
    for (t, p), _ in zip(ref_generate.generate_step(toks, ref, paged=True), range(16)):
        want.append((t[:, 0].copy(), p[:, 0].copy()))
    cache = model.make_cache(2, paged=True)
    from mlx_parallm_amd.models.base import group_of
    group_of(cache).kv_dtype = "float32"
    got = []
    for (t, p), _ in zip(utils.generate_step(toks, model, cache=cache), range(16)):
        got.append((t[:, 0].copy(), p[:, 0].copy()))
    for (gt, gp), (wt, wp) in zip(got, want):
        assert np.array_equal(gt, wt)
        assert np.allclose(gp, wp, rtol=2e-3, atol=1e-6)
    # one step is always computed ahead (utils.py:420-427, quirk Q4), and zip() pulls the generator
    # once more before it notices that range() is exhausted -- exactly like the reference's loops
    assert cache[0].offsets == [10 + 16 + 1] * 2
    model.engine.close()


def test_top_p_sampling_with_logprobs_matches_oracle(tiny_dirs):
    """BASELINE config 3 semantics at tiny scale: top-p=0.9 sampling + logprobs, injected noise."""
    model, ref, cfg = _load_pair(tiny_dirs, "llama_q4_f32")
    B, L0, steps = 4, 8, 12
    toks = _left_pad_prompts(cfg, B, L0)
    us = RNG.random((steps + 2, B)).astype(np.float32)     # the loop runs ahead of the consumer
    want = list(zip(ref_generate.generate_step(toks, ref, temp=1.0, top_p=0.9, uniforms_fn=lambda s: us[s],
                                               paged=False, return_logits=True), range(steps)))
    got = list(zip(utils.generate_step(toks, model, temp=1.0, top_p=0.9, uniforms_fn=lambda s: us[s],
                                       cache=model.make_cache(B, paged=False), return_details=True, top_logprobs=3),
                   range(steps)))
    for (g, _), ((wt, wp, wl, wlp), _) in zip(got, want):
        assert np.array_equal(g["tokens"], wt[:, 0])
        assert np.allclose(g["logprobs"], wlp, atol=1e-3)
        lsm = ref_sample.log_softmax(wl)
        for b in range(B):
            order = np.lexsort((np.arange(wl.shape[1]), -wl[b]))[:3]
            assert np.array_equal(g["top_ids"][b], order)
            assert np.allclose(g["top_logprobs"][b], lsm[b, order], atol=1e-3)
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_f32", "llama_bf16_gqa"])
def test_score_tokens_and_temperature_logprobs_match_oracle(tiny_dirs, name):
    """mi_score_tokens (teacher-forced log-softmax gather + top-k over all positions, through the KV
    cache in two chunks) and logprobs_at_temperature of the sampler."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    tol = 2e-4 if name == "llama_f32" else 3e-2
    B, Lt, V = 2, 12, cfg["vocab_size"]
    toks = RNG.integers(0, V, size=(B, Lt + 1))
    lg = np.asarray(ref(toks[:, :-1], cache=ref.make_cache(B, paged=False)), dtype=np.float32)      # (B, Lt, V)
    tg = toks[:, 1:].copy()
    tg[1, 3] = -1                                                  # skipped position
    for temp, bias in ((0.0, None), (0.5, {7: 1.5, 11: -2.0})):
        l2 = lg.copy()
        for k, v in (bias or {}).items():
            l2[:, :, k] += np.float32(v)
        if temp > 0:
            l2 = l2 * np.float32(1.0 / temp)
        lsm = ref_sample.log_softmax(l2.reshape(B * Lt, V)).reshape(B, Lt, V)
        kv = model.engine.new_kv(B, capacity=32, kv_dtype="model")
        sp = SampleArgs(temp=temp, logit_bias=bias, top_logprobs=3, logprobs_at_temperature=True)
        a = model.engine.score_tokens(kv, toks[:, :5], tg[:, :5], sp)
        b = model.engine.score_tokens(kv, toks[:, 5:Lt], tg[:, 5:], sp)
        assert kv.offsets == [Lt, Lt]
        got = np.concatenate([a["logprobs"], b["logprobs"]], axis=1)
        want = np.take_along_axis(lsm, np.clip(tg, 0, V - 1)[..., None], axis=2)[..., 0]
        want[1, 3] = 0.0
        assert np.abs(got - want).max() <= tol * max(1.0, 1.0 / max(temp, 1e-9) if temp else 1.0)
        top_l = np.concatenate([a["top_logprobs"], b["top_logprobs"]], axis=1)
        assert np.abs(top_l - np.sort(lsm, axis=2)[:, :, ::-1][:, :, :3]).max() <= tol * (2 if temp else 1)
        if name == "llama_f32":
            top_i = np.concatenate([a["top_ids"], b["top_ids"]], axis=1)
            assert np.array_equal(top_i[0, 0], np.lexsort((np.arange(V), -lsm[0, 0]))[:3])
        kv.close()
    # sampler: logprob of the sampled token under softmax(logits / T)
    us = RNG.random((4, B)).astype(np.float32)
    gen = utils.generate_step(toks[:, :6], model, temp=0.7, top_p=0.9, uniforms_fn=lambda s: us[s],
                              cache=model.make_cache(B, paged=False), return_details=True, top_logprobs=2,
                              logprobs_at_temperature=True)
    want = ref_generate.generate_step(toks[:, :6], ref, temp=0.7, top_p=0.9, uniforms_fn=lambda s: us[s], paged=False,
                                      return_logits=True)
    for (g, (wt, _wp, wl, _wlp)), _ in zip(zip(gen, want), range(2)):
        lsm_t = ref_sample.log_softmax(np.asarray(wl, dtype=np.float32) * np.float32(1.0 / 0.7))
        if name == "llama_f32":
            assert np.array_equal(g["tokens"], wt[:, 0])
            assert np.allclose(g["logprobs"], lsm_t[np.arange(B), wt[:, 0]], atol=1e-3)
        assert np.allclose(g["top_logprobs"], np.sort(lsm_t, axis=1)[:, ::-1][:, :2], atol=tol * 2)
    model.engine.close()


def test_kv_growth_keeps_contents_and_reset(tiny_dirs):
    model, ref, cfg = _load_pair(tiny_dirs, "llama_f32", max_pos=1024)
    toks = _left_pad_prompts(cfg, 2, 6, ragged=False)
    kv = model.engine.new_kv(2, capacity=8, kv_dtype="model", step=8)
    cache = ref.make_cache(2, paged=False)
    y = toks
    for s in range(20):                                 # crosses the 8-token capacity twice
        got = model.engine.forward(y.astype(np.int32), kv)
        want = ref(y, cache=cache)[:, -1]
        assert np.abs(got - want).max() <= 2e-4
        y = np.argmax(want, axis=-1)[:, None]
    assert kv.capacity >= 25 and kv.offsets == [25, 25]
    kv.reset()
    assert kv.offsets == [0, 0]
    got = model.engine.forward(toks, kv)
    want = ref(toks, cache=ref.make_cache(2, paged=False))[:, -1]
    assert np.abs(got - want).max() <= 2e-4
    model.engine.close()


def test_lora_adapter_applied(tiny_dirs, tmp_path):
    """BASELINE config 5 ingredient: adapters.safetensors + adapter_config.json applied to q/v of the
    last num_layers blocks (rl_training/lora_init.py:68-72,140-153)."""
    import json

    import torch
    from safetensors.torch import save_file

    d, cfg = tiny_dirs["llama_q4_bf16"]
    H, nh, nkv, D = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    rank, nl = 16, 1
    rng = np.random.default_rng(5)
    w = {}
    for i in range(cfg["num_hidden_layers"] - nl, cfg["num_hidden_layers"]):
        for key, n in (("self_attn.q_proj", nh * D), ("self_attn.v_proj", nkv * D)):
            w[f"model.layers.{i}.{key}.lora_a"] = torch.from_numpy(
                (rng.uniform(-1, 1, (H, rank)) / np.sqrt(H)).astype(np.float32))
            w[f"model.layers.{i}.{key}.lora_b"] = torch.from_numpy(rng.standard_normal((rank, n)).astype(np.float32) * 0.05)
    ad = tmp_path / "adapter"
    ad.mkdir()
    save_file(w, str(ad / "adapters.safetensors"))
    (ad / "adapter_config.json").write_text(json.dumps({
        "fine_tune_type": "lora", "num_layers": nl,
        "lora_parameters": {"rank": rank, "scale": 10.0, "dropout": 0.05, "keys": ["self_attn.q_proj", "self_attn.v_proj"]}}))
    model = utils.load_model(d, max_positions=256)
    utils.load_adapters(model, str(ad))
    ref = ref_generate.load(d, adapter_path=str(ad), max_pos=256)
    base = ref_generate.load(d, max_pos=256)
    toks = _left_pad_prompts(cfg, 2, 7)
    for kvd, paged, tol in (("float32", True, 4e-3), ("model", False, 0.08)):
        kv = model.engine.new_kv(2, capacity=16, kv_dtype=kvd)
        got = model.engine.forward(toks, kv)
        want = ref(toks, cache=ref.make_cache(2, paged=paged))[:, -1]
        plain = base(toks, cache=base.make_cache(2, paged=paged))[:, -1]
        assert np.abs(want - plain).max() > 10 * tol          # the adapter really changes the logits
        assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    model.engine.close()


@pytest.mark.parametrize("name", ["llama_f32", "qwen3_bf16", "llama_q8_f16", "llama_q4_bf16"])
def test_row_subset_steps_equal_independent_sequences(tiny_dirs, name):
    """mi_step_enqueue_rows / mi_kv_reset_row (continuous batching): sequences admitted into different slots
    at different times, decoded together in changing row sets, one slot recycled -- every sequence must produce
    what it produces alone (rows are independent), i.e. the oracle's single-sequence greedy run."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    eng = model.engine
    exact = name == "llama_f32"
    V = cfg["vocab_size"]
    prompts = {"A": RNG.integers(3, V, size=(1, 7)), "B": RNG.integers(3, V, size=(1, 5)), "C": RNG.integers(3, V, size=(1, 40))}

    def alone(p, n):
        cache = ref.make_cache(1, paged=False)
        y, out, margins = p, [], []
        for _ in range(n):
            lg = ref(y, cache=cache)[:, -1]
            top2 = np.sort(lg[0])[-2:]
            margins.append(float(top2[1] - top2[0]))
            y = np.argmax(lg, axis=-1)[:, None]
            out.append(int(y[0, 0]))
        return out, margins

    kv = eng.new_kv(4, capacity=64, kv_dtype="model")
    greedy = SampleArgs(temp=0.0)
    got = {k: [] for k in prompts}

    def step(rows, names, tokens):
        res = eng.step_wait(eng.step_enqueue_rows(kv, rows, tokens, greedy), len(rows))
        for nm, t in zip(names, res["tokens"]):
            got[nm].append(int(t))
        return res["tokens"]

    step([2], ["A"], prompts["A"])                                  # admit A into slot 2
    assert kv.offsets == [0, 0, 7, 0]
    step([2], ["A"], [[got["A"][-1]]])                              # A decodes alone
    step([0], ["B"], prompts["B"])                                  # admit B into slot 0
    step([2, 0], ["A", "B"], [[got["A"][-1]], [got["B"][-1]]])      # both, explicit tokens (row set changed)
    step([2, 0], ["A", "B"], None)                                  # same row set: tokens stay on the device
    step([2, 0], ["A", "B"], None)
    assert kv.offsets == [8, 0, 11, 0]
    kv.reset_row(2)                                                 # A is done; its slot is recycled for C
    assert kv.offsets == [8, 0, 0, 0]
    step([2], ["C"], prompts["C"])
    step([0, 2], ["B", "C"], [[got["B"][-1]], [got["C"][-1]]])      # other order of rows
    step([0, 2], ["B", "C"], None)
    for nm in prompts:
        want, margins = alone(prompts[nm], len(got[nm]))
        for i, (g, w) in enumerate(zip(got[nm], want)):
            if g != w:
                assert not exact and margins[i] <= 0.13, (nm, i, got[nm], want, margins[i])
                break                                              # a near-tie flip changes the continuation
    with pytest.raises(ValueError):
        eng.step_enqueue_rows(kv, [1, 1], [[3], [4]], greedy)
    with pytest.raises(ValueError):
        eng.step_enqueue_rows(kv, [4], [[3]], greedy)
    with pytest.raises(ValueError):
        eng.step_enqueue_rows(kv, [0, 1, 2], None, greedy)          # device feed with a different row count
    kv.close()
    eng.close()


@pytest.mark.parametrize("kvd", ["model", "float32"])
@pytest.mark.parametrize("name", ["llama_f32", "qwen3_bf16", "llama_q4_bf16"])
def test_mixed_steps_chunked_prefill_next_to_decode_rows(tiny_dirs, name, kvd):
    """mi_step_enqueue_mixed: a 70-token prompt enters the cache in chunks (32 + 32 + 6 tokens) while two live
    sequences keep decoding IN THE SAME STEPS (one pass over the weights each); a second short prompt rides along as
    one chunk.  Every sequence must produce what it produces alone (the oracle's single-sequence greedy run), the
    chunked prompt's KV must equal a one-shot prefill's (next-token logits), inner chunks return nothing."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    eng = model.engine
    exact = name == "llama_f32"
    V = cfg["vocab_size"]
    paged = kvd == "float32"
    P = {"A": RNG.integers(3, V, size=9), "B": RNG.integers(3, V, size=6), "C": RNG.integers(3, V, size=70), "D": RNG.integers(3, V, size=5)}

    def alone(p, n):
        cache = ref.make_cache(1, paged=paged)
        y, out, margins = p[None], [], []
        for _ in range(n):
            lg = ref(y, cache=cache)[:, -1]
            top2 = np.sort(lg[0])[-2:]
            margins.append(float(top2[1] - top2[0]))
            y = np.argmax(lg, axis=-1)[:, None]
            out.append(int(y[0, 0]))
        return out, margins

    kv = eng.new_kv(4, capacity=128, kv_dtype=kvd)
    greedy = SampleArgs(temp=0.0)
    got = {k: [] for k in P}

    def mixed(rows, names, toks, want):
        res = eng.step_wait(eng.step_enqueue_mixed(kv, rows, toks, want, greedy), sum(want))
        for nm, t in zip([n for n, w in zip(names, want) if w], res["tokens"]):
            got[nm].append(int(t))

    mixed([1, 3], ["A", "B"], [P["A"], P["B"]], [1, 1])                                     # two prompts admitted in one step
    assert kv.offsets == [0, 9, 0, 6]
    mixed([1, 3, 0], ["A", "B", "C"], [[got["A"][-1]], [got["B"][-1]], P["C"][:32]], [1, 1, 0])     # C: first chunk, no logits
    assert kv.offsets == [32, 10, 0, 7] and len(got["C"]) == 0
    mixed([1, 3, 0], ["A", "B", "C"], [[got["A"][-1]], [got["B"][-1]], P["C"][32:64]], [1, 1, 0])
    mixed([1, 3, 2, 0], ["A", "B", "D", "C"], [[got["A"][-1]], [got["B"][-1]], P["D"], P["C"][64:]], [1, 1, 1, 1])  # last chunk + D whole
    assert kv.offsets == [70, 12, 5, 9] and len(got["C"]) == 1 and len(got["D"]) == 1
    for _ in range(3):                                                                       # all four decode together
        mixed([0, 1, 2, 3], ["C", "A", "D", "B"], [[got["C"][-1]], [got["A"][-1]], [got["D"][-1]], [got["B"][-1]]], [1, 1, 1, 1])
    res = eng.step_wait(eng.step_enqueue_rows(kv, [0, 1, 2, 3], [[got["C"][-1]], [got["A"][-1]], [got["D"][-1]], [got["B"][-1]]], greedy), 4)
    for nm, t in zip(["C", "A", "D", "B"], res["tokens"]):                                    # ... and the plain row step continues them
        got[nm].append(int(t))
    for nm in P:
        want, margins = alone(P[nm], len(got[nm]))
        for i, (g, w) in enumerate(zip(got[nm], want)):
            if g != w:
                assert not exact and margins[i] <= (2e-3 if paged else 0.13), (nm, i, got[nm], want, margins[i])
                break                                              # a near-tie flip changes the continuation
    # the chunked prompt's cache equals a one-shot prefill's: logits of the same next token, rounding noise apart
    kv2 = eng.new_kv(1, capacity=128, kv_dtype=kvd)
    eng.forward(P["C"][None].astype(np.int32), kv2, want_logits=False)
    nxt = np.asarray([[got["C"][0]]], dtype=np.int32)
    kv3 = eng.new_kv(2, capacity=128, kv_dtype=kvd)
    eng.step_wait(eng.step_enqueue_mixed(kv3, [1], [P["C"][:40]], [0], greedy), 0)
    eng.step_wait(eng.step_enqueue_mixed(kv3, [1], [P["C"][40:]], [1], greedy), 1)
    a = eng.forward(nxt, kv2)
    b = eng.forward(np.concatenate([nxt, nxt]), kv3)[1:2]
    assert np.abs(a - b).max() <= (2e-4 if exact else (4e-3 if paged else 0.1)), np.abs(a - b).max()
    with pytest.raises(ValueError):
        eng.step_enqueue_mixed(kv, [0, 1], [P["C"][:8], [5]], [1, 1], greedy)              # one-token segments must come first
    with pytest.raises(ValueError):
        eng.step_enqueue_mixed(kv, [0, 0], [[5], [6]], [1, 1], greedy)
    kv.close(); kv2.close(); kv3.close()
    eng.close()


def test_converted_checkpoint_and_lora_hot_swap(tiny_dirs, tmp_path):
    """a directory written by convert(quantize=True) loads and matches the oracle on the same
    files; weight_updater swaps adapters on the live engine (adapters.safetensors + config, then adapter.npz)."""
    import json

    from safetensors.torch import save_file
    import torch

    from mlx_parallm_amd import convert as cv
    from mlx_parallm_amd.weight_updater import apply_lora_update

    src, cfg = tiny_dirs["llama_bf16_gqa"]
    q4 = tmp_path / "q4"
    cv.convert(src, str(q4), quantize=True, q_group_size=64, q_bits=4, dtype="bfloat16")
    model = utils.load_model(str(q4), max_positions=256)
    ref = ref_generate.load(str(q4), max_pos=256)
    toks = _left_pad_prompts(cfg, 2, 7)
    kv = model.engine.new_kv(2, capacity=16, kv_dtype="model")
    got = model.engine.forward(toks, kv)
    want = ref(toks, cache=ref.make_cache(2, paged=False))[:, -1]
    assert np.abs(got - want).max() <= 0.08, np.abs(got - want).max()
    kv.close()

    H, nh, D = cfg["hidden_size"], cfg["num_attention_heads"], cfg["head_dim"]
    rank, last = 8, cfg["num_hidden_layers"] - 1
    rng = np.random.default_rng(9)

    def factors(seed_scale):
        a = (rng.uniform(-1, 1, (H, rank)) / np.sqrt(H)).astype(np.float32)
        b = (rng.standard_normal((rank, nh * D)) * seed_scale).astype(np.float32)
        return a, b

    def logits():
        kv2 = model.engine.new_kv(2, capacity=16, kv_dtype="model")
        out = model.engine.forward(toks, kv2)
        kv2.close()
        return out

    base = logits()
    a1, b1 = factors(0.5)
    ad1 = tmp_path / "ad1"
    ad1.mkdir()
    save_file({f"model.layers.{last}.self_attn.q_proj.lora_a": torch.from_numpy(a1),
               f"model.layers.{last}.self_attn.q_proj.lora_b": torch.from_numpy(b1)}, str(ad1 / "adapters.safetensors"))
    (ad1 / "adapter_config.json").write_text(json.dumps({
        "fine_tune_type": "lora", "num_layers": 1,
        "lora_parameters": {"rank": rank, "scale": 4.0, "dropout": 0.0, "keys": ["self_attn.q_proj"]}}))
    apply_lora_update(model, str(ad1))
    ref1 = ref_generate.load(str(q4), adapter_path=str(ad1), max_pos=256)
    w1 = ref1(toks, cache=ref1.make_cache(2, paged=False))[:, -1]
    l1 = logits()
    assert np.abs(l1 - base).max() > 0.16 and np.abs(l1 - w1).max() <= 0.08     # the adapter's effect is >= 2x the tolerance

    a2, b2 = factors(1.0)                                        # second adapter: npz, no config -> scale of the first is kept
    ad2 = tmp_path / "ad2"
    ad2.mkdir()
    np.savez(ad2 / "adapter.npz", **{f"model.layers.{last}.self_attn.q_proj.lora_a": a2,
                                     f"model.layers.{last}.self_attn.q_proj.lora_b": b2})
    import threading
    assert apply_lora_update(model, str(ad2), lock=threading.RLock()) == 1
    save_file({f"model.layers.{last}.self_attn.q_proj.lora_a": torch.from_numpy(a2),
               f"model.layers.{last}.self_attn.q_proj.lora_b": torch.from_numpy(b2)}, str(ad1 / "adapters.safetensors"))
    ref2 = ref_generate.load(str(q4), adapter_path=str(ad1), max_pos=256)     # same factors, scale 4.0, through the oracle
    w2 = ref2(toks, cache=ref2.make_cache(2, paged=False))[:, -1]
    l2 = logits()
    assert np.abs(l2 - l1).max() > 0.16 and np.abs(l2 - w2).max() <= 0.08
    model.engine.close()


def test_error_behaviour(tiny_dirs, tmp_path):
    with pytest.raises(utils.ModelNotFoundError):
        utils.load(str(tmp_path / "nope"))
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        utils.load_model(tmp_path / "empty")
    (tmp_path / "empty" / "config.json").write_text('{"model_type": "gemma"}')
    with pytest.raises(FileNotFoundError):
        utils.load_model(tmp_path / "empty")            # no safetensors (utils.py:663-665)
    model, ref, cfg = _load_pair(tiny_dirs, "llama_f32")
    with pytest.raises(NotImplementedError):
        next(utils.generate_step(np.zeros((1, 3), np.int32), model, repetition_penalty=1.1))
    kv = model.engine.new_kv(2, capacity=8, kv_dtype="model")
    with pytest.raises(ValueError):
        model.engine.forward(np.zeros((3, 2), np.int32), kv)           # batch mismatch (base.py:125)
    with pytest.raises(ValueError):
        model.engine.forward(np.full((2, 2), cfg["vocab_size"], np.int32), kv)   # token id out of range
    model.engine.close()


@pytest.mark.parametrize("kvd", ["model", "float32"])
@pytest.mark.parametrize("name", ["llama_f32", "llama_bf16_gqa", "qwen3_bf16", "llama_q4_bf16"])
def test_paged_kv_is_bit_identical_to_contiguous_kv(tiny_dirs, name, kvd):
    """mi_kv_create_paged: the block table only changes WHERE a token's K / V row lives, so every kernel family
    (VALU and MFMA decode attention, prefill attention, rope_append, mixed steps) must give the same bits as the
    contiguous cache: logits of a ragged prefill, of decode steps on row subsets, and of a chunked prefill; blocks are
    handed out in a scrambled order (rows interleave their allocations), recycled on reset_row, and the arena's
    exhaustion is an error, not a fault."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    eng = model.engine
    V = cfg["vocab_size"]
    B, L0 = 3, 37
    toks = RNG.integers(3, V, size=(B, L0)).astype(np.int32)
    flat = eng.new_kv(B, capacity=128, kv_dtype=kvd)
    paged = eng.new_paged_kv(B, block_tokens=16, n_blocks=40, max_tokens_per_row=128, kv_dtype=kvd)
    assert paged.stats()["free_blocks"] == 39 and paged.capacity == 128
    a = eng.forward(toks, flat, all_positions=True)
    b = eng.forward(toks, paged, all_positions=True)
    assert np.array_equal(a, b)
    greedy = SampleArgs(temp=0.0, top_logprobs=3)
    nxt = np.argmax(a[:, -1], axis=-1).astype(np.int32)
    for step in range(20):                                   # crosses block boundaries (37 -> 57 tokens, 16-token blocks)
        rows = [0, 1, 2] if step % 3 else [2, 0]
        ra = eng.step_wait(eng.step_enqueue_rows(flat, rows, nxt[rows][:, None], greedy), len(rows), 3)
        rb = eng.step_wait(eng.step_enqueue_rows(paged, rows, nxt[rows][:, None], greedy), len(rows), 3)
        assert np.array_equal(ra["tokens"], rb["tokens"]) and np.array_equal(ra["logprobs"], rb["logprobs"])
        assert np.array_equal(ra["top_logprobs"], rb["top_logprobs"])
        nxt[rows] = ra["tokens"]
    assert flat.offsets == paged.offsets
    # recycle row 1, admit a new prompt there in chunks next to the other rows' decode steps (mixed steps)
    flat.reset_row(1); paged.reset_row(1)
    p2 = RNG.integers(3, V, size=45).astype(np.int32)
    for kv in (flat, paged):
        res = []
        res.append(eng.step_wait(eng.step_enqueue_mixed(kv, [0, 2, 1], [[int(nxt[0])], [int(nxt[2])], p2[:20]], [1, 1, 0], greedy), 2, 3))
        t0, t2 = res[-1]["tokens"]
        res.append(eng.step_wait(eng.step_enqueue_mixed(kv, [0, 2, 1], [[int(t0)], [int(t2)], p2[20:]], [1, 1, 1], greedy), 3, 3))
        kv.results = res
    for ra, rb in zip(flat.results, paged.results):
        assert np.array_equal(ra["tokens"], rb["tokens"]) and np.array_equal(ra["logprobs"], rb["logprobs"])
    st = paged.stats()
    assert st["free_blocks"] == 39 - sum((o + 15) // 16 for o in paged.offsets)
    # exhaustion: an arena with too few blocks for the step fails loudly
    small = eng.new_paged_kv(2, block_tokens=16, n_blocks=4, max_tokens_per_row=128, kv_dtype=kvd)
    with pytest.raises(RuntimeError, match="exhausted"):
        eng.step_wait(eng.step_enqueue_rows(small, [0, 1], toks[:2, :33], greedy), 2, 3)     # 2 x 3 blocks > 3 usable
    flat.close(); paged.close(); small.close()
    eng.close()


@pytest.mark.parametrize("name", ["llama_f32", "qwen3_bf16"])
def test_prefix_reuse_on_the_paged_cache(tiny_dirs, name):
    """mi_kv_prefix_attach / publish: a second prompt that starts with the same 40 tokens maps the first prompt's two
    full 16-token blocks instead of recomputing them, prefills only the rest, and still produces what it produces
    alone (oracle, single sequence); a prompt that differs inside the first block shares nothing; the shared blocks
    survive the first row's reset and are evicted, least recently used first, when the arena runs out."""
    model, ref, cfg = _load_pair(tiny_dirs, name)
    eng = model.engine
    exact = name == "llama_f32"
    V = cfg["vocab_size"]
    common = RNG.integers(3, V, size=40).astype(np.int32)
    pa = np.concatenate([common, RNG.integers(3, V, size=7).astype(np.int32)])
    pb = np.concatenate([common, RNG.integers(3, V, size=11).astype(np.int32)])
    pc = pa.copy(); pc[5] = (pc[5] + 1) % V
    kv = eng.new_paged_kv(3, block_tokens=16, n_blocks=12, max_tokens_per_row=96, kv_dtype="model")
    greedy = SampleArgs(temp=0.0)

    def alone(p, n):
        cache = ref.make_cache(1, paged=False)
        y, out, margins = p[None], [], []
        for _ in range(n):
            lg = ref(y, cache=cache)[:, -1]
            top2 = np.sort(lg[0])[-2:]
            margins.append(float(top2[1] - top2[0]))
            y = np.argmax(lg, axis=-1)[:, None]
            out.append(int(y[0, 0]))
        return out, margins

    def run(row, prompt, n):
        reused = kv.prefix_attach(row, prompt)
        assert kv.offsets[row] == reused
        out = [int(eng.step_wait(eng.step_enqueue_rows(kv, [row], prompt[None, reused:], greedy), 1)["tokens"][0])]
        kv.prefix_publish(row, prompt)
        for _ in range(n - 1):
            out.append(int(eng.step_wait(eng.step_enqueue_rows(kv, [row], [[out[-1]]], greedy), 1)["tokens"][0]))
        return reused, out

    def check(got, prompt):
        want, margins = alone(prompt, len(got))
        for i, (g, w) in enumerate(zip(got, want)):
            if g != w:
                assert not exact and margins[i] <= 0.13, (i, got, want, margins[i])
                break

    ra, ga = run(0, pa, 6)
    assert ra == 0 and kv.stats()["cached_blocks"] == 2                       # 47 tokens: two full blocks published
    rb, gb = run(1, pb, 6)
    assert rb == 32 and kv.stats()["reused_tokens"] == 32                      # both full blocks of the common prefix
    rc, gc = run(2, pc, 4)
    assert rc == 0                                                             # differs inside block 0: nothing shared
    check(ga, pa); check(gb, pb); check(gc, pc)
    kv.reset_row(0)                                                            # the owner leaves; its published blocks stay
    kv.reset_row(2)
    rd, gd = run(0, pa, 3)
    assert rd == 32
    check(gd, pa)
    with pytest.raises(ValueError):
        kv.prefix_attach(1, pb)                                                # row 1 is not empty
    # the cache gives its blocks back under pressure: fill the arena with a long unrelated sequence
    kv.reset_row(0); kv.reset_row(1)
    long = RNG.integers(3, V, size=90).astype(np.int32)
    eng.step_wait(eng.step_enqueue_rows(kv, [2], long[None], greedy), 1)       # 6 blocks; 11 usable, some held by the cache
    eng.step_wait(eng.step_enqueue_rows(kv, [1], long[None, :80], greedy), 1)  # 5 more: cached blocks must be evicted
    assert kv.stats()["evictions"] >= 1
    eng.invalidate_prefix_caches()
    assert kv.stats()["cached_blocks"] == 0
    kv.close()
    eng.close()


def test_scheduler_waits_for_kv_blocks_instead_of_failing_live_rows(tiny_dirs):
    """The continuous scheduler on a deliberately small block arena (7 usable blocks of 16 tokens, 2 slots): the second
    request finds a free slot but not the blocks its prompt + max_tokens may need, so it waits until the first one
    finishes -- both then equal their solo oracle runs; a request that can never fit is refused with an error, nothing
    else is disturbed; prompts enter through chunked mixed steps and the second prompt reuses the first one's prefix block."""
    import threading

    from mlx_parallm_amd.server.scheduler import ContinuousScheduler

    d, cfg = tiny_dirs["llama_f32"]
    model, tok = utils.load(d)
    ref = ref_generate.load(d, max_pos=512)
    sched = ContinuousScheduler(model, tok, max_slots=2, kv_dtype="model", chunk_tokens=16, block_tokens=16, kv_blocks=8)
    sched.start()
    done, ev = {}, threading.Event()

    def sink(name):
        def f(seq, delta, reason):
            if reason is not None:
                done[name] = (list(seq.generated), reason)
                if len(done) == 3:
                    ev.set()
        return f

    common = list(range(40, 60))                            # 20 shared prompt tokens: one full 16-token block
    pa, pb = common + [7, 8, 9], common + [11, 12]
    sched.submit(pa, 50, 0.0, 1.0, sink("a"))               # 23 + 50 tokens -> 5 blocks
    sched.submit(pb, 50, 0.0, 1.0, sink("b"))               # 22 + 50 -> 5 blocks: must wait for a's blocks
    sched.submit(list(range(3, 100)), 400, 0.0, 1.0, sink("huge"))   # 497 tokens -> 32 blocks: never fits
    assert ev.wait(timeout=120)
    stats = sched.kv.stats()
    sched.stop()
    assert done["huge"][1] == "error" and done["huge"][0] == []
    eos = tok.eos_token_id
    for name, p in (("a", pa), ("b", pb)):
        want = []
        for _, (t, _p) in zip(range(50), ref_generate.generate_step(np.asarray(p)[None], ref, paged=False)):
            if int(t[0, 0]) == eos:
                break
            want.append(int(t[0, 0]))
        assert done[name][0] == want, name
    assert sched.max_rows_seen == 1                          # never both at once: the arena, not the slots, was the limit
    assert sched.prefix_hit_tokens == 16 and stats["evictions"] >= 0
    model.engine.close()


def test_prefix_eviction_takes_leaves_before_parents_and_stats_count_what_is_evictable(tiny_dirs):
    """A published chain is only reachable from its root (attach stops at the first miss), so under pressure the deepest
    block must go first: after ONE eviction a prompt with the same prefix still maps the two blocks in front of it instead
    of nothing (round-2 advisory).  `evictable_blocks` counts published blocks that no live row maps -- what the
    scheduler's admission control may rely on -- while `cached_blocks` counts every published block."""
    model, ref, cfg = _load_pair(tiny_dirs, "llama_f32")
    eng = model.engine
    V = cfg["vocab_size"]
    greedy = SampleArgs(temp=0.0)
    p = RNG.integers(3, V, size=52).astype(np.int32)                           # three full 16-token blocks + 4 tokens
    kv = eng.new_paged_kv(2, block_tokens=16, n_blocks=8, max_tokens_per_row=112, kv_dtype="model")   # 7 usable blocks
    eng.step_wait(eng.step_enqueue_rows(kv, [0], p[None], greedy), 1)          # row 0: 4 blocks
    kv.prefix_publish(0, p)
    st = kv.stats()
    assert st["cached_blocks"] == 3 and st["evictable_blocks"] == 0 and st["free_blocks"] == 3   # the live row pins them
    kv.reset_row(0)
    st = kv.stats()
    assert st["cached_blocks"] == 3 and st["evictable_blocks"] == 3 and st["free_blocks"] == 4
    other = RNG.integers(3, V, size=80).astype(np.int32)                       # 5 blocks: 4 free + ONE eviction
    eng.step_wait(eng.step_enqueue_rows(kv, [1], other[None], greedy), 1)
    st = kv.stats()
    assert st["evictions"] == 1 and st["cached_blocks"] == 2
    assert kv.prefix_attach(0, p) == 32                                        # the leaf went; root and middle block are still a chain
    kv.close()
    eng.close()


def test_scheduler_admission_counts_only_evictable_prefix_blocks(tiny_dirs):
    """Two UNRELATED prompts on a 7-block arena (16-token blocks, 2 slots).  A (23 + 50 tokens -> 5 blocks) publishes its
    first prompt block when its prefill is done; that block is cached but pinned by A itself.  B (22 + 20 -> 3 blocks)
    arrives while A holds 2 blocks and may still claim 3: 5 free blocks do not cover 3 + 3, so B must WAIT although
    free + cached = 6 would (round-2 advisory: the over-commit ended in 'block arena is exhausted' and failed every live
    row).  Both must equal their solo oracle runs and must never have been live together."""
    import threading

    from mlx_parallm_amd.server.scheduler import ContinuousScheduler

    d, cfg = tiny_dirs["llama_f32"]
    model, tok = utils.load(d)
    ref = ref_generate.load(d, max_pos=512)
    sched = ContinuousScheduler(model, tok, max_slots=2, kv_dtype="model", chunk_tokens=16, block_tokens=16, kv_blocks=8)
    sched.start()
    done, ev = {}, threading.Event()

    def sink(name):
        def f(seq, delta, reason):
            if reason is not None:
                done[name] = (list(seq.generated), reason)
                if len(done) == 2:
                    ev.set()
        return f

    pa, pb = list(range(40, 63)), list(range(200, 222))
    eos = tok.eos_token_id

    def solo(p, n):
        want = []
        for _, (t, _p) in zip(range(n), ref_generate.generate_step(np.asarray(p)[None], ref, paged=False)):
            if int(t[0, 0]) == eos:
                break
            want.append(int(t[0, 0]))
        return want

    wa, wb = solo(pa, 50), solo(pb, 20)
    assert len(wa) >= 40                                    # A really grows into its fifth block (else the arena never fills)
    sched.submit(pa, 50, 0.0, 1.0, sink("a"))
    sched.submit(pb, 20, 0.0, 1.0, sink("b"))
    assert ev.wait(timeout=120)
    sched.stop()
    assert done["a"] == (wa, done["a"][1]) and done["a"][1] != "error"
    assert done["b"] == (wb, done["b"][1]) and done["b"][1] != "error"
    assert sched.max_rows_seen == 1
    model.engine.close()


def test_scheduler_soak_random_arrivals_prefixes_and_cancels(tiny_dirs):
    """120 requests against 4 slots and a 40-block arena (16-token blocks): random prompt lengths around a few shared
    prefixes (prefix-KV reuse), random max_tokens, every seventh request cancelled while it runs, arrivals in bursts while
    others decode (chunked prefill in mixed steps, admission by free blocks).  Every request must end exactly once with a
    sane reason; every uncancelled greedy request must equal the oracle's solo run of its prompt; the arena must be whole
    again at the end (no leaked blocks, nothing left mapped)."""
    import threading
    import time

    from mlx_parallm_amd.server.scheduler import ContinuousScheduler

    d, cfg = tiny_dirs["llama_f32"]
    model, tok = utils.load(d)
    ref = ref_generate.load(d, max_pos=512)
    sched = ContinuousScheduler(model, tok, max_slots=4, kv_dtype="model", chunk_tokens=24, block_tokens=16, kv_blocks=41)
    sched.start()
    rng = np.random.default_rng(2024)
    V = cfg["vocab_size"]
    prefixes = [rng.integers(3, V, size=n).tolist() for n in (16, 33, 48)]
    N = 120
    done, lock, ev = {}, threading.Lock(), threading.Event()
    dup = []

    def sink(i):
        def f(seq, delta, reason):
            if reason is not None:
                with lock:
                    if i in done:
                        dup.append(i)
                    done[i] = (list(seq.generated), reason)
                    if len(done) == N:
                        ev.set()
        return f

    reqs, seqs = [], []
    for i in range(N):
        p = list(prefixes[i % 3]) if i % 4 else []
        p += rng.integers(3, V, size=int(rng.integers(1, 40))).tolist()
        mt = int(rng.integers(1, 24))
        reqs.append((p, mt))
        seqs.append(sched.submit(p, mt, 0.0, 1.0, sink(i)))
        if i % 7 == 3:
            time.sleep(0.002)
            seqs[-1].cancel()
        if i % 10 == 9:
            time.sleep(0.01)                      # a burst, then a pause while the others decode
    assert ev.wait(timeout=300), f"only {len(done)} of {N} requests finished"
    time.sleep(0.05)
    stats = sched.kv.stats()
    sched.stop()
    assert not dup, f"finished twice: {dup}"
    eos = tok.eos_token_id
    checked = 0
    for i, (p, mt) in enumerate(reqs):
        got, reason = done[i]
        assert reason in ("stop", "length", "cancelled"), (i, reason)
        if reason == "cancelled":
            assert i % 7 == 3
            continue
        want = []
        for _, (t, _p) in zip(range(mt), ref_generate.generate_step(np.asarray(p)[None], ref, paged=False)):
            if int(t[0, 0]) == eos:
                break
            want.append(int(t[0, 0]))
        if i % 3 == 0 or len(p) > 48:             # (the oracle's solo runs are the slow part: a third of them + the long ones)
            assert got == want, (i, len(p), mt, got, want)
            checked += 1
    assert checked >= 40
    # every row released: the only blocks not free are the ones the prefix cache keeps (and those are evictable)
    assert stats["free_blocks"] + stats["cached_blocks"] == stats["usable_blocks"], stats
    model.engine.close()
