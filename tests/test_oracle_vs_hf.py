"""Pins the oracle's decoder maths against an INDEPENDENT implementation: HuggingFace
``transformers`` Llama / Mistral / Qwen3 run in float32 on the same random weights.

This does not pin MLX-specific behaviour (rounding points, the float32 PagedKVCache quirk,
affine quantisation); it pins what the reference's model files compute: pre-norm blocks,
half-split RoPE with per-row offsets, GQA head mapping, qwen3 q/k norms, SwiGLU, tied /
untied heads, KV-cached decode == full recompute.
"""
import numpy as np
import pytest
import torch

from oracle import ref_generate


def _hf_model(cfg, model_dir):
    from safetensors.torch import load_file
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen3Config, Qwen3ForCausalLM

    common = dict(
        hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
        intermediate_size=cfg["intermediate_size"], num_attention_heads=cfg["num_attention_heads"],
        num_key_value_heads=cfg["num_key_value_heads"], rms_norm_eps=cfg["rms_norm_eps"],
        vocab_size=cfg["vocab_size"], rope_theta=cfg["rope_theta"],
        tie_word_embeddings=cfg["tie_word_embeddings"], max_position_embeddings=4096,
        attention_bias=False, head_dim=cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"],
    )
    if cfg["model_type"] == "qwen3":
        m = Qwen3ForCausalLM(Qwen3Config(**common))
    else:
        m = LlamaForCausalLM(LlamaConfig(**common, mlp_bias=False))
    sd = {k: v.to(torch.float32) for k, v in load_file(f"{model_dir}/model.safetensors").items()}
    if cfg["tie_word_embeddings"]:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "rotary" not in k], missing
    assert not unexpected, unexpected
    return m.eval()


@pytest.fixture(scope="module")
def f32_dirs(tmp_path_factory):
    from mlx_parallm_amd.tiny_model import build_tiny_model

    root = tmp_path_factory.mktemp("hf")
    out = {}
    out["llama"] = (str(root / "llama"), build_tiny_model(
        root / "llama", seed=11, vocab_size=300, dtype="float32", quantize_model=False, hidden_size=64, layers=3,
        heads=8, kv_heads=2, intermediate_size=96, head_dim=8, tie_word_embeddings=True, norm_jitter=0.2,
        with_tokenizer=False))
    out["qwen3"] = (str(root / "qwen3"), build_tiny_model(
        root / "qwen3", seed=12, model_type="qwen3", vocab_size=300, dtype="float32", quantize_model=False,
        hidden_size=64, layers=2, heads=5, kv_heads=1, intermediate_size=96, head_dim=16,
        tie_word_embeddings=False, norm_jitter=0.2, with_tokenizer=False))
    return out


@pytest.mark.parametrize("name", ["llama", "qwen3"])
def test_oracle_matches_hf_prefill_and_decode(f32_dirs, name):
    d, cfg = f32_dirs[name]
    ref = ref_generate.load(d)
    hf = _hf_model(cfg, d)
    rng = np.random.default_rng(0)
    B, L0, steps = 3, 7, 4
    toks = rng.integers(0, cfg["vocab_size"], size=(B, L0 + steps))
    with torch.no_grad():
        want = hf(torch.from_numpy(toks)).logits.numpy()
    # full-sequence forward without a cache
    got = ref(toks, cache=None)
    assert np.allclose(got, want, atol=2e-4, rtol=1e-4), np.abs(got - want).max()
    # prefill + token-by-token decode with BOTH cache classes must equal the full forward
    for paged in (True, False):
        cache = ref.make_cache(B, paged=paged)
        pre = ref(toks[:, :L0], cache=cache)
        assert np.allclose(pre, want[:, :L0], atol=2e-4, rtol=1e-4)
        for t in range(steps):
            lg = ref(toks[:, L0 + t:L0 + t + 1], cache=cache)
            assert np.allclose(lg[:, 0], want[:, L0 + t], atol=2e-4, rtol=1e-4), (paged, t)
        assert cache[0].offsets == [L0 + steps] * B
