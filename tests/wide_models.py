"""Truncated-depth checkpoints at PRODUCTION widths for the parity tests at BASELINE.json's dimensions
(configs 2-5): the Mistral-7B and Qwen3-14B layer shapes (hidden / heads / intermediate / vocabulary /
eps / rope base of `bench.SHAPES`) with 2 decoder blocks, seeded N(0, 0.02^2) weights,
optionally MLX-affine int4 g64 and a rank-16 LoRA adapter on q/v of the last block
(rl_training/lora_init.py:68-72 defaults: rank 16, scale 10; A ~ U(+-1/sqrt(K)), B ~ N(0, 0.01^2)).

Used by tests/golden/make_golden_wide.py (oracle side, in the build container) and
tests/test_gpu_golden_wide.py (HIP side, on the GPU box): both build the SAME files from the same
seeds with the CPU generators below, so only small outputs need to be committed.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

FAMILIES = {
    # layer shapes of Mistral-7B-Instruct-v0.1 (BASELINE configs 2, 3)
    "mistral-7b": dict(model_type="llama", hidden_size=4096, heads=32, kv_heads=8, intermediate_size=14336,
                       vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0),
    # layer shapes of Qwen3-14B (BASELINE configs 4, 5): q/k norms, 5 query heads per kv head
    "qwen3-14b": dict(model_type="qwen3", hidden_size=5120, heads=40, kv_heads=8, head_dim=128, intermediate_size=17408,
                      vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1000000.0),
}
LAYERS = 2
MAX_POS = 2048


def model_kwargs(family: str, precision: str, seed: int) -> dict:
    kw = dict(FAMILIES[family])
    kw.update(seed=seed, layers=LAYERS, dtype=("float16" if precision == "f16" else "bfloat16"), tie_word_embeddings=False, weight_std=0.02,
              quantize_model=(precision in ("int4", "int8")), q_bits=(8 if precision == "int8" else 4), q_group_size=64,
              with_tokenizer=False,
              max_position_embeddings=MAX_POS)
    return kw


def build_checkpoint(dst, family: str, precision: str, seed: int) -> dict:
    from mlx_parallm_amd.tiny_model import build_tiny_model

    return build_tiny_model(dst, **model_kwargs(family, precision, seed))


def build_adapter(dst, cfg: dict, seed: int, num_layers: int = 1, rank: int = 16, scale: float = 10.0) -> None:
    """adapters.safetensors + adapter_config.json as rl_training/lora_init.py:140-153 writes them."""
    import torch
    from safetensors.torch import save_file

    dst = Path(dst)
    dst.mkdir(parents=True, exist_ok=True)
    H, nh, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    rng = np.random.default_rng(seed)
    w = {}
    for i in range(cfg["num_hidden_layers"] - num_layers, cfg["num_hidden_layers"]):
        for key, n in (("self_attn.q_proj", nh * D), ("self_attn.v_proj", nkv * D)):
            w[f"model.layers.{i}.{key}.lora_a"] = torch.from_numpy(
                (rng.uniform(-1, 1, (H, rank)) / np.sqrt(H)).astype(np.float32))
            w[f"model.layers.{i}.{key}.lora_b"] = torch.from_numpy(
                (rng.standard_normal((rank, n)) * 0.01).astype(np.float32))
    save_file(w, str(dst / "adapters.safetensors"))
    (dst / "adapter_config.json").write_text(json.dumps({
        "fine_tune_type": "lora", "num_layers": num_layers,
        "lora_parameters": {"rank": rank, "scale": scale, "dropout": 0.05,
                            "keys": ["self_attn.q_proj", "self_attn.v_proj"]}}))


def prompts_for(case: dict, vocab: int) -> np.ndarray:
    """Seeded token ids (B, L0); `ragged` left-pads rows with id 1 (pads ARE attended, quirk Q1)."""
    rng = np.random.default_rng(case["prompt_seed"])
    p = rng.integers(3, vocab, size=(case["B"], case["L0"]))
    if case.get("ragged"):
        for b in range(case["B"]):
            p[b, : int(rng.integers(0, case["L0"] // 2))] = 1
    return p.astype(np.int32)
