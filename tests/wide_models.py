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


def model_kwargs(family: str, precision: str, seed: int, layers: int = LAYERS) -> dict:
    kw = dict(FAMILIES[family])
    kw.update(seed=seed, layers=layers, dtype=("float16" if precision == "f16" else "bfloat16"), tie_word_embeddings=False, weight_std=0.02,
              quantize_model=(precision in ("int4", "int8")), q_bits=(8 if precision == "int8" else 4), q_group_size=64,
              with_tokenizer=False,
              max_position_embeddings=MAX_POS)
    return kw


FULL_LAYERS = {"mistral-7b": 32, "qwen3-14b": 40}      # llama.py:228 / qwen3.py:182 run every block of the real models


def build_checkpoint(dst, family: str, precision: str, seed: int, layers: int = LAYERS) -> dict:
    from mlx_parallm_amd.tiny_model import build_tiny_model

    if layers != LAYERS:
        return build_checkpoint_streamed(dst, family, precision, seed, layers)
    return build_tiny_model(dst, **model_kwargs(family, precision, seed))


def build_checkpoint_streamed(dst, family: str, precision: str, seed: int, layers: int) -> dict:
    """The FULL-DEPTH checkpoints (7.2e9 / 14.8e9 weights): same distributions, dtypes and file format as build_tiny_model,
    but every tensor has its own generator -- np.random.default_rng([seed, index]) in a fixed tensor order -- so that the
    tensors can be drawn (and quantised, in row chunks) by a pool of threads: minutes -> seconds on the GPU box, where this
    runs inside the timed GPU test suite.  Oracle side and device side both call this function."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from safetensors.torch import save_file

    from mlx_parallm_amd import quant
    from mlx_parallm_amd.tiny_model import build_config

    kw = model_kwargs(family, precision, seed, layers)
    dst = Path(dst)
    dst.mkdir(parents=True, exist_ok=True)
    qz = {"group_size": int(kw["q_group_size"]), "bits": int(kw["q_bits"])} if kw["quantize_model"] else None
    cfg = build_config(model_type=kw["model_type"], vocab_size=kw["vocab_size"], hidden_size=kw["hidden_size"],
                       num_hidden_layers=layers, intermediate_size=kw["intermediate_size"],
                       num_attention_heads=kw["heads"], num_key_value_heads=kw["kv_heads"], rope_theta=kw["rope_theta"],
                       tie_word_embeddings=False, quantization=qz, head_dim=kw.get("head_dim"),
                       rms_norm_eps=kw["rms_norm_eps"], max_position_embeddings=kw["max_position_embeddings"])
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    tdt = torch.float16 if kw["dtype"] == "float16" else torch.bfloat16
    mats = [("model.embed_tokens", V, H)]
    for i in range(layers):
        p = f"model.layers.{i}."
        mats += [(p + "self_attn.q_proj", nh * D, H), (p + "self_attn.k_proj", nkv * D, H), (p + "self_attn.v_proj", nkv * D, H),
                 (p + "self_attn.o_proj", H, nh * D), (p + "mlp.gate_proj", I, H), (p + "mlp.down_proj", H, I),
                 (p + "mlp.up_proj", I, H)]
    mats.append(("lm_head", V, H))
    out = {}
    std = np.float32(kw["weight_std"])

    def make(item):
        idx, (name, n, k) = item
        rng = np.random.default_rng([seed, idx])
        res = {}
        parts = []
        rows = max(1, (8 << 20) // k)                      # ~8 M elements per piece
        for r0 in range(0, n, rows):
            a = torch.from_numpy(rng.standard_normal((min(rows, n - r0), k), dtype=np.float32) * std).to(tdt)
            parts.append(quant.quantize(a, qz["group_size"], qz["bits"]) if qz else a)
        if qz:
            res[name + ".weight"] = torch.cat([q[0] for q in parts])
            res[name + ".scales"] = torch.cat([q[1] for q in parts])
            res[name + ".biases"] = torch.cat([q[2] for q in parts])
        else:
            res[name + ".weight"] = torch.cat(parts)
        return res

    nthreads = max(2, min(16, (os.cpu_count() or 8)))
    old = torch.get_num_threads()
    torch.set_num_threads(1)                               # the pool is the parallelism
    try:
        with ThreadPoolExecutor(nthreads) as ex:
            for res in ex.map(make, list(enumerate(mats))):
                out.update(res)
    finally:
        torch.set_num_threads(old)
    ones = lambda n: torch.ones(n, dtype=tdt)              # noqa: E731  (RMSNorm weights: ones, as nn.RMSNorm initialises)
    for i in range(layers):
        p = f"model.layers.{i}."
        if cfg["model_type"] == "qwen3":
            out[p + "self_attn.q_norm.weight"] = ones(D)
            out[p + "self_attn.k_norm.weight"] = ones(D)
        out[p + "input_layernorm.weight"] = ones(H)
        out[p + "post_attention_layernorm.weight"] = ones(H)
    out["model.norm.weight"] = ones(H)
    save_file({k: v.contiguous() for k, v in out.items()}, str(dst / "model.safetensors"), metadata={"format": "mlx"})
    (dst / "config.json").write_text(json.dumps(cfg, indent=4, sort_keys=True))
    return cfg


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
