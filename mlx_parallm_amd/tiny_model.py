"""Build a small local checkpoint for smoke tests -- the job of the reference's
``scripts/build_tiny_model.py`` (defaults :108-118: hidden 64, 8 layers, 4 heads, 4 kv heads,
intermediate 128, rope_theta 1e4, tied embeddings, int4 group 64; vocab fallback 151936 :125;
rms_norm_eps 1e-6 :89) without MLX:

  * weights are drawn with a seeded NumPy generator using MLX's layer initialisers
    (nn.Linear: U(-1/sqrt(K), 1/sqrt(K)); nn.Embedding: N(0, 1/sqrt(H)); RMSNorm: ones) --
    ``mx.random.seed`` streams cannot be reproduced, so the VALUES differ from a reference-built
    tiny model while the architecture, dtypes and file format are the same;
  * ``model.safetensors`` carries ``metadata={"format": "mlx"}`` (utils.py:870) and, when
    quantised, the ``.weight``(uint32)/``.scales``/``.biases`` triples of ``nn.quantize``;
    ``config.json`` carries ``quantization`` + ``quantization_config`` (build_tiny_model.py:98-100);
  * the reference copies tokenizer files from ``models/hermes-qwen3-14b-4bit`` (git-ignored and
    absent); here a byte-level tokenizer is generated instead (``build_byte_tokenizer``).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np
import torch

from .quant import quantize

_TORCH_DT = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}

CHATML_TEMPLATE = (
    "{% for message in messages %}{{ '<|im_start|>' + message['role'] + '\n' + message['content'] + '<|im_end|>' + '\n' }}"
    "{% endfor %}{% if add_generation_prompt %}{{ '<|im_start|>assistant\n' }}{% endif %}"
)


def build_config(*, model_type: str, vocab_size: int, hidden_size: int, num_hidden_layers: int,
                 intermediate_size: int, num_attention_heads: int, num_key_value_heads: Optional[int],
                 rope_theta: float, tie_word_embeddings: bool, quantization: Optional[Dict[str, Any]],
                 head_dim: Optional[int] = None, rms_norm_eps: float = 1e-6,
                 max_position_embeddings: int = 4096) -> Dict[str, Any]:
    """build_tiny_model.py:70-101 (+ head_dim / max_position_embeddings for qwen3)."""
    cfg: Dict[str, Any] = {
        "model_type": model_type,
        "hidden_size": int(hidden_size),
        "num_hidden_layers": int(num_hidden_layers),
        "intermediate_size": int(intermediate_size),
        "num_attention_heads": int(num_attention_heads),
        "num_key_value_heads": int(num_key_value_heads) if num_key_value_heads else int(num_attention_heads),
        "rms_norm_eps": float(rms_norm_eps),
        "vocab_size": int(vocab_size),
        "rope_theta": float(rope_theta),
        "rope_traditional": False,
        "rope_scaling": None,
        "attention_bias": False,
        "mlp_bias": False,
        "tie_word_embeddings": bool(tie_word_embeddings),
        "max_position_embeddings": int(max_position_embeddings),
    }
    if model_type == "qwen3" or head_dim is not None:
        cfg["head_dim"] = int(head_dim or hidden_size // num_attention_heads)
    if quantization is not None:
        cfg["quantization"] = dict(quantization)
        cfg["quantization_config"] = dict(quantization)
    return cfg


def init_weights(cfg: Dict[str, Any], seed: int = 0, dtype: str = "float32", norm_jitter: float = 0.0,
                 weight_std: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Random weights keyed like an MLX / HF checkpoint.  ``weight_std``: N(0, std) instead of
    the MLX initialisers (SURVEY §8d synthetic weights)."""
    rng = np.random.default_rng(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    tdt = _TORCH_DT[dtype]

    def lin(n, k):
        if weight_std is not None:
            a = rng.standard_normal((n, k), dtype=np.float32) * np.float32(weight_std)
        else:
            s = 1.0 / np.sqrt(k)
            a = rng.uniform(-s, s, size=(n, k)).astype(np.float32)
        return torch.from_numpy(a).to(tdt)

    def norm(n):
        a = np.ones(n, dtype=np.float32)
        if norm_jitter:
            a = a + norm_jitter * rng.standard_normal(n).astype(np.float32)
        return torch.from_numpy(a).to(tdt)

    w: Dict[str, torch.Tensor] = {}
    std = weight_std if weight_std is not None else 1.0 / np.sqrt(H)
    w["model.embed_tokens.weight"] = torch.from_numpy(
        (rng.standard_normal((V, H), dtype=np.float32) * np.float32(std))).to(tdt)
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        w[p + "self_attn.q_proj.weight"] = lin(nh * D, H)
        w[p + "self_attn.k_proj.weight"] = lin(nkv * D, H)
        w[p + "self_attn.v_proj.weight"] = lin(nkv * D, H)
        w[p + "self_attn.o_proj.weight"] = lin(H, nh * D)
        if cfg["model_type"] == "qwen3":
            w[p + "self_attn.q_norm.weight"] = norm(D)
            w[p + "self_attn.k_norm.weight"] = norm(D)
        w[p + "mlp.gate_proj.weight"] = lin(I, H)
        w[p + "mlp.down_proj.weight"] = lin(H, I)
        w[p + "mlp.up_proj.weight"] = lin(I, H)
        w[p + "input_layernorm.weight"] = norm(H)
        w[p + "post_attention_layernorm.weight"] = norm(H)
    w["model.norm.weight"] = norm(H)
    if not cfg["tie_word_embeddings"]:
        w["lm_head.weight"] = lin(V, H)
    return w


def quantize_weights(w: Dict[str, torch.Tensor], group_size: int, bits: int) -> Dict[str, torch.Tensor]:
    """``nn.quantize``: every Linear and Embedding (2-D ``.weight``) becomes a packed triple."""
    out: Dict[str, torch.Tensor] = {}
    for k, t in w.items():
        if k.endswith(".weight") and t.ndim == 2 and t.shape[1] % group_size == 0:
            base = k[: -len(".weight")]
            packed, scales, biases = quantize(t, group_size, bits)
            out[base + ".weight"] = packed
            out[base + ".scales"] = scales
            out[base + ".biases"] = biases
        else:
            out[k] = t
    return out


def build_byte_tokenizer(dst: Path) -> int:
    """Writes a byte-level HF fast tokenizer (256 byte tokens + <|endoftext|>, <|im_start|>,
    <|im_end|>) with a ChatML template.  Returns its vocabulary size."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {ch: i for i, ch in enumerate(alphabet)}
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    fast = PreTrainedTokenizerFast(
        tokenizer_object=tok, eos_token="<|endoftext|>",
        additional_special_tokens=["<|im_start|>", "<|im_end|>"],
    )
    fast.chat_template = CHATML_TEMPLATE
    dst.mkdir(parents=True, exist_ok=True)
    fast.save_pretrained(str(dst))
    return len(fast)


def build_tiny_model(dst, *, seed: int = 0, model_type: str = "llama", hidden_size: int = 64, layers: int = 8,
                     heads: int = 4, kv_heads: int = 4, intermediate_size: int = 128, rope_theta: float = 10000.0,
                     tie_word_embeddings: bool = True, quantize_model: bool = True, q_bits: int = 4,
                     q_group_size: int = 64, vocab_size: Optional[int] = None, dtype: str = "float32",
                     head_dim: Optional[int] = None, with_tokenizer: bool = True, norm_jitter: float = 0.0,
                     weight_std: Optional[float] = None, rms_norm_eps: float = 1e-6,
                     max_position_embeddings: int = 4096) -> Dict[str, Any]:
    """scripts/build_tiny_model.py:104-160.  Returns the config dict it wrote."""
    from safetensors.torch import save_file

    dst = Path(dst)
    dst.mkdir(parents=True, exist_ok=True)
    tok_vocab = build_byte_tokenizer(dst) if with_tokenizer else 0
    if vocab_size is None:
        vocab_size = 151936                                  # build_tiny_model.py:125 fallback
    if vocab_size < tok_vocab:
        raise ValueError(f"vocab_size {vocab_size} smaller than the tokenizer's {tok_vocab}")
    quant = {"group_size": int(q_group_size), "bits": int(q_bits)} if quantize_model else None
    cfg = build_config(model_type=model_type, vocab_size=vocab_size, hidden_size=hidden_size,
                       num_hidden_layers=layers, intermediate_size=intermediate_size,
                       num_attention_heads=heads, num_key_value_heads=kv_heads, rope_theta=rope_theta,
                       tie_word_embeddings=tie_word_embeddings, quantization=quant, head_dim=head_dim,
                       rms_norm_eps=rms_norm_eps, max_position_embeddings=max_position_embeddings)
    w = init_weights(cfg, seed=seed, dtype=dtype, norm_jitter=norm_jitter, weight_std=weight_std)
    if quantize_model:
        w = quantize_weights(w, q_group_size, q_bits)
    save_file({k: v.contiguous() for k, v in w.items()}, str(dst / "model.safetensors"), metadata={"format": "mlx"})
    (dst / "config.json").write_text(json.dumps(cfg, indent=4, sort_keys=True))   # utils.save_config sorts keys
    return cfg
