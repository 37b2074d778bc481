"""generate_step / batch_generate token loop + model-dir loader, restated
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows ``mlx_parallm/utils.py:315-427`` (generate_step: prefill once, then one token per
call; sampling closure; cache from ``_kv_pool.get(..., paged=True)`` :390-394) and
``mlx_parallm/utils.py:620-747`` (load_config / load_model / load: config.json, glob
``model*.safetensors``, quantised modules = those with a ``.scales`` key :679-690,
``load_adapters`` :742-744 with the ``adapter_config.json`` layout of
``rl_training/lora_init.py:140-153``).
"""
from __future__ import annotations

import glob
import json
from pathlib import Path
from typing import Dict, Iterator, Optional, Tuple

import numpy as np

from .ref_model import Linear, RefConfig, RefModel
from .ref_sample import sample

_TORCH_DT = {"torch.float32": "float32", "torch.bfloat16": "bfloat16", "torch.float16": "float16"}


def _load_safetensors(path: str) -> Dict[str, Tuple[np.ndarray, str]]:
    import torch  # only used as a bf16-capable safetensors reader
    from safetensors.torch import load_file

    out = {}
    for k, t in load_file(path).items():
        if t.dtype in (torch.uint32, torch.int32):
            out[k] = (t.view(torch.int32).numpy().view(np.uint32), "uint32")
        else:
            out[k] = (t.to(torch.float32).numpy(), _TORCH_DT[str(t.dtype)])
    return out


def _load_safetensors_compact(path: str) -> Dict[str, Tuple[np.ndarray, str]]:
    """As _load_safetensors, but 16-bit matrices stay 16-bit patterns (uint16, dtype kept in the tag as "w16:<dtype>"): the
    full-depth production-width checkpoints do not fit this container as float32 (ref_model.COMPACT)."""
    import torch
    from safetensors import safe_open

    out = {}
    with safe_open(path, framework="pt") as f:
        for k in f.keys():
            t = f.get_tensor(k)
            if t.dtype in (torch.uint32, torch.int32):
                out[k] = (t.view(torch.int32).numpy().view(np.uint32), "uint32")
            elif t.ndim == 2 and t.dtype in (torch.bfloat16, torch.float16) and not (k.endswith(".scales") or k.endswith(".biases")):
                out[k] = (t.contiguous().view(torch.int16).numpy().view(np.uint16), "w16:" + _TORCH_DT[str(t.dtype)])
            else:
                out[k] = (t.to(torch.float32).numpy(), _TORCH_DT[str(t.dtype)])
    return out


def load_weights(model_dir: str, cfg_dict: dict, compact: bool = False) -> Dict[str, object]:
    files = sorted(glob.glob(str(Path(model_dir) / "model*.safetensors")))
    if not files:
        raise FileNotFoundError(f"No safetensors found in {model_dir}")          # utils.py:663-665
    raw: Dict[str, Tuple[np.ndarray, str]] = {}
    for f in files:
        raw.update(_load_safetensors_compact(f) if compact else _load_safetensors(f))
    q = cfg_dict.get("quantization")
    w: Dict[str, object] = {}
    for name, (arr, dt) in raw.items():
        if "rotary_emb.inv_freq" in name:                                       # llama.py:255-259
            continue
        if name.endswith(".weight"):
            base = name[: -len(".weight")]
            if q is not None and (base + ".scales") in raw:                       # utils.py:681-684
                w[base] = Linear(dtype=raw[base + ".scales"][1], packed=arr,
                                 scales=raw[base + ".scales"][0], biases=raw[base + ".biases"][0],
                                 group_size=int(q["group_size"]), bits=int(q["bits"]))
            elif arr.ndim == 1:
                w[base] = (arr, dt)                                               # RMSNorm weight
            elif dt.startswith("w16:"):
                w[base] = Linear(dtype=dt[4:], w16=arr)
            else:
                w[base] = Linear(dtype=dt, weight=arr)
    return w


def apply_adapters(weights: Dict[str, object], n_layers: int, adapter_dir: str) -> None:
    """mlx-lm ``load_adapters`` as used at utils.py:742-744: wrap the target Linear modules of
    the LAST ``num_layers`` blocks with LoRALinear, then load ``adapters.safetensors``."""
    cfg = json.loads((Path(adapter_dir) / "adapter_config.json").read_text())
    lp = cfg["lora_parameters"]
    raw = _load_safetensors(str(Path(adapter_dir) / "adapters.safetensors"))
    keys = lp.get("keys") or ["self_attn.q_proj", "self_attn.v_proj"]
    for i in range(n_layers - int(cfg["num_layers"]), n_layers):
        for key in keys:
            base = f"model.layers.{i}.{key}"
            lin: Linear = weights[base]
            a = raw.get(base + ".lora_a")
            b = raw.get(base + ".lora_b")
            if a is None or b is None:
                continue
            lin.lora_a, lin.lora_b = a[0], b[0]
            lin.lora_dtype = a[1]
            lin.lora_scale = float(lp["scale"])


def load(model_dir: str, adapter_path: Optional[str] = None, max_pos: int = 4096, compact: bool = False) -> RefModel:
    cfg_path = Path(model_dir) / "config.json"
    if not cfg_path.exists():
        raise FileNotFoundError(str(cfg_path))                                    # utils.py:620-627
    cfg_dict = json.loads(cfg_path.read_text())
    if {"mistral": "llama"}.get(cfg_dict["model_type"], cfg_dict["model_type"]) not in ("llama", "qwen3"):
        raise ValueError(f"Model type {cfg_dict['model_type']} not supported.")   # utils.py:60-65
    cfg = RefConfig.from_dict(cfg_dict)
    weights = load_weights(model_dir, cfg_dict, compact=compact)
    if adapter_path is not None:
        apply_adapters(weights, cfg.num_hidden_layers, adapter_path)
    return RefModel(cfg, weights, max_pos=max_pos)


def generate_step(prompts: np.ndarray, model: RefModel, temp: float = 0.0,
                  repetition_penalty: Optional[float] = None, repetition_context_size: int = 20,
                  top_p: float = 1.0, logit_bias=None, cache=None, uniforms_fn=None, paged: bool = True,
                  return_logits: bool = False, last_only: bool = False) -> Iterator:
    """utils.py:315-427.  Yields (tokens (B,1), probs (B,1)) [+ logits, logprobs if asked].
    ``uniforms_fn(step) -> (B,) uniforms`` supplies the noise for temp>0 (see ref_sample)."""
    if repetition_penalty:
        raise NotImplementedError("repetition_penalty not supported.")          # utils.py:366-367
    y = np.asarray(prompts, dtype=np.int64)
    B = y.shape[0]
    if cache is None:
        cache = model.make_cache(B, paged=paged)                                # utils.py:390-394
    step = 0
    while True:
        logits = model(y, cache=cache, last_only=last_only)[:, -1, :]           # utils.py:403-404
        u = uniforms_fn(step) if (temp != 0 and uniforms_fn is not None) else None
        s = sample(logits, temp=temp, top_p=top_p, logit_bias=logit_bias, uniforms=u)
        y = s["tokens"]
        if return_logits:
            yield y, s["probs"], logits, s["logprobs"]
        else:
            yield y, s["probs"]
        step += 1
