"""On-disk formats of the generation path: MLX-format (quantised) safetensors read / write.

Mirror of ``mlx_parallm/utils.py:759-968``: ``make_shards``, ``save_weights``, ``save_config``,
``quantize_model``, ``dequantize_model``, ``convert`` -- same names, arguments and file layout
(``model.safetensors`` or ``model-0000i-of-0000n.safetensors`` + ``model.safetensors.index.json``,
safetensors metadata ``{"format": "mlx"}``, ``config.json["quantization"] = {"group_size", "bits"}``,
packed ``<p>.weight`` uint32 + ``<p>.scales`` + ``<p>.biases``), so that a directory written here loads in
the reference and vice versa.  The tensors are torch CPU tensors (the engine is not involved: this is
file conversion, not the hot path); the affine quantiser is ``mlx_parallm_amd.quant``.
Hub upload (``upload_to_hub``, utils.py:782-833) is not available offline and is not provided.
"""
from __future__ import annotations

import copy
import glob
import json
import shutil
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from .quant import dequantize, quantize

MAX_FILE_SIZE_GB = 5            # utils.py:47

_DTYPES = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


def make_shards(weights: Dict[str, torch.Tensor], max_file_size_gb: int = MAX_FILE_SIZE_GB) -> List[Dict[str, torch.Tensor]]:
    """utils.py:759-779: greedy split in insertion order; a shard is closed before the tensor that would
    take it over the limit."""
    limit = max_file_size_gb << 30
    shards: List[Dict[str, torch.Tensor]] = []
    shard: Dict[str, torch.Tensor] = {}
    size = 0
    for k, v in weights.items():
        if size + _nbytes(v) > limit:
            shards.append(shard)
            shard, size = {}, 0
        shard[k] = v
        size += _nbytes(v)
    shards.append(shard)
    return shards


def save_weights(save_path: Union[str, Path], weights: Dict[str, torch.Tensor], *, donate_weights: bool = False,
                 max_file_size_gb: int = MAX_FILE_SIZE_GB) -> None:
    """utils.py:836-885."""
    from safetensors.torch import save_file

    save_path = Path(save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    shards = make_shards(weights, max_file_size_gb)
    n = len(shards)
    name_format = "model-{:05d}-of-{:05d}.safetensors" if n > 1 else "model.safetensors"
    index = {"metadata": {"total_size": sum(_nbytes(v) for v in weights.values())}, "weight_map": {}}
    if donate_weights:
        weights.clear()
    for i in range(n):
        shard, shards[i] = shards[i], None
        name = name_format.format(i + 1, n)
        save_file({k: v.contiguous() for k, v in shard.items()}, str(save_path / name), metadata={"format": "mlx"})
        for k in shard:
            index["weight_map"][k] = name
        del shard
    index["weight_map"] = {k: index["weight_map"][k] for k in sorted(index["weight_map"])}
    with open(save_path / "model.safetensors.index.json", "w") as f:
        json.dump(index, f, indent=4)


def save_config(config: dict, config_path: Union[str, Path]) -> None:
    """utils.py:910-930: drop ``_name_or_path``, sort keys, indent 4."""
    config.pop("_name_or_path", None)
    with open(config_path, "w") as f:
        json.dump(dict(sorted(config.items())), f, indent=4)


def _quantizable(name: str, t: torch.Tensor, group_size: int) -> bool:
    # nn.quantize's default predicate: modules with to_quantized (Linear, Embedding) whose input dimension is
    # a multiple of the group size -- i.e. 2-D ``.weight`` tensors; norms are 1-D
    return name.endswith(".weight") and t.ndim == 2 and t.shape[1] % group_size == 0 and t.is_floating_point()


def quantize_model(weights: Dict[str, torch.Tensor], config: dict, q_group_size: int, q_bits: int
                   ) -> Tuple[Dict[str, torch.Tensor], dict]:
    """utils.py:888-908 (``nn.quantize`` over the module tree) on the flat weight dict."""
    out: Dict[str, torch.Tensor] = {}
    for k, t in weights.items():
        if _quantizable(k, t, q_group_size):
            base = k[: -len(".weight")]
            out[base + ".weight"], out[base + ".scales"], out[base + ".biases"] = quantize(t, q_group_size, q_bits)
        else:
            out[k] = t
    qc = copy.deepcopy(config)
    qc["quantization"] = {"group_size": q_group_size, "bits": q_bits}
    return out, qc


def dequantize_model(weights: Dict[str, torch.Tensor], config: dict) -> Tuple[Dict[str, torch.Tensor], dict]:
    """Inverse of ``quantize_model`` (mlx-lm ``dequantize_model``): packed triples back to dense weights in
    the dtype of their scales; ``quantization`` leaves the config."""
    q = config.get("quantization")
    if not q:
        return dict(weights), copy.deepcopy(config)
    out: Dict[str, torch.Tensor] = {}
    for k, t in weights.items():
        if k.endswith(".scales") or k.endswith(".biases"):
            continue
        base = k[: -len(".weight")] if k.endswith(".weight") else None
        if base is not None and base + ".scales" in weights:
            s = weights[base + ".scales"]
            out[k] = dequantize(t, s, weights[base + ".biases"], int(q["group_size"]), int(q["bits"])).to(s.dtype)
        else:
            out[k] = t
    dc = copy.deepcopy(config)
    dc.pop("quantization", None)
    return out, dc


def load_weights_dir(model_path: Union[str, Path]) -> Dict[str, torch.Tensor]:
    """All ``model*.safetensors`` (or legacy ``weight*.safetensors``) of a directory as one dict (utils.py:655-670)."""
    from safetensors.torch import load_file

    files = sorted(glob.glob(str(Path(model_path) / "model*.safetensors"))) or \
        sorted(glob.glob(str(Path(model_path) / "weight*.safetensors")))
    if not files:
        raise FileNotFoundError(f"No safetensors found in {model_path}")
    weights: Dict[str, torch.Tensor] = {}
    for f in files:
        weights.update(load_file(f))
    return weights


def convert(hf_path: str, mlx_path: str = "mlx_model", quantize: bool = False, q_group_size: int = 64, q_bits: int = 4,
            dtype: str = "float16", upload_repo: Optional[str] = None, revision: Optional[str] = None,
            dequantize: bool = False) -> None:
    """utils.py:933-980: checkpoint directory -> MLX-format directory (weights cast to ``dtype``, float16 when
    quantising, as the reference does), optionally quantised or de-quantised; tokenizer files, ``*.py`` and the
    config travel along.  ``hf_path`` must be a local directory (no hub access); ``upload_repo`` is rejected."""
    from .utils import get_model_path, load_config

    if quantize and dequantize:
        raise ValueError("Choose either quantize or dequantize, not both.")
    if upload_repo is not None:
        raise NotImplementedError("upload_to_hub is not available in this build (no network)")
    print("[INFO] Loading")
    model_path = get_model_path(hf_path, revision=revision)
    config = load_config(model_path)
    weights = load_weights_dir(model_path)
    target = torch.float16 if quantize else _DTYPES[dtype]
    weights = {k: (v.to(target) if v.is_floating_point() and not _is_packed_aux(k, weights) else v) for k, v in weights.items()}
    if quantize:
        print("[INFO] Quantizing")
        weights, config = quantize_model(weights, config, q_group_size, q_bits)
    if dequantize:
        print("[INFO] Dequantizing")
        weights, config = dequantize_model(weights, config)
    out = Path(mlx_path)
    save_weights(out, weights, donate_weights=True)
    for f in glob.glob(str(model_path / "*.py")):
        shutil.copy(f, out)
    for pat in ("tokenizer*", "special_tokens_map.json", "vocab.*", "merges.txt", "added_tokens.json", "chat_template*",
                "generation_config.json"):
        for f in glob.glob(str(model_path / pat)):
            shutil.copy(f, out)
    save_config(config, config_path=out / "config.json")


def _is_packed_aux(name: str, weights: Dict[str, Any]) -> bool:
    """scales / biases of an already-quantised checkpoint keep their dtype (they define the dequantised dtype)."""
    return name.endswith(".scales") or (name.endswith(".biases") and name[: -len(".biases")] + ".scales" in weights)
