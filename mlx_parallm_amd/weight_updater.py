"""LoRA hot-swap on a live engine (mirror of ``mlx_parallm/rl_training/weight_updater.py:17-90``).

``apply_lora_update(model, adapter_path, lock=...)`` re-reads an adapter directory and replaces the LoRA
factors of the running model in place -- < 10 MB over PCIe, no weight reload; in a multi-GPU deployment each
rank calls it for its own replica.  Accepted layouts, in the reference's order of preference:
``adapter_config.json`` + ``adapters.safetensors`` (mlx-lm ``load_adapters``), else ``adapter.npz``, else
``adapters.safetensors`` / ``model*.safetensors`` without a config; keys
``model.layers.<i>.<proj>.lora_a`` (K, r) / ``.lora_b`` (r, N).  Without a config the scale of the currently
loaded adapter is kept (``default_scale`` if there is none).
"""
from __future__ import annotations

import glob
import json
import logging
import os
import re
from threading import RLock
from typing import Any, Dict, Optional

import numpy as np

_KEY = re.compile(r"^(?:model\.)?layers\.(\d+)\.(.+)\.lora_a$")


def _load_adapter_arrays(path: str) -> Dict[str, Any]:
    npz = os.path.join(path, "adapter.npz")
    if os.path.exists(npz):
        with np.load(npz) as z:
            return {k: z[k] for k in z.files}
    from safetensors.numpy import load_file

    st = os.path.join(path, "adapters.safetensors")
    if os.path.exists(st):
        return dict(load_file(st))
    files = glob.glob(os.path.join(path, "model*.safetensors"))
    if not files:
        raise FileNotFoundError(f"No adapter weights found in {path}")
    out: Dict[str, Any] = {}
    for f in files:
        out.update(load_file(f))
    return out


def apply_lora_update(model, adapter_path: str, *, lock: Optional[RLock] = None, default_scale: float = 10.0) -> int:
    """Returns the number of (layer, projection) pairs updated."""
    from .utils import load_adapters

    def run() -> int:
        cfg_file = os.path.join(adapter_path, "adapter_config.json")
        if os.path.exists(cfg_file) and os.path.exists(os.path.join(adapter_path, "adapters.safetensors")):
            load_adapters(model, adapter_path)
            cfg = json.load(open(cfg_file))
            model._lora_scale = float(cfg["lora_parameters"]["scale"])
            return int(cfg["num_layers"]) * len(cfg["lora_parameters"].get("keys") or ["self_attn.q_proj", "self_attn.v_proj"])
        arrays = _load_adapter_arrays(adapter_path)
        scale = float(getattr(model, "_lora_scale", default_scale))
        n = 0
        for k, a in arrays.items():
            m = _KEY.match(k)
            if not m:
                continue
            b = arrays.get(k[: -len("lora_a")] + "lora_b")
            if b is None:
                continue
            model.engine.set_lora(int(m.group(1)), m.group(2), np.asarray(a), np.asarray(b), scale)
            n += 1
        if n == 0:
            logging.warning(f"No matching adapter parameters found in {adapter_path}")
        return n

    if lock is not None:
        with lock:
            return run()
    return run()


def apply_lora_update_for_record(record, adapter_path: str, *, lock: Optional[RLock] = None) -> None:
    """weight_updater.py:80-90: update the registry record's model and remember the adapter path."""
    if record.model_instance is None:
        raise RuntimeError("No model instance present in the InternalModelRecord.")
    apply_lora_update(record.model_instance, adapter_path, lock=lock)
    record.adapter_path = adapter_path
    try:
        record.model_instance.eval()
    except Exception:
        logging.debug("Model eval() after adapter update failed; continuing.")
