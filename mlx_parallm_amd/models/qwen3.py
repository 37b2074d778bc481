"""Qwen3 model handle (mirror of ``mlx_parallm/models/qwen3.py``).

Same duck type as llama plus per-head ``q_norm`` / ``k_norm`` RMSNorm applied before RoPE
(qwen3.py:42-43,65-70), which the engine fuses into its rope/append kernel.  ``ModelArgs``
restates the fields of mlx-lm's ``qwen3.ModelArgs`` that the reference imports (qwen3.py:7-13).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Union

from .base import BaseModelArgs
from .llama import Model as _LlamaModel


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    num_key_value_heads: int
    max_position_embeddings: int
    rope_theta: float
    head_dim: int
    tie_word_embeddings: bool = False
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None

    def __post_init__(self):
        if self.rope_scaling:
            kind = self.rope_scaling.get("type", self.rope_scaling.get("rope_type"))
            if kind not in (None, "default", "linear"):
                raise NotImplementedError(f"rope_scaling type {kind} is not supported by the MI355X engine")


class Model(_LlamaModel):
    _ARCH = "qwen3"

    def sanitize(self, weights):                                               # qwen3.py:211-214
        if self.args.tie_word_embeddings:
            weights.pop("lm_head.weight", None)
        return weights

    @property
    def head_dim(self):                                                        # qwen3.py:220-222
        return self.args.head_dim
