"""Llama / Mistral model handle (mirror of ``mlx_parallm/models/llama.py``).

``ModelArgs`` keeps the reference's fields and validation (llama.py:15-46).  ``Model`` keeps
the duck type the generation loop and the server rely on -- ``model(inputs, cache=...)``,
``.layers``, ``.head_dim``, ``.n_kv_heads``, ``.sanitize``, ``.load_weights``, ``.eval``
(llama.py:234-271) -- but owns no arrays: the transformer itself
(Attention / MLP / TransformerBlock, llama.py:49-231) runs inside libmi355_decode.so.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple, Union

import numpy as np

from .base import BaseModelArgs, BatchedKVCache, PagedKVCache, group_of, make_cache_list


@dataclass
class ModelArgs(BaseModelArgs):
    model_type: str
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    rms_norm_eps: float
    vocab_size: int
    num_key_value_heads: Optional[int] = None
    attention_bias: bool = False
    mlp_bias: bool = False
    rope_theta: float = 10000
    rope_traditional: bool = False
    rope_scaling: Optional[Dict[str, Union[float, str]]] = None
    tie_word_embeddings: bool = True
    head_dim: Optional[int] = None
    max_position_embeddings: int = 4096

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.rope_scaling:                                                  # llama.py:36-46
            if "factor" not in self.rope_scaling:
                raise ValueError("rope_scaling must contain 'factor'")
            if "type" in self.rope_scaling:
                if self.rope_scaling["type"] != "linear":
                    raise ValueError("rope_scaling 'type' currently only supports 'linear'")
            elif "rope_type" in self.rope_scaling:
                if self.rope_scaling["rope_type"] not in ["llama3", "linear"]:
                    raise ValueError(f"rope_scaling 'rope_type' {self.rope_scaling['rope_type']} not supported")
        if self.attention_bias or self.mlp_bias:
            raise NotImplementedError("attention_bias / mlp_bias are not supported by the MI355X engine")
        if self.rope_traditional:
            raise NotImplementedError("rope_traditional=True is not supported by the MI355X engine")


class _LayerStub:
    """Placeholder so that ``len(model.layers)`` / ``model.layers[-n:]`` keep working."""

    def __init__(self, index: int):
        self.index = index


class Model:
    """``Model(args)`` -> device engine.  Weights arrive through ``load_weights``."""

    _ARCH = "llama"

    def __init__(self, args: ModelArgs, *, config: Optional[dict] = None, device: int = 0,
                 dtype: str = "bfloat16", max_positions: Optional[int] = None):
        from ..engine import Engine

        self.args = args
        self.model_type = args.model_type
        cfg = dict(config) if config is not None else {k: getattr(args, k) for k in args.__dataclass_fields__}
        cfg["model_type"] = self._ARCH
        if cfg.get("head_dim") is None:
            cfg.pop("head_dim", None)
        self.engine = Engine(cfg, device=device, max_positions=max_positions, act_dtype=dtype)
        self._layers = [_LayerStub(i) for i in range(args.num_hidden_layers)]
        self._param_names: List[str] = []

    # -- reference surface --------------------------------------------------------------
    def __call__(self, inputs, cache=None, last_only: bool = False) -> np.ndarray:
        """(B, L) ids -> float32 logits (B, L, V)  (llama.py:243-253).  ``cache`` is a list of
        ``BatchedKVCache``/``PagedKVCache`` views, or None (stateless: positions start at 0)."""
        tokens = np.asarray(inputs)
        if tokens.ndim == 1:
            tokens = tokens[None]
        B, L = tokens.shape
        temp = None
        if cache is None:
            temp = self.engine.new_kv(B, capacity=L, kv_dtype="model")
            handle = temp
        else:
            handle = self.bind_cache(cache, B, L)
        try:
            out = self.engine.forward(tokens, handle, all_positions=not last_only)
        finally:
            if temp is not None:
                temp.close()
        return out[:, None, :] if last_only else out

    def bind_cache(self, cache, batch: int, new_tokens: int):
        g = group_of(cache)
        if g.batch_size != batch:
            raise ValueError("PagedKVCache batch size mismatch")               # base.py:125
        h = g.bind(self.engine, capacity=max(g.step, new_tokens))
        return h

    def make_cache(self, batch_size: int, paged: bool = True, step: Optional[int] = None):
        kv_heads = [self.n_kv_heads] * len(self.layers)
        return make_cache_list(PagedKVCache if paged else BatchedKVCache, self.head_dim, kv_heads, batch_size, step)

    def sanitize(self, weights):                                               # llama.py:255-259
        return {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}

    def load_weights(self, items: Union[Iterable[Tuple[str, object]], Dict[str, object]], strict: bool = False):
        if isinstance(items, dict):
            items = items.items()
        items = list(items)
        self._param_names += [k for k, _ in items]
        skipped = self.engine.load_tensors(items, strict=strict)
        return skipped

    def finalize(self) -> None:
        self.engine.finalize()

    def eval(self):
        return self

    def parameters(self):
        """Names only: the arrays live in HBM behind the engine handle."""
        return {k: None for k in self._param_names}

    @property
    def layers(self):
        return self._layers

    @property
    def head_dim(self):                                                        # llama.py:265-267
        return self.args.head_dim or self.args.hidden_size // self.args.num_attention_heads

    @property
    def n_kv_heads(self):                                                      # llama.py:269-271
        return self.args.num_key_value_heads
