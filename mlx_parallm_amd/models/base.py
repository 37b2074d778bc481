"""KV-cache duck types of the reference (``mlx_parallm/models/base.py:42-150``).

In the reference a cache is a per-layer Python object holding MLX arrays and the model's
attention calls ``update_and_fetch`` on it.  Here the append is fused into the HIP decode
kernels, so a per-layer object is only a *view* onto one device-side ``mi_kv`` handle that
covers every layer of the batch; what stays observable is the bookkeeping the reference's
callers read: ``offsets`` / ``offset`` / ``step`` / ``batch_size`` / ``reset``.

KV dtype follows the class, as in the reference:
  * ``BatchedKVCache``  -> buffers in the model dtype            (base.py:71-72)
  * ``PagedKVCache``    -> float32 buffers, which promotes everything after the layer-0
                           attention to float32                    (base.py:111-112, quirk Q2)
"""
from __future__ import annotations

import inspect
from dataclasses import dataclass
from typing import List, Optional

import numpy as np


def create_additive_causal_mask(N: int, offset: int = 0):
    """(N, offset+N) float32, -1e9 where query_pos < key_pos (base.py:6-14).  Kept for API
    parity only: the HIP attention kernel applies this mask by construction (query t of a row
    with ``off`` cached tokens reads keys [0, off+t]) and never materialises it."""
    rinds = np.arange(offset + N)
    linds = np.arange(offset, offset + N) if offset else rinds
    return (linds[:, None] < rinds[None]).astype(np.float32) * np.float32(-1e9)


def create_additive_causal_mask_variable(N: int, offsets, total_length: int):
    """(B, N, total_length) per-row masks (base.py:17-40)."""
    if not isinstance(offsets, (list, tuple)):
        return create_additive_causal_mask(N, int(offsets))[None]
    rinds = np.arange(total_length)
    out = []
    for off in offsets:
        linds = np.arange(int(off), int(off) + N)
        out.append((linds[:, None] < rinds[None]).astype(np.float32) * np.float32(-1e9))
    return np.stack(out, axis=0)


class _KVGroup:
    """The shared state behind one ``List[BatchedKVCache]``: created unbound, bound to a
    device handle by the first ``model(...)`` call that receives the list."""

    def __init__(self, batch_size: int, kv_dtype: str, step: int = 256):
        self.batch_size = batch_size
        self.kv_dtype = kv_dtype
        self.step = step
        self.handle = None          # engine.KVCache

    def bind(self, engine, capacity: int):
        from ..engine import KVCache

        if self.handle is None or self.handle.engine is not engine or self.handle.batch_size != self.batch_size:
            if self.handle is not None:
                self.handle.close()
            self.handle = KVCache(engine, self.batch_size, capacity, self.kv_dtype, self.step)
        self.handle.step = self.step
        return self.handle


class BatchedKVCache:
    _KV_DTYPE = "model"

    def __init__(self, head_dim, n_kv_heads, batch_size=1):
        self.n_kv_heads = n_kv_heads
        self.head_dim = head_dim
        self.batch_size = batch_size
        self.keys = None            # device resident; not exposed as arrays
        self.values = None
        self._step = 256
        self._group: Optional[_KVGroup] = None

    # -- grouping (one device handle per list of per-layer caches)
    def _ensure_group(self) -> _KVGroup:
        if self._group is None:
            self._group = _KVGroup(self.batch_size, self._KV_DTYPE, self._step)
        return self._group

    @property
    def step(self) -> int:
        return self._step

    @step.setter
    def step(self, v: int) -> None:
        self._step = int(v)
        if self._group is not None:
            self._group.step = int(v)

    @property
    def offsets(self) -> List[int]:
        g = self._group
        if g is None or g.handle is None:
            return [0] * self.batch_size
        return g.handle.offsets

    @property
    def offset(self) -> int:
        return max(self.offsets) if self.batch_size else 0

    def reset(self, batch_size: Optional[int] = None) -> None:           # base.py:53-64 / 146-149
        g = self._group
        if batch_size is not None and batch_size != self.batch_size:
            self.batch_size = batch_size
            if g is not None:
                if g.handle is not None and g.batch_size != batch_size:
                    g.handle.close()
                    g.handle = None
                g.batch_size = batch_size
            return
        if g is not None and g.handle is not None:
            g.handle.reset()

    def update_and_fetch(self, keys, values):
        raise NotImplementedError(
            "update_and_fetch is fused into the MI355X decode kernels (RoPE + KV append + attention); "
            "pass the cache list to model(...) / generate_step(...) instead.")


class PagedKVCache(BatchedKVCache):
    _KV_DTYPE = "float32"

    def __init__(self, head_dim, n_kv_heads, batch_size=1):
        super().__init__(head_dim, n_kv_heads, batch_size)

    @property
    def offsets_list(self) -> List[int]:
        return self.offsets


def make_cache_list(klass, head_dim: int, kv_heads: List[int], batch_size: int, step: Optional[int] = None):
    """One cache object per layer, all views of one group (``_KVPool.get``, utils.py:208-223)."""
    caches = [klass(head_dim, n, batch_size) for n in kv_heads]
    group = _KVGroup(batch_size, klass._KV_DTYPE, step or 256)
    for c in caches:
        c._group = group
        if step is not None:
            c._step = step
    return caches


def group_of(cache_list) -> _KVGroup:
    """The shared group of a cache list (creating one if the caller built the list by hand,
    e.g. ``[PagedKVCache(model.head_dim, n, B) for n in kv_heads]``, utils.py:394)."""
    first = cache_list[0]
    g = first._ensure_group()
    for c in cache_list[1:]:
        if c._group is not g:
            c._group = g
    return g


@dataclass
class BaseModelArgs:
    @classmethod
    def from_dict(cls, params):
        return cls(**{k: v for k, v in params.items() if k in inspect.signature(cls).parameters})
