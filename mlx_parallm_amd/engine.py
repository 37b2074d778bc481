"""Python handles over the C ABI: ``Engine`` (one model replica on one GPU) and ``KVCache``.

These are the build's stand-ins for the objects the reference's hot path passes around:
the ``nn.Module`` returned by ``utils.load`` (utils.py:711-747) and the per-layer
``List[PagedKVCache]`` from ``_KVPool.get`` (utils.py:199-223).  All arithmetic happens in
libmi355_decode.so; PyTorch is used only to hold host/device tensors on their way in.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L

_DT_NAMES = {"float32": L.MI_F32, "bfloat16": L.MI_BF16, "float16": L.MI_F16}


def _tensor_info(t) -> Tuple[int, Tuple[int, ...], int, int, object]:
    """-> (data_ptr, shape, mi_dtype, on_device, keepalive) for a torch tensor or numpy array."""
    try:
        import torch
    except Exception:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(t, torch.Tensor):
        t = t.detach()
        if not t.is_contiguous():
            t = t.contiguous()
        dt = {torch.float32: L.MI_F32, torch.bfloat16: L.MI_BF16, torch.float16: L.MI_F16,
              torch.uint32: L.MI_U32, torch.int32: L.MI_U32}.get(t.dtype)
        if dt is None:
            raise ValueError(f"unsupported tensor dtype {t.dtype}")
        return t.data_ptr(), tuple(t.shape), dt, int(t.is_cuda), t
    a = np.ascontiguousarray(t)
    dt = {np.dtype(np.float32): L.MI_F32, np.dtype(np.float16): L.MI_F16,
          np.dtype(np.uint32): L.MI_U32, np.dtype(np.int32): L.MI_U32}.get(a.dtype)
    if dt is None:
        raise ValueError(f"unsupported array dtype {a.dtype}")
    return a.ctypes.data, tuple(a.shape), dt, 0, a


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


class SampleArgs:
    """Arguments of the reference's ``sample`` closure (utils.py:345-364)."""

    def __init__(self, temp: float = 0.0, top_p: float = 1.0, logit_bias: Optional[Dict[int, float]] = None,
                 uniforms: Optional[Sequence[float]] = None, seed: int = 0, top_logprobs: int = 0,
                 logprobs_at_temperature: bool = False, stream_position: int = -1):
        self.c = L.SampleParams()
        self.c.struct_size = C.sizeof(L.SampleParams)
        self.c.temperature = float(temp)
        self.c.top_p = float(top_p)
        self.c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.c.top_logprobs = int(top_logprobs)
        self.c.logprobs_at_temperature = 1 if logprobs_at_temperature else 0
        self.c.stream_position = int(stream_position)      # >= 0: Philox counter = the caller's step index (reproducible)
        self._keep = []
        if logit_bias:
            ids = np.ascontiguousarray(list(logit_bias.keys()), dtype=np.int32)
            vals = np.ascontiguousarray(list(logit_bias.values()), dtype=np.float32)
            self.c.n_logit_bias = len(ids)
            self.c.logit_bias_ids = ids.ctypes.data_as(C.POINTER(C.c_int32))
            self.c.logit_bias_values = vals.ctypes.data_as(C.POINTER(C.c_float))
            self._keep += [ids, vals]
        self.set_uniforms(uniforms)

    def set_row_params(self, temps, top_ps) -> None:
        """Per-row temperature / top_p (one entry per row of the step) instead of the scalars."""
        t = np.ascontiguousarray(temps, dtype=np.float32)
        p = np.ascontiguousarray(top_ps, dtype=np.float32)
        if t.shape != p.shape or t.ndim != 1:
            raise ValueError("set_row_params: temps and top_ps must be 1-D and of equal length")
        self._row = (t, p)
        self.c.row_temperature = t.ctypes.data_as(C.POINTER(C.c_float))
        self.c.row_top_p = p.ctypes.data_as(C.POINTER(C.c_float))

    def set_uniforms(self, uniforms) -> None:
        if uniforms is None:
            self.c.uniforms = None
            return
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        self._u = u
        self.c.uniforms = u.ctypes.data_as(C.POINTER(C.c_float))


class KVCache:
    """Opaque KV handle (``mi_kv``): all layers of one batch.  Duck-types the reference's
    ``PagedKVCache`` bookkeeping: ``offsets``, ``offset``, ``step``, ``reset`` (base.py:42-150);
    the tensor-level ``update_and_fetch`` is fused into the decode kernels and not exposed."""

    def __init__(self, engine: "Engine", batch_size: int, capacity: int = 256, kv_dtype: str = "float32",
                 step: int = 256):
        self.engine = engine
        self.batch_size = int(batch_size)
        self.step = int(step)
        self.kv_dtype = kv_dtype
        cap = min(max(int(capacity), 1), engine.max_positions)
        code = L.MI_KV_MODEL if kv_dtype == "model" else _DT_NAMES[kv_dtype]
        self._h = C.c_void_p()
        L.check(L.lib().mi_kv_create(engine._h, self.batch_size, cap, code, C.byref(self._h)))
        engine._kvs.add(self)

    # -- reference-compatible surface
    @property
    def offsets(self):
        out = np.zeros(self.batch_size, dtype=np.int32)
        L.check(L.lib().mi_kv_offsets(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return [int(x) for x in out]

    @property
    def offset(self) -> int:
        return max(self.offsets)

    @property
    def capacity(self) -> int:
        return int(L.lib().mi_kv_capacity(self._h))

    def reset(self, batch_size: Optional[int] = None) -> None:
        if batch_size is not None and batch_size != self.batch_size:
            raise ValueError("KVCache.reset with a different batch size: create a new cache")
        L.check(L.lib().mi_kv_reset(self._h, self.batch_size))

    def reset_row(self, row: int) -> None:
        """Forget one row (continuous batching: the slot of a finished sequence)."""
        L.check(L.lib().mi_kv_reset_row(self._h, int(row)))

    def ensure(self, needed_tokens: int) -> None:
        """Grow in ``step``-token blocks like base.py:104-117 (contents are kept)."""
        cap = self.capacity
        if needed_tokens <= cap:
            return
        new_cap = ((needed_tokens + self.step - 1) // self.step) * self.step
        new_cap = max(new_cap, min(2 * cap, self.engine.max_positions))
        new_cap = min(new_cap, self.engine.max_positions)
        if new_cap < needed_tokens:
            raise ValueError(f"context of {needed_tokens} tokens exceeds max_positions={self.engine.max_positions}")
        L.check(L.lib().mi_kv_reserve(self._h, new_cap))

    def close(self) -> None:
        if self._h:
            L.lib().mi_kv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PagedKVCache(KVCache):
    """Block-paged KV handle (``mi_kv_create_paged``): ``slots`` rows over an arena of ``n_blocks`` blocks of
    ``block_tokens`` tokens; rows grow without copies and can share the full blocks of a common prompt prefix."""

    def __init__(self, engine: "Engine", slots: int, block_tokens: int = 64, n_blocks: Optional[int] = None,
                 max_tokens_per_row: Optional[int] = None, kv_dtype: str = "model"):
        self.engine = engine
        self.batch_size = int(slots)
        self.step = int(block_tokens)
        self.kv_dtype = kv_dtype
        self.block_tokens = int(block_tokens)
        max_row = int(max_tokens_per_row or engine.max_positions)
        if n_blocks is None:                       # enough for every row at full length, + the reserved block 0
            n_blocks = self.batch_size * ((max_row + self.block_tokens - 1) // self.block_tokens) + 1
        code = L.MI_KV_MODEL if kv_dtype == "model" else _DT_NAMES[kv_dtype]
        self._h = C.c_void_p()
        L.check(L.lib().mi_kv_create_paged(engine._h, self.batch_size, self.block_tokens, int(n_blocks), max_row, code,
                                           C.byref(self._h)))
        engine._kvs.add(self)

    def ensure(self, needed_tokens: int) -> None:
        if needed_tokens > self.capacity:
            raise ValueError(f"context of {needed_tokens} tokens exceeds the row limit of this paged cache ({self.capacity})")

    def prefix_attach(self, row: int, tokens) -> int:
        """Map the published full blocks of this prompt's prefix into the (empty) row; -> tokens already in the cache."""
        t = _i32(tokens).reshape(-1)
        n = C.c_int(0)
        L.check(L.lib().mi_kv_prefix_attach(self._h, int(row), t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), C.byref(n)))
        return int(n.value)

    def prefix_publish(self, row: int, tokens) -> None:
        t = _i32(tokens).reshape(-1)
        L.check(L.lib().mi_kv_prefix_publish(self._h, int(row), t.ctypes.data_as(C.POINTER(C.c_int32)), len(t)))

    def prefix_clear(self) -> None:
        L.check(L.lib().mi_kv_prefix_clear(self._h))

    def stats(self) -> Dict[str, int]:
        out = (C.c_int64 * 7)()
        L.check(L.lib().mi_kv_stats(self._h, out, 7))
        keys = ("free_blocks", "usable_blocks", "cached_blocks", "reused_tokens", "lookup_tokens", "evictions",
                "evictable_blocks")
        return {k: int(v) for k, v in zip(keys, out)}


class Engine:
    """One model replica on one MI355X (``mi_engine``)."""

    def __init__(self, config: dict, device: int = 0, max_positions: Optional[int] = None,
                 act_dtype: str = "bfloat16"):
        mt = {"mistral": "llama"}.get(config["model_type"], config["model_type"])   # utils.py:33-36
        if mt not in ("llama", "qwen3"):
            raise ValueError(f"Model type {config['model_type']} not supported.")    # utils.py:62-65
        nh = int(config["num_attention_heads"])
        d = L.ModelDesc()
        d.struct_size = C.sizeof(L.ModelDesc)
        d.arch = L.MI_ARCH_QWEN3 if mt == "qwen3" else L.MI_ARCH_LLAMA
        d.hidden_size = int(config["hidden_size"])
        d.num_layers = int(config["num_hidden_layers"])
        d.num_heads = nh
        d.num_kv_heads = int(config.get("num_key_value_heads") or nh)
        d.head_dim = int(config.get("head_dim") or d.hidden_size // nh)
        d.intermediate_size = int(config["intermediate_size"])
        d.vocab_size = int(config["vocab_size"])
        d.rms_norm_eps = float(config["rms_norm_eps"])
        d.rope_theta = float(config.get("rope_theta", 10000.0))
        scale = 1.0
        rs = config.get("rope_scaling")
        if rs:                                                                      # llama.py:36-46,69-76
            if "factor" not in rs:
                raise ValueError("rope_scaling must contain 'factor'")
            kind = rs.get("type", rs.get("rope_type"))
            if kind == "linear":
                scale = 1.0 / float(rs["factor"])
            elif kind not in ("llama3", "default", None):
                raise ValueError(f"rope_scaling type {kind} not supported")
        d.rope_scale = scale
        tie_default = mt == "llama"           # llama.py:30 default True; mlx-lm qwen3 default False
        d.tie_word_embeddings = int(bool(config.get("tie_word_embeddings", tie_default)))
        d.act_dtype = _DT_NAMES[act_dtype]
        q = config.get("quantization")
        d.quant_bits = int(q["bits"]) if q else 0
        d.quant_group_size = int(q["group_size"]) if q else 0
        d.max_positions = int(max_positions or min(int(config.get("max_position_embeddings", 4096)), 32768))
        self.desc = d
        self.config = dict(config)
        self.model_type = mt
        self.act_dtype = act_dtype
        self.max_positions = d.max_positions
        self.device = device
        self._h = C.c_void_p()
        self._kvs = weakref.WeakSet()
        L.check(L.lib().mi_engine_create(C.byref(d), device, C.byref(self._h)))
        self._finalized = False

    # -- loading
    def set_tensor(self, name: str, tensor) -> None:
        ptr, shape, dt, on_dev, keep = _tensor_info(tensor)
        if on_dev:
            import torch

            torch.cuda.synchronize()        # the producer ran on torch's stream, the copy runs on ours
        shp = (C.c_int64 * len(shape))(*shape)
        L.check(L.lib().mi_engine_set_tensor(self._h, name.encode(), C.c_void_p(ptr), shp, len(shape), dt, on_dev))

    def load_tensors(self, items: Iterable[Tuple[str, object]], strict: bool = False) -> int:
        """Feeds (name, tensor) pairs; unknown names are skipped unless ``strict`` (the
        reference filters unmatched tensors, utils.py:693-698).  Returns the number skipped."""
        skipped = 0
        for name, t in items:
            if "rotary_emb.inv_freq" in name:                                       # llama.py:255-259
                continue
            try:
                self.set_tensor(name, t)
            except FileNotFoundError:
                if strict:
                    raise
                skipped += 1
        return skipped

    def set_lora(self, layer: int, proj: str, a, b, scale: float) -> None:
        import torch
        ta = torch.as_tensor(a).to(torch.float32).contiguous()
        tb = torch.as_tensor(b).to(torch.float32).contiguous()
        if ta.is_cuda != tb.is_cuda:
            raise ValueError("LoRA factors must live on the same device")
        rank = int(ta.shape[1])
        if ta.is_cuda:
            torch.cuda.synchronize()
        L.check(L.lib().mi_engine_set_lora(self._h, int(layer), proj.encode(), C.c_void_p(ta.data_ptr()),
                                           C.c_void_p(tb.data_ptr()), rank, float(scale), L.MI_F32, int(ta.is_cuda)))
        self.invalidate_prefix_caches()

    def invalidate_prefix_caches(self) -> None:
        """K / V computed with other weights must not be reused: forget the published prefixes of every paged cache
        of this engine (called by set_lora, i.e. by load_adapters and the LoRA hot-swap of weight_updater.py)."""
        for kv in list(self._kvs):
            if isinstance(kv, PagedKVCache) and kv._h:
                kv.prefix_clear()

    def finalize(self) -> None:
        L.check(L.lib().mi_engine_finalize(self._h))
        self._finalized = True

    # -- hot path
    def new_kv(self, batch_size: int, capacity: int = 256, kv_dtype: str = "float32", step: int = 256) -> KVCache:
        return KVCache(self, batch_size, capacity, kv_dtype, step)

    def new_paged_kv(self, slots: int, block_tokens: int = 64, n_blocks: Optional[int] = None,
                     max_tokens_per_row: Optional[int] = None, kv_dtype: str = "model") -> PagedKVCache:
        return PagedKVCache(self, slots, block_tokens, n_blocks, max_tokens_per_row, kv_dtype)

    def forward(self, tokens, kv: KVCache, all_positions: bool = False, want_logits: bool = True):
        """``model(y, cache=cache)`` (utils.py:403): tokens (B, L) -> float32 logits
        (B, V) of the last position, or (B, L, V) with ``all_positions``."""
        tok = _i32(tokens)
        B, Lq = tok.shape
        kv.ensure(max(kv.offsets) + Lq)
        out = None
        ptr = None
        if want_logits:
            V = self.desc.vocab_size
            out = np.empty((B, Lq, V) if all_positions else (B, V), dtype=np.float32)
            ptr = out.ctypes.data_as(C.POINTER(C.c_float))
        L.check(L.lib().mi_forward(self._h, kv._h, tok.ctypes.data_as(C.POINTER(C.c_int32)), B, Lq, ptr,
                                   int(all_positions)))
        return out

    def step_enqueue(self, kv: KVCache, tokens=None, sample: Optional[SampleArgs] = None, L_tokens: int = 1) -> int:
        """Launch one generate_step iteration (forward + sample) without waiting.  ``tokens``
        None feeds the tokens sampled by the previous step straight from device memory."""
        sp = sample or SampleArgs()
        ticket = C.c_int64(-1)
        if tokens is None:
            kv.ensure(max(kv.offsets) + 1)
            L.check(L.lib().mi_step_enqueue(self._h, kv._h, None, kv.batch_size, 1, C.byref(sp.c), C.byref(ticket)))
        else:
            tok = _i32(tokens)
            B, Lq = tok.shape
            kv.ensure(max(kv.offsets) + Lq)
            L.check(L.lib().mi_step_enqueue(self._h, kv._h, tok.ctypes.data_as(C.POINTER(C.c_int32)), B, Lq,
                                            C.byref(sp.c), C.byref(ticket)))
        self._last_B = kv.batch_size
        self._last_topk = sp.c.top_logprobs
        return int(ticket.value)

    def step_enqueue_rows(self, kv: KVCache, rows, tokens=None, sample: Optional[SampleArgs] = None) -> int:
        """``mi_step_enqueue_rows``: one step on the cache rows ``rows`` only (continuous batching).  ``tokens``
        (len(rows), L) or None = the tokens the previous step sampled (same rows, same order)."""
        sp = sample or SampleArgs()
        r = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1)
        ticket = C.c_int64(-1)
        offs = kv.offsets
        if tokens is None:
            tok_ptr, Lq = None, 1
        else:
            tok = _i32(tokens)
            if tok.shape[0] != len(r):
                raise ValueError("step_enqueue_rows: tokens must have one row per entry of rows")
            tok_ptr, Lq = tok.ctypes.data_as(C.POINTER(C.c_int32)), tok.shape[1]
        if len(r) == 0 or r.min() < 0 or r.max() >= kv.batch_size:
            raise ValueError("step_enqueue_rows: row out of range")
        kv.ensure(max(offs[int(i)] for i in r) + Lq)
        L.check(L.lib().mi_step_enqueue_rows(self._h, kv._h, r.ctypes.data_as(C.POINTER(C.c_int32)), len(r), tok_ptr, Lq,
                                             C.byref(sp.c), C.byref(ticket)))
        return int(ticket.value)

    def step_enqueue_mixed(self, kv: KVCache, rows, token_lists, want=None, sample: Optional[SampleArgs] = None) -> int:
        """``mi_step_enqueue_mixed``: one pass over the weights for rows that advance by different numbers of tokens --
        the live decode rows (one token each, listed first) and chunks of arriving prompts.  ``token_lists[i]`` = the
        tokens appended to cache row ``rows[i]``; ``want[i]`` (default: all) = sample after the segment's last token.
        ``step_wait(ticket, n_wanted)`` returns the wanted segments' results in order."""
        sp = sample or SampleArgs()
        r = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1)
        lens = np.ascontiguousarray([len(t) for t in token_lists], dtype=np.int32)
        if len(lens) != len(r) or len(r) == 0:
            raise ValueError("step_enqueue_mixed: one token list per row")
        w = np.ones(len(r), dtype=np.int32) if want is None else np.ascontiguousarray(want, dtype=np.int32).reshape(-1)
        toks = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in token_lists]))
        offs = kv.offsets
        kv.ensure(max(offs[int(i)] + int(n) for i, n in zip(r, lens)))
        ticket = C.c_int64(-1)
        I32 = C.POINTER(C.c_int32)
        L.check(L.lib().mi_step_enqueue_mixed(self._h, kv._h, r.ctypes.data_as(I32), lens.ctypes.data_as(I32),
                                              w.ctypes.data_as(I32), len(r), toks.ctypes.data_as(I32), C.byref(sp.c),
                                              C.byref(ticket)))
        return int(ticket.value)

    def step_wait(self, ticket: int, batch_size: int, top_logprobs: int = 0):
        B = int(batch_size)
        toks = np.empty(B, dtype=np.int32)
        lp = np.empty(B, dtype=np.float32)
        p0 = np.empty(B, dtype=np.float32)
        k = int(top_logprobs)
        tk_i = np.empty((B, max(k, 1)), dtype=np.int32)
        tk_l = np.empty((B, max(k, 1)), dtype=np.float32)
        L.check(L.lib().mi_step_wait(self._h, int(ticket), toks.ctypes.data_as(C.POINTER(C.c_int32)),
                                     lp.ctypes.data_as(C.POINTER(C.c_float)), p0.ctypes.data_as(C.POINTER(C.c_float)),
                                     tk_i.ctypes.data_as(C.POINTER(C.c_int32)), tk_l.ctypes.data_as(C.POINTER(C.c_float))))
        res = {"tokens": toks, "logprobs": lp, "probs_row0": p0}
        if k > 0:
            res["top_ids"], res["top_logprobs"] = tk_i[:, :k], tk_l[:, :k]
        return res

    def score_tokens(self, kv: KVCache, tokens, targets, sample: Optional[SampleArgs] = None):
        """Teacher-forced scoring (``mi_score_tokens``): tokens / targets (B, L) -> dict ``logprobs (B, L)``
        [+ ``top_ids`` / ``top_logprobs`` (B, L, k)]; the tokens are appended to ``kv``."""
        sp = sample or SampleArgs()
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        tgt = np.ascontiguousarray(targets, dtype=np.int32)
        if tok.ndim != 2 or tok.shape != tgt.shape:
            raise ValueError("score_tokens: tokens and targets must both be (B, L)")
        B, Lt = tok.shape
        kv.ensure(max(kv.offsets) + Lt)
        k = int(sp.c.top_logprobs)
        lp = np.empty((B, Lt), dtype=np.float32)
        ti = np.empty((B, Lt, max(k, 1)), dtype=np.int32)
        tl = np.empty((B, Lt, max(k, 1)), dtype=np.float32)
        L.check(L.lib().mi_score_tokens(self._h, kv._h, tok.ctypes.data_as(C.POINTER(C.c_int32)),
                                        tgt.ctypes.data_as(C.POINTER(C.c_int32)), B, Lt, C.byref(sp.c),
                                        lp.ctypes.data_as(C.POINTER(C.c_float)),
                                        ti.ctypes.data_as(C.POINTER(C.c_int32)), tl.ctypes.data_as(C.POINTER(C.c_float))))
        res = {"logprobs": lp}
        if k > 0:
            res["top_ids"], res["top_logprobs"] = ti[:, :, :k], tl[:, :, :k]
        return res

    def decode_sample(self, kv: KVCache, tokens, sample: Optional[SampleArgs] = None):
        sp = sample or SampleArgs()
        t = self.step_enqueue(kv, tokens, sp)
        return self.step_wait(t, kv.batch_size, sp.c.top_logprobs)

    # -- measurement / options
    def profile_select(self, name: Optional[str]) -> None:
        L.check(L.lib().mi_profile_select(self._h, (name or "").encode()))

    def profile_read(self) -> Tuple[int, float]:
        n = C.c_int64(0)
        ms = C.c_double(0.0)
        L.check(L.lib().mi_profile_read(self._h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def set_option(self, key: str, value: int) -> None:
        L.check(L.lib().mi_engine_set_option(self._h, key.encode(), int(value)))

    def sync(self) -> None:
        L.check(L.lib().mi_engine_sync(self._h))

    def close(self) -> None:
        if self._h:
            for kv in list(self._kvs):      # KV handles point into the engine: free them first
                kv.close()
            L.lib().mi_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
