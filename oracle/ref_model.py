"""Decoder forward + KV caches + masks, restated (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows, as text:
  * ``mlx_parallm/models/base.py:6-40``   additive causal masks
  * ``mlx_parallm/models/base.py:42-90``  BatchedKVCache (KV dtype = keys.dtype)
  * ``mlx_parallm/models/base.py:93-150`` PagedKVCache (first allocation float32, :111-112)
  * ``mlx_parallm/models/llama.py:49-253`` Attention / MLP / TransformerBlock / LlamaModel / Model
  * ``mlx_parallm/models/qwen3.py:23-209`` same + q_norm / k_norm before RoPE (:65-70)
  * LoRALinear of mlx-lm 0.24.1 as applied by ``load_adapters`` (``utils.py:742-744``)
and the MLX op semantics of SURVEY.md App. A (RMSNorm A.2, RoPE A.3, SDPA A.4, LoRA A.6).

Tensors are float32 arrays + a logical dtype string; ``round_to`` is applied exactly where
an MLX op would produce an array of that dtype.  dtype promotion follows MLX:
(T, T) -> T, (anything, float32) -> float32.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import numerics
from .numerics import matmul_nt, round_to
from .ref_quant import dequantize


CACHE_F64 = False      # speed only: keep a float64 copy of every weight matrix (make_golden_wide.py sets it)
COMPACT = False        # memory only (full-depth goldens): weights stay in their checkpoint storage (16-bit patterns / packed
                       # codes); dense float32 / float64 copies are made per call and dropped, small-row calls go through
                       # oracle/c/exact_gemm.c.  Same arithmetic as the default path (tests/test_oracle_units.py).


def promote(a: str, b: str) -> str:
    if a == b:
        return a
    return "float32"


# --------------------------------------------------------------------------- weights
@dataclass
class Linear:
    """nn.Linear / nn.QuantizedLinear (no bias on this path: llama.py:64-67, qwen3.py:37-40)."""
    dtype: str                       # dtype of weight (dense) or of scales (quantised)
    weight: Optional[np.ndarray] = None        # dense (N, K) float32 values representable in dtype
    w16: Optional[np.ndarray] = None           # COMPACT: the same matrix as uint16 bit patterns of `dtype` (weight is None)
    packed: Optional[np.ndarray] = None        # quantised (N, K*bits/32) uint32
    scales: Optional[np.ndarray] = None
    biases: Optional[np.ndarray] = None
    group_size: int = 64
    bits: int = 4
    lora_a: Optional[np.ndarray] = None        # (K, r)
    lora_b: Optional[np.ndarray] = None        # (r, N)
    lora_scale: float = 0.0
    lora_dtype: str = "float32"
    _dense_cache: Optional[np.ndarray] = field(default=None, repr=False)
    _dense64: Optional[np.ndarray] = field(default=None, repr=False)

    def dense(self) -> np.ndarray:
        if self.weight is not None:
            return self.weight
        if self.w16 is not None:
            return numerics.widen_w16(self.w16, self.dtype)              # transient
        if COMPACT:
            w = numerics.dequant_f32(self.packed, self.scales, self.biases, self.group_size, self.bits)
            return w if w is not None else dequantize(self.packed, self.scales, self.biases, self.group_size, self.bits)
        if self._dense_cache is None:
            self._dense_cache = dequantize(self.packed, self.scales, self.biases,
                                           self.group_size, self.bits)
        return self._dense_cache

    def _matmul(self, x: np.ndarray) -> np.ndarray:
        """x @ W.T, exactly summed (or under the accumulation envelope's float32 orders)."""
        if numerics.ACCUM == "exact" and COMPACT:
            if self.w16 is not None:
                return numerics.matmul_nt_w16(x, self.w16, self.dtype)
            if self.weight is None:
                y = numerics.matmul_nt_q(x, self.packed, self.scales, self.biases, self.group_size, self.bits)
                if y is not None:
                    return y
        return matmul_nt(x, self.dense64() if numerics.ACCUM == "exact" else self.dense())

    def dense64(self) -> np.ndarray:
        """float64 copy of ``dense()`` when CACHE_F64 is on (same values; saves the per-call widening of a
        production-width matrix in the decode steps of tests/golden/make_golden_wide.py)."""
        if not CACHE_F64 or COMPACT:
            return self.dense()
        if self._dense64 is None:
            self._dense64 = self.dense().astype(np.float64)
        return self._dense64

    def __call__(self, x: np.ndarray, xdt: str):
        """x @ W.T with fp32 accumulation; output dtype = result_type(x, W) (App. A.1)."""
        odt = promote(xdt, self.dtype)
        if numerics.X_SPLIT2 and xdt == "float32" and self.dtype != "float32" and x.size // x.shape[-1] > 16:
            x = numerics.split2(x, self.dtype)
        y = round_to(self._matmul(x), odt)
        if self.lora_a is not None:
            # y + (scale * ((x @ A) @ B)).astype(x.dtype)     (App. A.6)
            zdt = promote(xdt, self.lora_dtype)
            z = round_to(matmul_nt(x, self.lora_a.T), zdt)
            z = round_to(matmul_nt(z, self.lora_b.T), zdt)
            z = round_to(np.float32(self.lora_scale) * z, zdt)
            z = round_to(z, xdt)
            odt2 = promote(odt, xdt)
            y = round_to(y + z, odt2)
            odt = odt2
        return y, odt

    def rows(self, ids: np.ndarray):
        """nn.Embedding / nn.QuantizedEmbedding lookup: rows of the (de)quantised table."""
        if self.weight is not None:
            return self.weight[ids], self.dtype
        if self.w16 is not None:
            return numerics.widen_w16(self.w16[ids.reshape(-1)], self.dtype).reshape(*ids.shape, -1), self.dtype
        # dequantise only the gathered rows: round_T(scale*q + bias)
        w = dequantize(self.packed[ids.reshape(-1)], self.scales[ids.reshape(-1)],
                       self.biases[ids.reshape(-1)], self.group_size, self.bits)
        return round_to(w.reshape(*ids.shape, -1), self.dtype), self.dtype


def rms_norm(x: np.ndarray, xdt: str, w: np.ndarray, wdt: str, eps: float):
    """w * cast_T(x32 * rsqrt(mean(x32^2) + eps)); out dtype result_type(x, w)  (App. A.2;
    llama.py:175-177,205; qwen3.py:42-43,127-130,159)."""
    if numerics.ACCUM != "exact":                  # accumulation envelope (numerics.set_accum): float32 throughout
        x32 = x.astype(np.float32)
        ss = numerics.sum_last_f32(x32 * x32, numerics.ACCUM)[..., None]
        rs32 = (np.float32(1.0) / np.sqrt(ss / np.float32(x.shape[-1]) + np.float32(eps))).astype(np.float32)
        xn = round_to(x32 * rs32, xdt)
        odt = promote(xdt, wdt)
        return round_to(xn * w.astype(np.float32), odt), odt
    x64 = x.astype(np.float64)
    rs = 1.0 / np.sqrt(np.mean(x64 * x64, axis=-1, keepdims=True) + float(eps))
    xn = round_to((x64 * rs).astype(np.float32), xdt)
    odt = promote(xdt, wdt)
    return round_to(xn * w.astype(np.float32), odt), odt


def rope_tables(head_dim: int, base: float, scale: float, max_pos: int):
    """cos/sin of (pos * scale * base^(-2i/D)), i in [0, D/2) -- non-traditional RoPE (App. A.3;
    llama.py:77-82; qwen3.py:46-53 with rope_scaling=None).  Angles in float64, table float32."""
    i = np.arange(head_dim // 2, dtype=np.float64)
    inv = np.power(float(base), -2.0 * i / head_dim)
    ang = np.arange(max_pos, dtype=np.float64)[:, None] * float(scale) * inv[None, :]
    return np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)


def rope(x: np.ndarray, xdt: str, pos: np.ndarray, cos: np.ndarray, sin: np.ndarray):
    """x (B, H, L, D); pos (B, L) absolute positions.  Half-split rotation, fp32 math, cast back."""
    d2 = x.shape[-1] // 2
    c = cos[pos][:, None, :, :]          # (B,1,L,D/2)
    s = sin[pos][:, None, :, :]
    x1, x2 = x[..., :d2], x[..., d2:]
    o1 = x1 * c - x2 * s
    o2 = x1 * s + x2 * c
    return round_to(np.concatenate([o1, o2], axis=-1), xdt)


# --------------------------------------------------------------------------- KV caches
class RefBatchedKVCache:
    """base.py:42-90: buffers in the dtype of the incoming keys; uniform scalar offset."""

    paged = False

    def __init__(self, head_dim: int, n_kv_heads: int, batch_size: int = 1):
        self.n_kv_heads, self.head_dim, self.batch_size = n_kv_heads, head_dim, batch_size
        self.keys = self.values = None
        self.kdt = None
        self.offset = 0
        self.step = 256

    def reset(self, batch_size=None):                                   # base.py:53-64
        if batch_size is not None and batch_size != self.batch_size:
            self.batch_size = batch_size
            self.keys = self.values = None
        self.offset = 0

    def update_and_fetch(self, keys, values, kdt: str):                 # base.py:66-85
        prev = self.offset
        L = keys.shape[2]
        if self.keys is None or (prev + L) > self.keys.shape[2]:
            n_steps = (self.step + L - 1) // self.step
            shape = (self.batch_size, self.n_kv_heads, n_steps * self.step, self.head_dim)
            nk, nv = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
            if self.keys is not None:
                if prev % self.step != 0:
                    self.keys, self.values = self.keys[..., :prev, :], self.values[..., :prev, :]
                self.keys = np.concatenate([self.keys, nk], axis=2)
                self.values = np.concatenate([self.values, nv], axis=2)
            else:
                self.keys, self.values, self.kdt = nk, nv, kdt
        self.offset += L
        self.keys[..., prev:self.offset, :] = round_to(keys, self.kdt)
        self.values[..., prev:self.offset, :] = round_to(values, self.kdt)
        return self.keys[..., :self.offset, :], self.values[..., :self.offset, :], self.kdt

    @property
    def offsets(self):                                                   # base.py:87-90
        return [self.offset] * self.batch_size


class RefPagedKVCache(RefBatchedKVCache):
    """base.py:93-150: per-row offsets; FIRST ALLOCATION IS FLOAT32 (:111-112, quirk Q2)."""

    paged = True

    def __init__(self, head_dim, n_kv_heads, batch_size=1):
        super().__init__(head_dim, n_kv_heads, batch_size)
        self.offsets_list = [0] * batch_size

    def _ensure_capacity_for(self, needed_max: int):                     # base.py:104-117
        prev = self.keys.shape[2] if self.keys is not None else 0
        if prev >= needed_max:
            return
        n_steps = (needed_max - prev + self.step - 1) // self.step
        shape = (self.batch_size, self.n_kv_heads, n_steps * self.step, self.head_dim)
        nk, nv = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
        if self.keys is not None:
            self.keys = np.concatenate([self.keys, nk], axis=2)
            self.values = np.concatenate([self.values, nv], axis=2)
        else:
            self.keys, self.values, self.kdt = nk, nv, "float32"

    def update_and_fetch(self, keys, values, kdt: str):                  # base.py:119-140
        B, _, L, _ = keys.shape
        assert B == self.batch_size, "PagedKVCache batch size mismatch"
        new_offsets = [self.offsets_list[i] + L for i in range(B)]
        max_needed = max(new_offsets)
        self._ensure_capacity_for(max_needed)
        for i in range(B):
            s = self.offsets_list[i]
            self.keys[i, :, s:s + L, :] = keys[i]          # float32 buffer: no rounding
            self.values[i, :, s:s + L, :] = values[i]
            self.offsets_list[i] = s + L
        return self.keys[..., :max_needed, :], self.values[..., :max_needed, :], "float32"

    @property
    def offsets(self):
        return list(self.offsets_list)

    def reset(self, batch_size=None):                                    # base.py:146-149
        super().reset(batch_size)
        self.offsets_list = [0] * self.batch_size


def create_additive_causal_mask_variable(N: int, offsets, total_length: int) -> np.ndarray:
    """base.py:17-40: (B, N, total_length), -1e9 where query_pos < key_pos."""
    rinds = np.arange(total_length)
    masks = []
    for off in offsets:
        linds = np.arange(int(off), int(off) + N)
        masks.append((linds[:, None] < rinds[None]).astype(np.float32) * np.float32(-1e9))
    return np.stack(masks, axis=0)


def sdpa(q, k, v, scale: float, mask: Optional[np.ndarray], qdt: str, kdt: str):
    """softmax_fp32(q k^T scale + mask) v ; GQA by head broadcast; out dtype = promoted (App. A.4;
    llama.py:139-141; qwen3.py:111-113)."""
    B, Hq, L, D = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    odt = promote(qdt, kdt)
    if numerics.ACCUM != "exact":
        p16 = kdt if (numerics.SDPA_P16 and kdt != "float32") else None
        s2 = numerics.SDPA_SPLIT2 and (L > 1 or numerics.SDPA_SPLIT2_DECODE) and odt == "float32"
        if s2:
            q, k, v = (numerics.split2(a, "bfloat16") for a in (q, k, v))
        return round_to(_sdpa_f32(q, k, v, scale, mask, numerics.ACCUM, p16, s2), odt), odt
    k = np.repeat(k, rep, axis=1).astype(np.float64)
    v = np.repeat(v, rep, axis=1).astype(np.float64)
    s = np.matmul(q.astype(np.float64), k.transpose(0, 1, 3, 2)) * float(scale)     # (B,H,L,S), float64 sums
    if mask is not None:
        s = s + mask[:, None, :, :].astype(np.float64)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)
    o = np.matmul(p, v).astype(np.float32)
    return round_to(o, odt), odt


def _sdpa_f32(q, k, v, scale, mask, mode, p16=None, p_split2=False):
    """The same attention with float32 accumulators (accumulation envelope, numerics.set_accum): scores = chunks of 32
    head-dim products, softmax in float32 (max, exp, sum over chunks of 32 keys), P.V over chunks of 32 keys; per
    (row, kv head) to bound memory.  Output float32 values, rounded to the output dtype by the caller."""
    B, Hq, L, D = q.shape
    Hkv, S = k.shape[1], k.shape[2]
    G = Hq // Hkv
    out = np.empty((B, Hq, L, D), np.float32)
    CH = numerics.CHUNK
    for b in range(B):
        for h in range(Hkv):
            qq = q[b, h * G:(h + 1) * G].astype(np.float32)                 # (G, L, D)
            kk, vv = k[b, h].astype(np.float32), v[b, h].astype(np.float32)  # (S, D)
            a = numerics.Accum(mode)
            for d0 in range(0, D, CH):
                a.add(np.matmul(qq[..., d0:d0 + CH].astype(np.float64), kk[:, d0:d0 + CH].astype(np.float64).T).astype(np.float32))
            s = (a.result() * np.float32(scale)).astype(np.float32)        # (G, L, S)
            if mask is not None:
                s = (s + mask[b][None].astype(np.float32)).astype(np.float32)
            s = s - s.max(axis=-1, keepdims=True)
            p = np.exp(s.astype(np.float32)).astype(np.float32)
            den = numerics.sum_last_f32(p, mode)[..., None]
            if p16 is not None:                                            # numerics.SDPA_P16: 16-bit P operands, float32 sum
                p = round_to(p, p16)
            if p_split2:                                                   # numerics.SDPA_SPLIT2: P as hi + lo
                p = numerics.split2(p, "bfloat16")
            a = numerics.Accum(mode)
            for s0 in range(0, S, CH):
                a.add(np.matmul(p[..., s0:s0 + CH].astype(np.float64), vv[s0:s0 + CH].astype(np.float64)).astype(np.float32))
            out[b, h * G:(h + 1) * G] = (a.result() / den).astype(np.float32)
    return out


# --------------------------------------------------------------------------- model
@dataclass
class RefConfig:
    model_type: str                  # "llama" (mistral remaps to llama: utils.py:33-36) | "qwen3"
    hidden_size: int
    num_hidden_layers: int
    intermediate_size: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    vocab_size: int
    rms_norm_eps: float
    rope_theta: float = 10000.0
    rope_scale: float = 1.0          # 1/factor for linear scaling (llama.py:69-76)
    tie_word_embeddings: bool = True
    max_position_embeddings: int = 4096

    @staticmethod
    def from_dict(c: dict) -> "RefConfig":
        mt = {"mistral": "llama"}.get(c["model_type"], c["model_type"])
        nh = c["num_attention_heads"]
        scale = 1.0
        rs = c.get("rope_scaling")
        if rs:
            if rs.get("type") == "linear" or rs.get("rope_type") == "linear":
                scale = 1.0 / float(rs["factor"])
        # llama.ModelArgs default tie_word_embeddings=True (llama.py:30); qwen3 (mlx-lm) default False
        tie_default = True if mt == "llama" else False
        return RefConfig(
            model_type=mt,
            hidden_size=c["hidden_size"],
            num_hidden_layers=c["num_hidden_layers"],
            intermediate_size=c["intermediate_size"],
            num_attention_heads=nh,
            num_key_value_heads=c.get("num_key_value_heads") or nh,
            head_dim=c.get("head_dim") or c["hidden_size"] // nh,
            vocab_size=c["vocab_size"],
            rms_norm_eps=c["rms_norm_eps"],
            rope_theta=float(c.get("rope_theta", 10000.0)),
            rope_scale=scale,
            tie_word_embeddings=bool(c.get("tie_word_embeddings", tie_default)),
            max_position_embeddings=int(c.get("max_position_embeddings", 4096)),
        )


class RefModel:
    """``Model(args)(inputs, cache) -> logits`` (llama.py:243-253 / qwen3.py:199-209)."""

    def __init__(self, cfg: RefConfig, weights: Dict[str, object], max_pos: int = 4096):
        self.cfg = cfg
        self.w = weights                        # name -> Linear | (ndarray, dtype) for norm weights
        self.cos, self.sin = rope_tables(cfg.head_dim, cfg.rope_theta, cfg.rope_scale, max_pos)
        self.layers = list(range(cfg.num_hidden_layers))
        self.head_dim = cfg.head_dim
        self.n_kv_heads = cfg.num_key_value_heads

    def make_cache(self, batch: int, paged: bool = True):
        """utils.py:199-223 _KVPool.get: one cache object per layer."""
        klass = RefPagedKVCache if paged else RefBatchedKVCache
        return [klass(self.cfg.head_dim, self.cfg.num_key_value_heads, batch)
                for _ in range(self.cfg.num_hidden_layers)]

    # -- one attention block (llama.py:84-146 / qwen3.py:55-115)
    def _attention(self, i, x, xdt, mask, cache):
        cfg, p = self.cfg, f"model.layers.{i}.self_attn."
        B, L, _ = x.shape
        q, qdt = self.w[p + "q_proj"](x, xdt)
        k, kdt = self.w[p + "k_proj"](x, xdt)
        v, vdt = self.w[p + "v_proj"](x, xdt)
        q = q.reshape(B, L, cfg.num_attention_heads, -1)
        k = k.reshape(B, L, cfg.num_key_value_heads, -1)
        v = v.reshape(B, L, cfg.num_key_value_heads, -1).transpose(0, 2, 1, 3)
        if cfg.model_type == "qwen3":                                   # qwen3.py:65-70
            qw, qwdt = self.w[p + "q_norm"]
            kw, kwdt = self.w[p + "k_norm"]
            q, qdt = rms_norm(q, qdt, qw, qwdt, cfg.rms_norm_eps)
            k, kdt = rms_norm(k, kdt, kw, kwdt, cfg.rms_norm_eps)
        q = q.transpose(0, 2, 1, 3)
        k = k.transpose(0, 2, 1, 3)
        offsets = cache.offsets if cache is not None else [0] * B        # llama.py:100-117
        pos = np.array([[int(o) + t for t in range(L)] for o in offsets], dtype=np.int64)
        q = rope(q, qdt, pos, self.cos, self.sin)
        k = rope(k, kdt, pos, self.cos, self.sin)
        if cache is not None:                                            # llama.py:125
            k, v, kdt = cache.update_and_fetch(k, v, kdt)
            vdt = kdt
        o, odt = sdpa(q, k, v, cfg.head_dim ** -0.5, mask, qdt, promote(kdt, vdt))
        o = o.transpose(0, 2, 1, 3).reshape(B, L, -1)
        return self.w[p + "o_proj"](o, odt)

    def _mlp(self, i, x, xdt):                                           # llama.py:164-165
        p = f"model.layers.{i}.mlp."
        g, gdt = self.w[p + "gate_proj"](x, xdt)
        u, udt = self.w[p + "up_proj"](x, xdt)
        # nn.silu = x * sigmoid(x), each op producing an array of g's dtype
        sig = round_to(1.0 / (1.0 + np.exp(-g.astype(np.float64))), gdt)
        s = round_to(g * sig, gdt)
        hdt = promote(gdt, udt)
        h = round_to(s * u, hdt)
        return self.w[p + "down_proj"](h, hdt)

    def hidden(self, inputs: np.ndarray, cache=None):
        """LlamaModel.__call__ (llama.py:207-231) / Qwen3Model.__call__ (qwen3.py:161-185)."""
        cfg = self.cfg
        inputs = np.asarray(inputs, dtype=np.int64)
        B, L = inputs.shape
        h, hdt = self.w["model.embed_tokens"].rows(inputs)
        h = h.astype(np.float32)
        mask = None
        if L > 1:                                                        # llama.py:214-223
            offsets = cache[0].offsets if cache is not None else [0] * B
            total = int(max(offsets)) + L
            mask = round_to(create_additive_causal_mask_variable(L, offsets, total), hdt)
        for i in range(cfg.num_hidden_layers):                            # llama.py:181-191
            c = cache[i] if cache is not None else None
            p = f"model.layers.{i}."
            nw, nwdt = self.w[p + "input_layernorm"]
            xn, xndt = rms_norm(h, hdt, nw, nwdt, cfg.rms_norm_eps)
            r, rdt = self._attention(i, xn, xndt, mask, c)
            hdt2 = promote(hdt, rdt)
            h = round_to(h + r, hdt2)
            hdt = hdt2
            nw, nwdt = self.w[p + "post_attention_layernorm"]
            xn, xndt = rms_norm(h, hdt, nw, nwdt, cfg.rms_norm_eps)
            r, rdt = self._mlp(i, xn, xndt)
            hdt2 = promote(hdt, rdt)
            h = round_to(h + r, hdt2)
            hdt = hdt2
        nw, nwdt = self.w["model.norm"]
        return rms_norm(h, hdt, nw, nwdt, cfg.rms_norm_eps)

    def __call__(self, inputs: np.ndarray, cache=None, last_only: bool = False) -> np.ndarray:
        """``last_only``: logits of the last position only, shape (B, 1, V).  The reference computes every
        position and slices (utils.py:403-404); the head is row-wise, so the values are the same -- this only
        spares the oracle a (B*L, V) matmul at production widths."""
        h, hdt = self.hidden(inputs, cache)
        if last_only:
            h = h[:, -1:, :]
        head = self.w["model.embed_tokens"] if self.cfg.tie_word_embeddings else self.w["lm_head"]
        logits, _ = head(h, hdt)                                          # llama.py:249-252
        return logits
