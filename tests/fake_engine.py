"""A stand-in for the device engine, backed by the oracle, so that the HOST logic of
mlx_parallm_amd.utils (generation loop, pipelining order, padding, EOS bookkeeping, KV pooling)
can be tested on a GPU-less machine.  Test infrastructure: it lives under tests/ and is never
importable from the product package."""
from __future__ import annotations

import numpy as np

from mlx_parallm_amd.models.base import BatchedKVCache, PagedKVCache, group_of, make_cache_list
from oracle import ref_generate, ref_sample


class FakeKV:
    def __init__(self, engine, ref_caches, batch_size, kv_dtype):
        self.engine, self.caches, self.batch_size, self.kv_dtype = engine, ref_caches, batch_size, kv_dtype
        self.step = 256
        self.closed = False

    @property
    def offsets(self):
        return self.caches[0].offsets

    def reset(self, batch_size=None):
        for c in self.caches:
            c.reset(batch_size)

    def ensure(self, n):
        pass

    def close(self):
        self.closed = True


class FakeSlotKV:
    """KV handle for the row-subset API (continuous batching): one oracle cache of batch 1 per row."""

    def __init__(self, engine, batch_size):
        self.engine, self.batch_size = engine, batch_size
        self.rows = [engine.ref.make_cache(1, paged=False) for _ in range(batch_size)]

    @property
    def offsets(self):
        return [c[0].offsets[0] for c in self.rows]

    def reset_row(self, row):
        self.rows[row] = self.engine.ref.make_cache(1, paged=False)
        self.engine.trace.append(("reset_row", row))

    def close(self):
        pass


class FakeEngine:
    """step_enqueue computes eagerly (there is no device) but hands results out only through
    step_wait, and records the call order so tests can check the one-step-ahead pipelining."""

    max_positions = 2048

    def __init__(self, ref_model):
        self.ref = ref_model
        self.trace = []
        self._results = {}
        self._next = 0
        self._last_tokens = None

    def new_kv(self, batch_size, capacity=256, kv_dtype="model", step=256):
        return FakeSlotKV(self, batch_size)

    def step_enqueue_rows(self, kv, rows, tokens=None, sample=None):
        rows = [int(r) for r in rows]
        if len(set(rows)) != len(rows) or min(rows) < 0 or max(rows) >= kv.batch_size:
            raise ValueError("bad rows")
        if tokens is None:
            if self._last_tokens is None or len(self._last_tokens) != len(rows):
                raise ValueError("device-resident token feed needs the row set of the previous step")
            tokens = self._last_tokens
            self.trace.append(("enqueue_rows", tuple(rows), "device-tokens"))
        else:
            tokens = np.asarray(tokens)
            self.trace.append(("enqueue_rows", tuple(rows), tokens.shape))
        c = sample.c
        temps, top_ps = getattr(sample, "_row", ([float(c.temperature)] * len(rows), [float(c.top_p)] * len(rows)))
        out = []
        for i, r in enumerate(rows):
            logits = self.ref(tokens[i:i + 1], cache=kv.rows[r])[:, -1]
            u = None
            if temps[i] != 0:
                u = np.random.default_rng(int(c.seed) * 131 + self._next * 17 + i).random(1)
            s = ref_sample.sample(logits, temp=float(temps[i]), top_p=float(top_ps[i]), uniforms=u)
            out.append(int(s["tokens"][0, 0]))
        self._last_tokens = np.asarray(out)[:, None]
        t = self._next
        self._next += 1
        self._results[t] = {"tokens": np.asarray(out, dtype=np.int32), "logprobs": np.zeros(len(rows), np.float32),
                            "probs_row0": np.zeros(len(rows), np.float32)}
        return t

    def step_enqueue_mixed(self, kv, rows, token_lists, want=None, sample=None):
        """Segments of different lengths in one step (chunked prefill next to decode rows): per row through the oracle."""
        rows = [int(r) for r in rows]
        lens = [len(t) for t in token_lists]
        if len(set(rows)) != len(rows) or sorted(lens, key=lambda n: n != 1) != lens:
            raise ValueError("bad rows / one-token segments must come first")
        want = [1] * len(rows) if want is None else [int(w) for w in want]
        self.trace.append(("enqueue_mixed", tuple(rows), tuple(lens), tuple(want)))
        c = sample.c if sample is not None else None
        nw = sum(1 for w in want if w)
        temps, top_ps = (getattr(sample, "_row", ([float(c.temperature)] * nw, [float(c.top_p)] * nw)) if c is not None else ([], []))
        out, j = [], 0
        for r, toks, w in zip(rows, token_lists, want):
            logits = self.ref(np.asarray(toks).reshape(1, -1), cache=kv.rows[r])[:, -1]
            if not w:
                continue
            u = None
            if temps[j] != 0:
                u = np.random.default_rng(int(c.seed) * 131 + self._next * 17 + j).random(1)
            out.append(int(ref_sample.sample(logits, temp=float(temps[j]), top_p=float(top_ps[j]), uniforms=u)["tokens"][0, 0]))
            j += 1
        self._last_tokens = None
        t = self._next
        self._next += 1
        self._results[t] = {"tokens": np.asarray(out, dtype=np.int32), "logprobs": np.zeros(len(out), np.float32),
                            "probs_row0": np.zeros(len(out), np.float32)}
        return t

    def forward(self, tokens, kv, all_positions=False, want_logits=True):
        lg = self.ref(np.asarray(tokens), cache=kv.caches)
        self.trace.append(("forward", np.asarray(tokens).shape))
        if not want_logits:
            return None
        return lg if all_positions else lg[:, -1]

    def step_enqueue(self, kv, tokens=None, sample=None, L_tokens=1):
        if tokens is None:
            tokens = self._last_tokens
            self.trace.append(("enqueue", "device-tokens"))
        else:
            tokens = np.asarray(tokens)
            self.trace.append(("enqueue", tokens.shape))
        logits = self.ref(tokens, cache=kv.caches)[:, -1]
        c = sample.c
        bias = None
        if c.n_logit_bias:
            bias = {int(c.logit_bias_ids[i]): float(c.logit_bias_values[i]) for i in range(c.n_logit_bias)}
        u = getattr(sample, "_u", None) if c.uniforms else None
        if c.temperature != 0 and u is None:
            u = np.random.default_rng(int(c.seed) + self._next).random(tokens.shape[0])
        s = ref_sample.sample(logits, temp=float(c.temperature), top_p=float(c.top_p), logit_bias=bias, uniforms=u,
                              logprobs_at_temperature=bool(c.logprobs_at_temperature))
        self._last_tokens = s["tokens"]
        t = self._next
        self._next += 1
        res = {"tokens": s["tokens"][:, 0].astype(np.int32), "logprobs": s["logprobs"], "probs_row0": s["probs"][:, 0]}
        k = int(c.top_logprobs)
        if k > 0:   # (value desc, id asc), like the device sampler
            lsm = s["log_softmax"]
            order = np.stack([np.lexsort((np.arange(lsm.shape[1]), -lsm[b]))[:k] for b in range(lsm.shape[0])])
            res["top_ids"] = order.astype(np.int32)
            res["top_logprobs"] = np.take_along_axis(lsm, order, axis=1).astype(np.float32)
        self._results[t] = res
        return t

    def score_tokens(self, kv, tokens, targets, sample=None):
        tokens, targets = np.asarray(tokens), np.asarray(targets)
        lg = np.array(self.ref(tokens, cache=kv.caches), dtype=np.float32)
        self.trace.append(("score", tokens.shape))
        c = sample.c
        for i in range(c.n_logit_bias):
            lg[:, :, int(c.logit_bias_ids[i])] += np.float32(c.logit_bias_values[i])
        if c.logprobs_at_temperature and c.temperature > 0:
            lg = lg * np.float32(1.0 / c.temperature)
        B, Lt, V = lg.shape
        lsm = ref_sample.log_softmax(lg.reshape(B * Lt, V)).reshape(B, Lt, V)
        tg = np.clip(targets, 0, V - 1)
        lp = np.take_along_axis(lsm, tg[..., None], axis=2)[..., 0]
        res = {"logprobs": np.where(targets < 0, 0.0, lp).astype(np.float32)}
        k = int(c.top_logprobs)
        if k > 0:
            order = np.stack([np.lexsort((np.arange(V), -row))[:k] for row in lsm.reshape(B * Lt, V)]).reshape(B, Lt, k)
            res["top_ids"] = order.astype(np.int32)
            res["top_logprobs"] = np.take_along_axis(lsm, order, axis=2).astype(np.float32)
        return res

    def step_wait(self, ticket, batch_size, top_logprobs=0):
        self.trace.append(("wait", ticket))
        return self._results.pop(ticket)


class _Layer:
    pass


class FakeModel:
    def __init__(self, model_dir, max_pos=1024):
        self.ref = ref_generate.load(model_dir, max_pos=max_pos)
        self.engine = FakeEngine(self.ref)
        cfg = self.ref.cfg
        self.layers = [_Layer() for _ in range(cfg.num_hidden_layers)]
        self.head_dim = cfg.head_dim
        self.n_kv_heads = cfg.num_key_value_heads

    def __call__(self, inputs, cache=None, last_only=False):
        tokens = np.asarray(inputs)
        if tokens.ndim == 1:
            tokens = tokens[None]
        caches = self.ref.make_cache(tokens.shape[0], paged=False) if cache is None else \
            self.bind_cache(cache, tokens.shape[0], tokens.shape[1]).caches
        lg = self.ref(tokens, cache=caches).astype(np.float32)
        return lg[:, -1:, :] if last_only else lg

    def bind_cache(self, cache, batch, new_tokens):
        g = group_of(cache)
        if g.batch_size != batch:
            raise ValueError("PagedKVCache batch size mismatch")
        if g.handle is None or g.handle.engine is not self.engine or g.handle.batch_size != batch:
            g.handle = FakeKV(self.engine, self.ref.make_cache(batch, paged=(g.kv_dtype == "float32")), batch, g.kv_dtype)
        return g.handle

    def make_cache(self, batch_size, paged=True, step=None):
        kv_heads = [self.n_kv_heads] * len(self.layers)
        return make_cache_list(PagedKVCache if paged else BatchedKVCache, self.head_dim, kv_heads, batch_size, step)
