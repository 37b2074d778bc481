"""Continuous (admit-on-step) scheduler over the engine's row-subset steps.

The reference's ``--scheduler continuous`` (server/main.py:1404-1726) unifies streaming and non-streaming
requests in one batch and admits between batches by rebuilding it.  This scheduler keeps what that mode is
for -- a request does not wait for the running batch to finish -- and does it on a slot model instead:

  * the KV cache has ``max_slots`` rows; a sequence owns one row from admission to its last token;
  * admission = reset the row, prefill the prompt on that row alone (no padding: other rows are not part
    of the call, ``mi_step_enqueue_rows`` with n = 1), emit its first token;
  * every decode step runs on exactly the rows that are alive, each with its own temperature / top_p
    (``mi_sample_params.row_temperature / row_top_p``);
  * while the row set is stable the next step is enqueued before the current one is read back, sampled tokens
    stay on the device (the same one-step-ahead pipelining as generate_step, utils.py:420-427); when a sequence
    finishes or a request is waiting, the step already in flight completes, its tokens for finished rows are
    dropped, and the next step starts from explicit tokens on the new row set.

The scheduler runs on its own thread and owns the engine while it steps; other users of the same engine
(logprobs / echo / perplexity) take ``engine_mutex`` and are let in between steps.  Results are delivered
through a per-sequence ``sink(sequence, delta_text, finish_reason)`` callback invoked on the scheduler thread.
"""
from __future__ import annotations

import collections
import itertools
import logging
import os
import threading
import time
from typing import Callable, Deque, List, Optional

import numpy as np

from ..tokenizer_utils import NaiveStreamingDetokenizer, TokenizerWrapper

log = logging.getLogger("mlx_parallm_amd.scheduler")

Sink = Callable[["Sequence", Optional[str], Optional[str]], None]


class Sequence:
    _ids = itertools.count()

    def __init__(self, prompt_ids, max_tokens: int, temp: float, top_p: float, sink: Sink, detok):
        self.id = next(Sequence._ids)
        self.prompt = np.ascontiguousarray(prompt_ids, dtype=np.int32).reshape(-1)
        self.max_tokens = int(max_tokens)
        self.temp, self.top_p = float(temp), float(top_p)
        self.sink = sink
        self.detok = detok
        self.generated: List[int] = []
        self.slot: Optional[int] = None
        self.finished: Optional[str] = None
        self.last_token = -1
        self.t_submit = time.perf_counter()
        self.cancelled = False
        self.prefill_pos = 0               # prompt tokens already in the KV cache (chunked prefill)
        self.prefilled = False             # the whole prompt is in the cache and the first token has been sampled

    def cancel(self) -> None:
        """Ask the scheduler to drop this sequence (client gone / request timed out): it stops taking decode steps
        at the next step boundary, its slot is released and its sink gets finish_reason "cancelled"."""
        self.cancelled = True


class ContinuousScheduler:
    def __init__(self, model, tokenizer, max_slots: int = 8, kv_dtype: Optional[str] = None, capacity: int = 1024,
                 metrics=None, chunk_tokens: int = 256, paged: bool = True, block_tokens: int = 64,
                 kv_blocks: Optional[int] = None, prefix_cache: bool = True, step_rows: int = 256):
        from ..engine import SampleArgs      # noqa: F401  (fail early if the library is missing)
        from ..utils import DEFAULT_KV_DTYPE

        self.model = model
        self.tok = tokenizer if isinstance(tokenizer, TokenizerWrapper) else TokenizerWrapper(tokenizer)
        self.max_slots = int(max_slots)
        # KV: a block-paged arena (rows grow without copies, cost memory for what they hold and share the blocks of a
        # common prompt prefix); engines without it (the CPU test double) get the contiguous per-row slabs
        eng = model.engine
        self.paged = bool(paged) and hasattr(eng, "new_paged_kv")
        if self.paged:
            if kv_blocks is None:
                kv_blocks = self._default_blocks(eng, block_tokens, kv_dtype or DEFAULT_KV_DTYPE)
            self.kv = eng.new_paged_kv(self.max_slots, block_tokens=block_tokens, n_blocks=kv_blocks,
                                       max_tokens_per_row=eng.max_positions, kv_dtype=kv_dtype or DEFAULT_KV_DTYPE)
        else:
            self.kv = eng.new_kv(self.max_slots, capacity=capacity, kv_dtype=kv_dtype or DEFAULT_KV_DTYPE)
        self.prefix_cache = bool(prefix_cache) and self.paged
        self.prefix_hit_tokens = 0
        self.slots: List[Optional[Sequence]] = [None] * self.max_slots
        self.pending: Deque[Sequence] = collections.deque()
        self.cv = threading.Condition()
        self.engine_mutex = threading.Lock()
        self._waiters = 0
        self._step_pending = False         # a step that reads the device-resident token feed is enqueued but not read
        self._stop = False
        self._thread: Optional[threading.Thread] = None
        self._seed = int.from_bytes(os.urandom(4), "little")
        self.metrics = metrics
        self.steps = 0                     # decode steps executed
        self.prefills = 0
        self.max_rows_seen = 0
        # chunked prefill: an arriving prompt enters the cache in chunks of at most `chunk_tokens` tokens, each chunk in
        # the SAME pass over the weights as the live rows' decode step (engine.step_enqueue_mixed).  0 = the older
        # behaviour: the whole prompt is prefilled alone while the live rows wait.
        self.chunk_tokens = int(chunk_tokens) if hasattr(model.engine, "step_enqueue_mixed") else 0
        # tokens per mixed step (decode rows + chunk tokens): the chunk budget of a step is what the live rows leave of this.
        # 256: at that size the engine's K-split tile GEMM still reads every weight matrix once per step and the marginal
        # prompt token costs ~20 us (Mistral-7B bf16: 5.8 ms at 128 rows, 8.2 ms at 256); quantised models stream W once per
        # 96 / 128 rows (two slabs at 256)
        self.step_rows = int(step_rows)
        self.mixed_steps = 0               # steps that carried prompt chunks next to decode rows
        self.chunks = 0

    def _default_blocks(self, eng, block_tokens: int, kv_dtype: str) -> int:
        """Arena size: every slot at full length if that fits in 80 % of the device memory that is free now, else what
        fits (requests then wait for blocks instead of for slots: _fits)."""
        per_row = (eng.max_positions + block_tokens - 1) // block_tokens
        full = self.max_slots * per_row + 1
        try:
            import torch

            free, _total = torch.cuda.mem_get_info(eng.device)
            d = eng.desc
            esz = 4 if kv_dtype == "float32" else (4 if eng.act_dtype == "float32" else 2)
            block_bytes = 2 * d.num_layers * d.num_kv_heads * block_tokens * d.head_dim * esz
            return int(max(2 * per_row + 1, min(full, int(0.8 * free) // block_bytes)))
        except Exception:            # (no torch / no device query: the full size, and the allocation says if it is too much)
            return full

    def _fits(self, seq: "Sequence") -> bool:
        """Admission control on the block arena: the prompt and everything the sequence may generate must find blocks now
        (free ones, or published prefix blocks that NO live row maps: a live row publishes its own prompt blocks as soon as
        its prefill is done, and those can never be evicted while it runs -- `evictable_blocks`, not `cached_blocks`) --
        a step that runs out of blocks would fail every live row."""
        if not self.paged:
            return True
        st = self.kv.stats()
        bt = self.kv.block_tokens
        need = (len(seq.prompt) + seq.max_tokens + bt - 1) // bt
        growth = 0                   # blocks the live sequences may still ask for
        offs = self.kv.offsets
        for s in self.slots:
            if s is not None and not s.finished:
                end = len(s.prompt) + s.max_tokens
                growth += max(0, (end + bt - 1) // bt - (offs[s.slot] + bt - 1) // bt)
        return need + growth <= st["free_blocks"] + st["evictable_blocks"]

    # ------------------------------------------------------------------ client side (any thread)
    def submit(self, prompt_ids, max_tokens: int, temp: float, top_p: float, sink: Sink) -> Sequence:
        seq = Sequence(prompt_ids, max_tokens, temp, top_p, sink, NaiveStreamingDetokenizer(self.tok._tokenizer))
        if len(seq.prompt) == 0:
            raise ValueError("empty prompt")
        if len(seq.prompt) + seq.max_tokens > self.model.engine.max_positions:
            raise ValueError(f"prompt ({len(seq.prompt)}) + max_tokens ({seq.max_tokens}) exceeds the engine's "
                             f"max_positions ({self.model.engine.max_positions})")
        with self.cv:
            self.pending.append(seq)
            self.cv.notify_all()
        return seq

    def start(self) -> None:
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, name="mi355-scheduler", daemon=True)
            self._thread.start()

    def stop(self) -> None:
        with self.cv:
            self._stop = True
            self.cv.notify_all()
        if self._thread is not None:
            self._thread.join(timeout=30)
            self._thread = None
        try:
            self.kv.close()
        except Exception:          # pragma: no cover
            pass

    class _Borrow:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            # A borrower overwrites the engine's device-resident token feed (d_next / last_n), so it may only get
            # the engine when no enqueued step is still going to read that feed.  The scheduler publishes that in
            # _step_pending (written while it holds the mutex); once _waiters > 0 it drains before every release.
            s = self.s
            with s.cv:
                s._waiters += 1
                s.cv.notify_all()
            while True:
                with s.cv:
                    while s._step_pending and not s._stop:
                        s.cv.wait(timeout=0.05)
                s.engine_mutex.acquire()
                with s.cv:
                    clear = not s._step_pending
                if clear:
                    return self
                s.engine_mutex.release()

        def __exit__(self, *exc):
            with self.s.cv:
                self.s._waiters -= 1
            self.s.engine_mutex.release()

    def borrow_engine(self):
        """``with scheduler.borrow_engine(): ...`` -- exclusive use of the engine between two scheduler steps."""
        return ContinuousScheduler._Borrow(self)

    # ------------------------------------------------------------------ scheduler thread
    def _emit(self, seq: Sequence, delta: Optional[str], reason: Optional[str]) -> None:
        try:
            seq.sink(seq, delta, reason)
        except Exception:          # a broken consumer must not stop the batch
            log.exception("sequence %d: sink failed", seq.id)

    def _finish(self, seq: Sequence, reason: str) -> None:
        seq.detok.finalize()
        tail = seq.detok.last_segment
        seq.finished = reason
        self._emit(seq, tail if tail else None, reason)

    def _on_token(self, seq: Sequence, tok: int) -> None:
        eos = self.tok.eos_token_id
        seq.last_token = tok
        if tok == eos:
            self._finish(seq, "stop")
            return
        seq.generated.append(tok)
        seq.detok.add_token(tok)
        delta = seq.detok.last_segment
        if len(seq.generated) >= seq.max_tokens:
            seq.detok.finalize()
            tail = seq.detok.last_segment
            seq.finished = "length"
            self._emit(seq, (delta or "") + (tail or "") or None, "length")
        elif delta:
            self._emit(seq, delta, None)

    def _sample_args(self, seqs: List[Sequence]):
        from ..engine import SampleArgs

        self._seed = (self._seed + 1) & 0xFFFFFFFF
        sp = SampleArgs(temp=seqs[0].temp, top_p=seqs[0].top_p, seed=self._seed)
        sp.set_row_params([s.temp for s in seqs], [s.top_p for s in seqs])
        return sp

    def _admit(self, seq: Sequence, slot: int) -> None:
        eng = self.model.engine
        seq.slot = slot
        self.slots[slot] = seq
        if seq.max_tokens <= 0:
            self._finish(seq, "length")
            return
        self.kv.reset_row(slot)
        seq.prefill_pos = 0
        if self.prefix_cache:                     # full blocks of an earlier prompt with the same prefix are mapped, not recomputed
            seq.prefill_pos = self.kv.prefix_attach(slot, seq.prompt)
            self.prefix_hit_tokens += seq.prefill_pos
        if self.chunk_tokens > 0:                 # the prompt enters the cache chunk by chunk, inside the decode steps
            return
        t0 = time.perf_counter()
        res = eng.step_wait(eng.step_enqueue_rows(self.kv, [slot], seq.prompt[None, seq.prefill_pos:], self._sample_args([seq])), 1)
        self.prefills += 1
        seq.prefill_pos, seq.prefilled = len(seq.prompt), True
        if self.prefix_cache:
            self.kv.prefix_publish(slot, seq.prompt)
        if self.metrics is not None:
            self.metrics.record_throughput({"prompt_tokens": float(len(seq.prompt)), "prompt_time": time.perf_counter() - t0})
        self._on_token(seq, int(res["tokens"][0]))

    def _mixed_step(self, decoding: List[Sequence], prefilling: List[Sequence]) -> None:
        """One pass over the weights: every live row decodes one token AND up to `chunk_tokens` prompt tokens of the
        arriving sequences (oldest first) enter the cache.  A sequence whose prompt is complete gets its first token
        from this same step."""
        eng = self.model.engine
        rows, toks, want, who = [], [], [], []
        for s in decoding:
            rows.append(s.slot); toks.append([s.last_token]); want.append(1); who.append(s)
        budget = self.chunk_tokens
        if self.step_rows > 0:
            budget = min(budget, max(16, self.step_rows - len(decoding)))
        n_chunk_tokens = 0
        for s in prefilling:
            if budget <= 0:
                break
            n = min(len(s.prompt) - s.prefill_pos, budget)
            last = s.prefill_pos + n == len(s.prompt)
            rows.append(s.slot); toks.append(s.prompt[s.prefill_pos:s.prefill_pos + n]); want.append(1 if last else 0); who.append(s)
            budget -= n
            n_chunk_tokens += n
        # one-token segments first (the engine's decode group): a 1-token chunk is a decode-shaped segment too
        order = sorted(range(len(rows)), key=lambda i: 0 if len(toks[i]) == 1 else 1)
        rows, toks, want, who = [rows[i] for i in order], [toks[i] for i in order], [want[i] for i in order], [who[i] for i in order]
        wanted = [s for s, w in zip(who, want) if w]
        sp = self._sample_args(wanted) if wanted else None
        t0 = time.perf_counter()
        ticket = eng.step_enqueue_mixed(self.kv, rows, toks, want, sp)
        res = eng.step_wait(ticket, len(wanted))
        dt = time.perf_counter() - t0
        self.steps += 1
        self.mixed_steps += 1
        for s, tk in zip(who, toks):
            if s in prefilling:
                s.prefill_pos += len(tk)
                self.chunks += 1
                if self.prefix_cache and s.prefill_pos == len(s.prompt):
                    self.kv.prefix_publish(s.slot, s.prompt)      # (enqueued above: later readers are ordered behind it)
        for s, t in zip(wanted, res["tokens"] if wanted else []):
            if s in prefilling:
                s.prefilled = True
                self.prefills += 1
            if not s.finished:
                self._on_token(s, int(t))
        if self.metrics is not None:
            self.metrics.record_throughput({"decode_tokens": float(len(decoding)), "decode_time": dt,
                                            "prompt_tokens": float(n_chunk_tokens), "prompt_time": dt})

    def _release_finished(self) -> None:
        for i, s in enumerate(self.slots):
            if s is not None and s.cancelled and not s.finished:
                s.finished = "cancelled"
                self._emit(s, None, "cancelled")
            if s is not None and s.finished:
                self.slots[i] = None
                if self.paged:                 # the row's blocks go back to the arena now (waiting requests count on them)
                    self.kv.reset_row(i)

    def _drop_cancelled_pending(self) -> None:
        with self.cv:
            gone = [s for s in self.pending if s.cancelled]
            for s in gone:
                self.pending.remove(s)
        for s in gone:
            s.finished = "cancelled"
            self._emit(s, None, "cancelled")

    def _run(self) -> None:
        eng = self.model.engine
        inflight = None                      # (ticket, seqs) of a decode step that has been enqueued but not read
        sp_keep = []                         # SampleArgs of the steps in flight (their host arrays must stay alive)

        def drain():
            nonlocal inflight
            if inflight is not None:
                ticket, seqs = inflight
                res = eng.step_wait(ticket, len(seqs))
                for s, t in zip(seqs, res["tokens"]):
                    if not s.finished:
                        self._on_token(s, int(t))
                inflight = None
                sp_keep.clear()
            with self.cv:
                self._step_pending = False
                self.cv.notify_all()

        while True:
            with self.cv:
                # (a finished sequence still in its slot is work too: its row gives its KV blocks back below, not at the next arrival)
                while not self._stop and not self.pending and inflight is None and not any(
                        s is not None for s in self.slots) and self._waiters == 0:
                    self.cv.wait(timeout=0.5)
                if self._stop:
                    break
                if self._waiters > 0 and inflight is None:
                    self.cv.wait(timeout=0.002)          # let the borrower take the mutex
            try:
                with self.engine_mutex:
                    # ---- admissions: one prefill per new sequence, on its own row
                    if inflight is not None and any(s.cancelled and not s.finished for s in inflight[1]):
                        drain()                              # the row set changes: finish the step in flight first
                    self._release_finished()
                    self._drop_cancelled_pending()
                    while True:
                        with self.cv:
                            free = [i for i, s in enumerate(self.slots) if s is None]
                            seq = self.pending[0] if (free and self.pending) else None
                            if seq is not None and not self._fits(seq):
                                if any(s is not None and not s.finished for s in self.slots):
                                    seq = None               # wait for blocks: a live sequence will give some back
                                else:
                                    self.pending.popleft()   # nothing is running and it still does not fit: it never will
                                    seq.finished = "error"
                                    self._emit(seq, "\n\nError during generation: prompt + max_tokens exceed the KV arena", "error")
                                    continue
                            elif seq is not None:
                                self.pending.popleft()
                        if seq is None:
                            break
                        drain()
                        self._admit(seq, free[0])
                        self._release_finished()
                    active = [s for s in self.slots if s is not None and not s.finished]
                    if not active:
                        drain()
                        continue
                    self.max_rows_seen = max(self.max_rows_seen, len(active))
                    prefilling = [s for s in active if not s.prefilled]
                    if prefilling:                           # chunked prefill rides in the live rows' decode step
                        drain()
                        prefilling.sort(key=lambda q: q.id)
                        self._mixed_step([s for s in active if s.prefilled and not s.finished], prefilling)
                        with self.cv:
                            self._step_pending = False
                            self.cv.notify_all()
                        continue
                    rows = [s.slot for s in active]
                    t0 = time.perf_counter()
                    if inflight is None:
                        sp = self._sample_args(active)
                        sp_keep.append(sp)
                        inflight = (eng.step_enqueue_rows(self.kv, rows, [[s.last_token] for s in active], sp), active)
                    # one step ahead while nothing asks for a change of the row set
                    nxt = None
                    with self.cv:
                        stable = not self.pending and self._waiters == 0
                    if stable and all(len(s.generated) + 2 <= s.max_tokens for s in active):
                        sp = self._sample_args(active)
                        sp_keep.append(sp)
                        nxt = (eng.step_enqueue_rows(self.kv, rows, None, sp), active)
                    ticket, seqs = inflight
                    res = eng.step_wait(ticket, len(seqs))
                    self.steps += 1
                    for s, t in zip(seqs, res["tokens"]):
                        if not s.finished:
                            self._on_token(s, int(t))
                    inflight = nxt
                    if len(sp_keep) > 2:
                        del sp_keep[0]
                    if inflight is not None and any(s.finished for s in active):
                        drain()                              # the row set changes: finish the step in flight first
                    if self.metrics is not None:
                        self.metrics.record_throughput({"decode_tokens": float(len(active)),
                                                        "decode_time": time.perf_counter() - t0})
                    with self.cv:
                        must_yield = self._waiters > 0
                    if must_yield:
                        drain()
                    with self.cv:                            # (still inside the mutex: see _Borrow.__enter__)
                        self._step_pending = inflight is not None
                        self.cv.notify_all()
            except Exception as e:          # engine failure: fail every sequence, keep the thread alive
                log.exception("scheduler step failed")
                inflight = None
                with self.cv:
                    self._step_pending = False
                    self.cv.notify_all()
                for i, s in enumerate(self.slots):
                    if s is not None and not s.finished:
                        s.finished = "error"
                        self._emit(s, f"\n\nError during generation: {e}", "error")
                    if s is not None and self.paged:
                        try:
                            self.kv.reset_row(i)             # the row's blocks go back to the arena
                        except Exception:
                            log.exception("could not release the KV row of a failed sequence")
                    self.slots[i] = None
        # shutdown: anything still queued or running is cut
        for s in list(self.pending) + [x for x in self.slots if x is not None]:
            if not s.finished:
                s.finished = "error"
                self._emit(s, None, "error")


class ReplicaPool:
    """One ContinuousScheduler per model replica (one replica per GPU, SURVEY §8e: sequences are independent, every
    GPU holds the full weights and its own KV rows, nothing crosses GPUs during decode).  A new sequence goes to
    the replica with the fewest sequences in flight; each replica steps on its own host thread."""

    def __init__(self, models, tokenizer, max_slots: int = 8, metrics=None, **kw):
        self.replicas: List[ContinuousScheduler] = [ContinuousScheduler(m, tokenizer, max_slots=max_slots, metrics=metrics, **kw)
                                                    for m in models]
        self._lock = threading.Lock()

    def _load(self, r: ContinuousScheduler) -> int:
        return len(r.pending) + sum(1 for s in r.slots if s is not None and not s.finished)

    def submit(self, prompt_ids, max_tokens: int, temp: float, top_p: float, sink: Sink) -> Sequence:
        with self._lock:                      # (choice and enqueue together, so concurrent submits spread out)
            r = min(self.replicas, key=self._load)
            return r.submit(prompt_ids, max_tokens, temp, top_p, sink)

    def start(self) -> None:
        for r in self.replicas:
            r.start()

    def stop(self) -> None:
        for r in self.replicas:
            r.stop()

    def borrow_engine(self):
        """Exclusive use of replica 0's engine (the model object the registry exposes)."""
        return self.replicas[0].borrow_engine()

    @property
    def max_rows_seen(self) -> int:
        return max(r.max_rows_seen for r in self.replicas)
