"""OpenAI-compatible HTTP server with dynamic request batching over the MI355X decode engine.

Mirror of ``mlx_parallm/server/main.py``: same routes, request/response schemas,
status codes, batching rules and ``/debug/metrics`` keys; the generation itself goes through
``mlx_parallm_amd.utils`` and from there through the C ABI into the HIP kernels.

  GET  /health                       main.py:212-218
  GET  /debug/metrics                main.py:220-259
  GET  /v1/models                    main.py:261-277
  POST /v1/completions               main.py:357-455   (+ logprobs / echo path :457-625)
  POST /v1/perplexity                main.py:627-659
  POST /v1/chat/completions          main.py:749-806
  batch worker                       main.py:808-1276  (window / drain, first-request params, n expansion, dedup)
  streaming chat co-batching         main.py:1286-1401

Structure differs from the reference (module globals + three long coroutines): one ``ServerState``
object owns config, queues and metrics; the batching rules are small pure functions
(``collect_batch``, ``expand_requests``, ``dedup_prompts``, ``assemble_responses``) that the unit tests
drive without HTTP.  One engine is not re-entrant (include/mi355_decode.h), so every generation
-- batch, stream, logprobs, perplexity -- runs under ``ServerState.engine_lock``; blocking engine calls
run on a worker thread so the event loop keeps accepting requests while the GPU is busy (the
reference blocks its loop while it streams).

Deliberate differences, each a defect of the reference's path:
  * logprobs / echo / perplexity use the KV cache and the device-side scorer (``mi_score_tokens``);
    the reference re-runs the whole sequence per generated token (main.py:564-594).
  * temperature is applied once under top-p (the reference divides by T twice, main.py:576-579).
  * ``prompt_tps_*`` / ``decode_tps_*`` metrics are filled in (the reference never updates them).
"""
from __future__ import annotations

import asyncio
import json
import logging
import math
import os
import threading
import time
import uuid
from collections import OrderedDict
from contextlib import asynccontextmanager
from dataclasses import dataclass, field
from typing import Any, AsyncGenerator, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
from fastapi import FastAPI, HTTPException
from starlette.responses import StreamingResponse

from ..tokenizer_utils import TokenizerWrapper
from .schemas import (ChatCompletionChoice, ChatCompletionChunk, ChatCompletionRequest, ChatCompletionResponse,
                      ChatCompletionStreamChoice, ChatMessage, CompletionChoice, CompletionRequest,
                      CompletionResponse, CompletionUsage, DeltaMessage, InternalModelRecord, ModelList, ModelStatus,
                      PerplexityRequest, PerplexityResponse)
from .state import model_registry

log = logging.getLogger("mlx_parallm_amd.server")

AnyRequest = Union[CompletionRequest, ChatCompletionRequest]


# ------------------------------------------------------------------------------------------
# configuration
# ------------------------------------------------------------------------------------------
def _truthy(v: Optional[str]) -> Optional[bool]:
    if v is None:
        return None
    s = str(v).strip().lower()
    if s in ("1", "true", "yes", "on"):
        return True
    if s in ("0", "false", "no", "off"):
        return False
    return None


@dataclass
class ServerConfig:
    """CLI flags of the reference (cli.py:15-32) plus the engine placement."""
    model_path: Optional[str] = None
    host: str = "127.0.0.1"
    port: int = 8000
    lora_path: Optional[str] = None
    max_batch_size: int = 8
    batch_timeout: float = 0.1
    request_timeout_seconds: float = 86400.0
    max_concurrent_streams: int = 4
    scheduler: str = "default"
    diverse_mode: bool = False
    max_context_length: int = 32768
    stream_batch_timeout: float = 0.02            # main.py:85
    device: int = 0
    devices: Optional[List[int]] = None           # more than one: a model replica per GPU (continuous scheduler)
    # continuous scheduler (this build): prompt tokens per step that ride in the live rows' decode steps (0 = prefill each
    # prompt alone), blocks of the paged KV arena (None = from the free device memory), prefix-KV reuse
    chunk_tokens: int = 256
    kv_blocks: Optional[int] = None
    prefix_cache: bool = True

    @classmethod
    def from_env(cls, base: Optional["ServerConfig"] = None) -> "ServerConfig":
        """Environment overrides the reference honours when no CLI object exists (main.py:136-170)."""
        c = base or cls()
        env = os.environ
        c.model_path = c.model_path or env.get("MLX_PARALLM_MODEL") or env.get("MODEL_PATH") or env.get("MODEL")
        for key, attr, conv in (("MAX_BATCH_SIZE", "max_batch_size", int), ("BATCH_TIMEOUT", "batch_timeout", float),
                                ("REQUEST_TIMEOUT_SECONDS", "request_timeout_seconds", float),
                                ("MAX_CONCURRENT_STREAMS", "max_concurrent_streams", int)):
            if base is None and env.get(key):
                try:
                    setattr(c, attr, conv(env[key]))
                except ValueError:
                    log.warning("ignoring %s=%r", key, env[key])
        if base is None and (env.get("SCHEDULER") or env.get("MLX_PARALLM_SCHEDULER")):
            c.scheduler = str(env.get("SCHEDULER") or env.get("MLX_PARALLM_SCHEDULER"))
        div = _truthy(env.get("DIVERSE_MODE") or env.get("MLX_PARALLM_DIVERSE"))
        if div is not None:
            c.diverse_mode = div
        return c


class Metrics:
    """Counters behind ``/debug/metrics`` (main.py:52-67, 220-259)."""

    def __init__(self):
        self.batches_processed = 0
        self.batch_fill_acc = 0.0
        self.batch_fill_samples = 0
        self.queue_depth_last = 0
        self.stream_batches_processed = 0
        self.prompt_tokens_total = 0
        self.prompt_time_total = 0.0
        self.prompt_tps_last = 0.0
        self.decode_tokens_total = 0
        self.decode_time_total = 0.0
        self.decode_tps_last = 0.0
        self.batch_fill_hist = [0] * 10

    def record_batch(self, n_rows: int, max_batch: int, queue_depth: int) -> float:
        fill = (n_rows / max_batch) * 100.0 if max_batch > 0 else 0.0
        self.batches_processed += 1
        self.batch_fill_acc += fill
        self.batch_fill_samples += 1
        self.queue_depth_last = int(queue_depth)
        self.batch_fill_hist[int(min(9, max(0, fill // 10)))] += 1
        return fill

    def record_throughput(self, stats: Dict[str, float]) -> None:
        pt, ptime = stats.get("prompt_tokens", 0.0), stats.get("prompt_time", 0.0)
        dt, dtime = stats.get("decode_tokens", 0.0), stats.get("decode_time", 0.0)
        self.prompt_tokens_total += int(pt)
        self.prompt_time_total += ptime
        self.decode_tokens_total += int(dt)
        self.decode_time_total += dtime
        if ptime > 1e-9:
            self.prompt_tps_last = pt / ptime
        if dtime > 1e-9:
            self.decode_tps_last = dt / dtime

    def snapshot(self) -> Dict[str, Any]:
        return {
            "batches_processed": self.batches_processed,
            "avg_batch_fill_pct": self.batch_fill_acc / self.batch_fill_samples if self.batch_fill_samples else 0.0,
            "batch_fill_hist": list(self.batch_fill_hist),
            "queue_depth_last": self.queue_depth_last,
            "stream_batches_processed": self.stream_batches_processed,
            "prompt_tps_avg": self.prompt_tokens_total / self.prompt_time_total if self.prompt_time_total > 1e-9 else 0.0,
            "prompt_tps_last": self.prompt_tps_last,
            "decode_tps_avg": self.decode_tokens_total / self.decode_time_total if self.decode_time_total > 1e-9 else 0.0,
            "decode_tps_last": self.decode_tps_last,
            "prompt_tokens_total": self.prompt_tokens_total,
            "decode_tokens_total": self.decode_tokens_total,
        }


class QueuedRequest:
    """A non-streaming request waiting for the batch worker (main.py:95-103)."""

    def __init__(self, request_data: AnyRequest):
        self.future: asyncio.Future = asyncio.get_running_loop().create_future()
        self.request_data = request_data


class StreamQueuedChat:
    """A streaming chat request waiting to be co-batched (main.py:88-93)."""

    def __init__(self, request: ChatCompletionRequest):
        self.request = request
        self.queue: asyncio.Queue = asyncio.Queue()
        self.id = f"chatcmpl-{uuid.uuid4().hex[:28]}"


_DONE = "__DONE__"


class ServerState:
    def __init__(self, config: ServerConfig):
        self.config = config
        self.metrics = Metrics()
        self.request_queue: asyncio.Queue = asyncio.Queue()
        self.stream_queue: asyncio.Queue = asyncio.Queue()
        self.engine_lock = asyncio.Lock()
        self.stream_slots = asyncio.Semaphore(max(1, config.max_concurrent_streams))
        self.tasks: List[asyncio.Task] = []
        self.model_id: Optional[str] = None
        self.scheduler = None              # ReplicaPool of ContinuousSchedulers when config.scheduler == "continuous"
        self.extra_replicas: List[Any] = []   # models on the 2nd .. nth device (config.devices)


# ------------------------------------------------------------------------------------------
# batching rules (pure functions; unit-tested without HTTP)
# ------------------------------------------------------------------------------------------
async def collect_batch(queue: asyncio.Queue, max_batch: int, window: float) -> list:
    """One batching window (main.py:877-931): take whatever is queued right now (up to
    ``max_batch``); otherwise wait up to ``window`` for a first item; then keep admitting until the
    window -- measured from the first item -- closes or the batch is full.  Returns [] when nothing
    arrived."""
    loop = asyncio.get_running_loop()
    batch: list = []

    def drain():
        while len(batch) < max_batch:
            try:
                batch.append(queue.get_nowait())
                queue.task_done()
            except asyncio.QueueEmpty:
                return

    drain()
    if not batch:
        try:
            batch.append(await asyncio.wait_for(queue.get(), timeout=window))
            queue.task_done()
        except asyncio.TimeoutError:
            return []
    t0 = loop.time()
    while len(batch) < max_batch:
        remaining = window - (loop.time() - t0)
        if remaining <= 0:
            break
        drain()
        if len(batch) >= max_batch:
            break
        try:
            batch.append(await asyncio.wait_for(queue.get(), timeout=max(0.0, remaining)))
            queue.task_done()
        except asyncio.TimeoutError:
            break
    return batch


def first_request_params(req: AnyRequest) -> Tuple[int, float, float]:
    """(max_tokens, temp, top_p) of a batch = those of its first request, with the worker's
    fall-backs 100 / 0.7 / 1.0 (main.py:945-961)."""
    mt = req.max_tokens if req.max_tokens is not None else 100
    temp = req.temperature if req.temperature is not None else 0.7
    top_p = req.top_p if req.top_p is not None else 1.0
    return int(mt), float(temp), float(top_p)


@dataclass
class ExpandedBatch:
    prompts: List[str] = field(default_factory=list)
    owners: List[QueuedRequest] = field(default_factory=list)       # one per expanded prompt
    requested_n: "OrderedDict[int, int]" = field(default_factory=OrderedDict)   # id(QueuedRequest) -> n

    @property
    def any_n_gt1(self) -> bool:
        return any(n > 1 for n in self.requested_n.values())


def prompt_text_of(req: AnyRequest, tokenizer) -> str:
    if isinstance(req, CompletionRequest):
        return req.prompt
    if isinstance(req, ChatCompletionRequest):
        from ..utils import apply_chat_template_cached

        history = [m.model_dump(exclude_none=True) for m in req.messages]
        return apply_chat_template_cached(tokenizer, history, add_generation_prompt=True)
    raise TypeError(f"Unsupported request data type: {type(req)}")


def expand_requests(batch: Sequence[QueuedRequest], tokenizer) -> ExpandedBatch:
    """Prompt preparation and ``n`` expansion (main.py:963-1035).  A request with n > 1 becomes n
    rows whose prompts differ by 0, 1, .. n-1 trailing zero-width spaces, so that the rows are not
    collapsed by the dedup step and sample different continuations.  Requests that fail here have
    their future completed with the exception and contribute no rows."""
    out = ExpandedBatch()
    for qr in batch:
        req = qr.request_data
        n = 1
        if getattr(req, "n", None) is not None:
            if isinstance(req.n, int) and req.n > 0:
                n = req.n
            else:
                if not qr.future.done():
                    qr.future.set_exception(ValueError(f"Parameter 'n' must be a positive integer, got {req.n}"))
                continue
        try:
            text = prompt_text_of(req, tokenizer)
        except Exception as e:       # template / type errors belong to this request only
            log.error("failed to prepare prompt: %s", e, exc_info=True)
            if not qr.future.done():
                qr.future.set_exception(e)
            continue
        for j in range(n):
            out.prompts.append(text + "\u200b" * j if n > 1 else text)
            out.owners.append(qr)
        out.requested_n[id(qr)] = n
    return out


def dedup_prompts(prompts: Sequence[str]) -> Tuple[List[str], List[List[int]]]:
    """Identical prompts of one batch are generated once (main.py:1087-1113): returns the unique
    prompts in first-seen order and, per unique prompt, the positions it fans back out to."""
    unique: List[str] = []
    index: Dict[str, int] = {}
    positions: List[List[int]] = []
    for i, p in enumerate(prompts):
        if p in index:
            positions[index[p]].append(i)
        else:
            index[p] = len(unique)
            unique.append(p)
            positions.append([i])
    return unique, positions


def assemble_responses(expanded: ExpandedBatch, results: Sequence[Tuple[str, int, int]], batch_max_tokens: int,
                       model_name: str) -> None:
    """Group the rows back per request, decide ``finish_reason`` against the request's own
    ``max_tokens`` and complete the futures (main.py:1115-1262).  Usage: prompt tokens of the first
    row, completion tokens summed over the n rows."""
    now = int(time.time())
    grouped: "OrderedDict[int, Dict[str, Any]]" = OrderedDict()
    for i, qr in enumerate(expanded.owners):
        if qr.future.done():
            continue
        text, n_prompt, n_compl = results[i]
        req = qr.request_data
        limit = req.max_tokens if getattr(req, "max_tokens", None) is not None else batch_max_tokens
        finish = "length" if n_compl >= limit else "stop"
        g = grouped.setdefault(id(qr), {"qr": qr, "choices": [], "prompt_tokens": n_prompt, "completion_tokens": 0})
        if isinstance(req, CompletionRequest):
            g["choices"].append({"text": text, "logprobs": None, "finish_reason": finish})
        else:
            g["choices"].append({"message": ChatMessage(role="assistant", content=text.strip()), "finish_reason": finish})
        g["completion_tokens"] += n_compl
    for key, g in grouped.items():
        qr: QueuedRequest = g["qr"]
        want = max(1, expanded.requested_n.get(key, 1))
        raw = g["choices"]
        if raw and len(raw) < want:                # pad with the last choice / trim (main.py:1196-1203)
            raw = raw + [dict(raw[-1]) for _ in range(want - len(raw))]
        raw = raw[:want]
        usage = CompletionUsage(prompt_tokens=g["prompt_tokens"], completion_tokens=g["completion_tokens"],
                                total_tokens=g["prompt_tokens"] + g["completion_tokens"])
        rid = uuid.uuid4().hex
        if isinstance(qr.request_data, CompletionRequest):
            resp: Any = CompletionResponse(id=f"cmpl-{('req_' + rid)[:29]}", object="text_completion", created=now,
                                           model=model_name, usage=usage,
                                           choices=[CompletionChoice(index=i, **c) for i, c in enumerate(raw)])
        else:
            resp = ChatCompletionResponse(id=f"chatcmpl-{('req_' + rid)[:28]}", object="chat.completion", created=now,
                                          model=model_name, usage=usage,
                                          choices=[ChatCompletionChoice(index=i, **c) for i, c in enumerate(raw)])
        if not qr.future.done():
            qr.future.set_result(resp)


def parse_logit_bias(logit_bias: Optional[Dict[str, float]], tokenizer) -> Optional[Dict[int, float]]:
    """Keys are token ids or token strings (main.py:532-539)."""
    if not logit_bias:
        return None
    out: Dict[int, float] = {}
    for k, v in logit_bias.items():
        try:
            tid = int(k)
        except ValueError:
            tid = tokenizer._tokenizer.convert_tokens_to_ids(k)
        if tid is not None and tid >= 0:
            out[int(tid)] = out.get(int(tid), 0.0) + float(v)
    return out or None


# ------------------------------------------------------------------------------------------
# blocking engine work, run on a worker thread under ServerState.engine_lock
# ------------------------------------------------------------------------------------------
SCORE_CHUNK = int(os.getenv("ECHO_CHUNK_SIZE", "512"))     # positions per scoring call (main.py:481)


def _wrap(tokenizer) -> TokenizerWrapper:
    return tokenizer if isinstance(tokenizer, TokenizerWrapper) else TokenizerWrapper(tokenizer)


def _ids_of(tok: TokenizerWrapper, text: str) -> np.ndarray:
    enc = tok._tokenizer([text], return_tensors="np", padding=False)
    return np.asarray(enc["input_ids"], dtype=np.int32).reshape(1, -1)


def _score_prefix(model, handle, ids: np.ndarray, sample) -> Dict[str, np.ndarray]:
    """Feed ids[:, :-1] through the KV cache in chunks; logprob (and top-k) of ids[:, 1:]."""
    L = ids.shape[1]
    parts: List[Dict[str, np.ndarray]] = []
    for s in range(0, L - 1, SCORE_CHUNK):
        e = min(s + SCORE_CHUNK, L - 1)
        parts.append(model.engine.score_tokens(handle, ids[:, s:e], ids[:, s + 1:e + 1], sample))
    return {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]} if parts else {}


def completion_with_logprobs(model, tokenizer, model_id: str, request: CompletionRequest) -> CompletionResponse:
    """``logprobs`` / ``echo`` completions (main.py:457-625), single sequence.

    echo: teacher-forced logprobs of prompt tokens 1..L-1 from the device scorer; then up to
    ``max_tokens`` sampled tokens (greedy, or nucleus when 0 < top_p < 1 -- with T = 1 if the
    request's temperature is 0, as the reference does), each with its logprob and top-k under
    softmax(logits / T).  Generation stops after an EOS token, which is kept in the output."""
    from ..engine import SampleArgs
    from ..utils import _take, generate_step

    tok = _wrap(tokenizer)
    hf = tok._tokenizer
    ids = _ids_of(tok, request.prompt)
    L = ids.shape[1]
    topk = int(request.logprobs) if request.logprobs else 0
    temperature, top_p = float(request.temperature), float(request.top_p)
    bias = parse_logit_bias(request.logit_bias, tok)

    def top_dicts(top_ids, top_lps) -> List[Dict[str, float]]:
        return [{hf.convert_ids_to_tokens([int(j)])[0]: float(lp) for j, lp in zip(row_i, row_l)}
                for row_i, row_l in zip(top_ids, top_lps)]

    cache = model.make_cache(1)
    echo_tokens: List[str] = []
    echo_lps: List[float] = []
    echo_top: List[Dict[str, float]] = []
    first = ids
    if request.echo and L > 1:
        handle = model.bind_cache(cache, 1, L)
        sc = _score_prefix(model, handle, ids,
                           SampleArgs(temp=temperature, logit_bias=bias, top_logprobs=topk, logprobs_at_temperature=True))
        echo_tokens = list(hf.convert_ids_to_tokens(ids[0, 1:].tolist()))
        echo_lps = [float(x) for x in sc["logprobs"][0]]
        if topk > 0:
            echo_top = top_dicts(sc["top_ids"][0], sc["top_logprobs"][0])
        first = ids[:, -1:]                       # the prefix is in the cache; resume from the last prompt token

    gen_ids: List[int] = []
    gen_lps: List[float] = []
    gen_top: List[Dict[str, float]] = []
    if request.max_tokens > 0:
        nucleus = 0.0 < top_p < 1.0
        temp_eff = (temperature if temperature > 0 else 1.0) if nucleus else 0.0      # main.py:578-581
        steps = generate_step(first, model, temp=temp_eff, top_p=top_p if nucleus else 1.0, logit_bias=bias, cache=cache,
                              top_logprobs=topk, return_details=True, logprobs_at_temperature=temperature > 0,
                              seed=int.from_bytes(os.urandom(4), "little"))
        for _, res in _take(steps, request.max_tokens):
            t = int(res["tokens"][0])
            gen_ids.append(t)
            gen_lps.append(float(res["logprobs"][0]))
            if topk > 0:
                gen_top += top_dicts(res["top_ids"][:1], res["top_logprobs"][:1])
            if t == tok.eos_token_id:
                break
        steps.close()

    gen_text = hf.decode(gen_ids)
    text = (request.prompt + gen_text) if request.echo else gen_text
    lp_obj = None
    if topk > 0:
        gen_tokens = list(hf.convert_ids_to_tokens(gen_ids))
        tokens = (echo_tokens + gen_tokens) if request.echo else gen_tokens
        lp_obj = {"tokens": tokens,
                  "token_logprobs": (echo_lps + gen_lps) if request.echo else gen_lps,
                  "top_logprobs": (echo_top + gen_top) if request.echo else gen_top,
                  "text_offset": [0] * len(tokens)}
    usage = CompletionUsage(prompt_tokens=L, completion_tokens=len(gen_ids), total_tokens=L + len(gen_ids))
    return CompletionResponse(model=model_id, usage=usage,
                              choices=[CompletionChoice(text=text, index=0, logprobs=lp_obj, finish_reason="stop")])


def perplexity_of(model, tokenizer, model_id: str, text: str) -> PerplexityResponse:
    """main.py:627-659: mean negative log-likelihood of tokens 1..T-1 given their prefix, no chat template."""
    from ..engine import SampleArgs

    tok = _wrap(tokenizer)
    ids = _ids_of(tok, text)
    if ids.shape[1] < 2:
        return PerplexityResponse(model=model_id, token_count=0, avg_nll=0.0, ppl=1.0)
    cache = model.make_cache(1)
    handle = model.bind_cache(cache, 1, ids.shape[1])
    lps = _score_prefix(model, handle, ids, SampleArgs())["logprobs"][0]
    avg_nll = float(-np.sum(lps.astype(np.float64)) / len(lps))
    return PerplexityResponse(model=model_id, token_count=int(len(lps)), avg_nll=avg_nll, ppl=math.exp(avg_nll))


async def iterate_in_thread(make_iter: Callable[[], Any], cancel: threading.Event) -> AsyncGenerator[Any, None]:
    """Drive a blocking generator on a worker thread and surface its items on the event loop."""
    loop = asyncio.get_running_loop()
    q: asyncio.Queue = asyncio.Queue()

    def run():
        try:
            for item in make_iter():
                loop.call_soon_threadsafe(q.put_nowait, ("item", item))
                if cancel.is_set():
                    break
            loop.call_soon_threadsafe(q.put_nowait, ("done", None))
        except BaseException as e:      # noqa: BLE001 -- forwarded to the consumer
            loop.call_soon_threadsafe(q.put_nowait, ("error", e))

    fut = loop.run_in_executor(None, run)
    try:
        while True:
            kind, val = await q.get()
            if kind == "item":
                yield val
            elif kind == "error":
                raise val
            else:
                return
    finally:
        cancel.set()
        await fut


# ------------------------------------------------------------------------------------------
# workers
# ------------------------------------------------------------------------------------------
def _loaded(model_id: Optional[str]) -> Optional[InternalModelRecord]:
    rec = model_registry.get(model_id) if model_id else None
    if rec is None:
        from .state import get_active_record

        rec = get_active_record()
    if rec and rec.status == ModelStatus.LOADED and rec.model_instance is not None and rec.tokenizer_instance is not None:
        return rec
    return None


async def batch_processing_worker(state: ServerState) -> None:
    """main.py:808-1276."""
    from ..utils import batch_generate_text

    cfg = state.config
    log.info("batch worker: max_batch_size=%d batch_timeout=%.3fs", cfg.max_batch_size, cfg.batch_timeout)
    while True:
        rec = _loaded(state.model_id)
        if rec is None:
            await asyncio.sleep(0.2)
            continue
        batch: List[QueuedRequest] = []
        expanded = ExpandedBatch()
        try:
            batch = await collect_batch(state.request_queue, cfg.max_batch_size, cfg.batch_timeout)
            if not batch:
                continue
            tok = _wrap(rec.tokenizer_instance)
            max_tokens, temp, top_p = first_request_params(batch[0].request_data)
            expanded = expand_requests(batch, tok)
            if not expanded.prompts:
                for qr in batch:
                    if not qr.future.done():
                        qr.future.set_exception(RuntimeError("Request could not be processed: no prompts generated."))
                continue
            fill = state.metrics.record_batch(len(expanded.prompts), cfg.max_batch_size, state.request_queue.qsize())
            log.info("batch of %d request(s) -> %d row(s), fill %.1f%%", len(batch), len(expanded.prompts), fill)
            stats: Dict[str, float] = {}
            kw = dict(model=rec.model_instance, tokenizer=tok, max_tokens=max_tokens, temp=temp, top_p=top_p,
                      max_context_length=cfg.max_context_length, stats=stats,
                      seed=int.from_bytes(os.urandom(4), "little"))
            async with state.engine_lock:
                if cfg.diverse_mode or expanded.any_n_gt1:          # main.py:1074-1086
                    results = await batch_generate_text(prompts=expanded.prompts, disable_prefix_cache=True, **kw)
                else:
                    unique, positions = dedup_prompts(expanded.prompts)
                    uniq_results = await batch_generate_text(prompts=unique, **kw)
                    results = [None] * len(expanded.prompts)
                    for u, pos in enumerate(positions):
                        for p in pos:
                            results[p] = uniq_results[u]
            state.metrics.record_throughput(stats)
            if len(results) != len(expanded.prompts):
                raise RuntimeError("Batch generation returned unexpected number of results.")
            assemble_responses(expanded, results, max_tokens, rec.id)
        except asyncio.CancelledError:
            raise
        except Exception as e:
            log.error("batch worker: %s", e, exc_info=True)
            for qr in list(batch):
                if not qr.future.done():
                    qr.future.set_exception(e)


def _chunk_sse(chunk_id: str, model_id: str, delta: DeltaMessage, finish: Optional[str]) -> str:
    c = ChatCompletionChunk(id=chunk_id, model=model_id,
                            choices=[ChatCompletionStreamChoice(index=0, delta=delta, finish_reason=finish)])
    return f"data: {c.model_dump_json()}\n\n"


def _error_sse(message: str) -> str:
    return "data: " + json.dumps({"error": {"message": message}}) + "\n\n"


async def streaming_batch_worker(state: ServerState) -> None:
    """Co-batches streaming chat requests (main.py:1286-1401): the requests that arrive within
    ``stream_batch_timeout`` of the first one and name the same model decode as one batch; every
    step's text deltas go to the owning client's SSE queue."""
    from ..utils import apply_chat_template_cached, batch_stream_generate_text

    cfg = state.config
    while True:
        items: List[StreamQueuedChat] = []
        try:
            first: StreamQueuedChat = await state.stream_queue.get()
            model_id = first.request.model
            rec = model_registry.get(model_id)
            if not rec or rec.status != ModelStatus.LOADED or rec.model_instance is None or rec.tokenizer_instance is None:
                await first.queue.put(_error_sse("Model not ready"))
                await first.queue.put(_DONE)
                continue
            items = [first]
            loop = asyncio.get_running_loop()
            t0 = loop.time()
            while len(items) < cfg.max_batch_size:
                remaining = cfg.stream_batch_timeout - (loop.time() - t0)
                if remaining <= 0:
                    break
                try:
                    nxt: StreamQueuedChat = await asyncio.wait_for(state.stream_queue.get(), timeout=remaining)
                except asyncio.TimeoutError:
                    break
                if nxt.request.model != model_id:
                    await state.stream_queue.put(nxt)
                    break
                items.append(nxt)

            tok = _wrap(rec.tokenizer_instance)
            live: List[StreamQueuedChat] = []
            texts: List[str] = []
            for it in items:
                try:
                    msgs = [m.model_dump(exclude_none=True) for m in it.request.messages]
                    texts.append(apply_chat_template_cached(tok, msgs, add_generation_prompt=True))
                    live.append(it)
                except Exception as e:          # the failing request leaves the batch (the reference keeps an empty row)
                    await it.queue.put(_error_sse(f"Template error: {e}"))
                    await it.queue.put(_DONE)
            if not live:
                continue
            hf = tok._tokenizer
            if hf.pad_token is None:
                hf.pad_token = tok.eos_token
            side = hf.padding_side
            hf.padding_side = "left"
            try:
                prompts = np.asarray(hf(texts, return_tensors="np", padding=True)["input_ids"], dtype=np.int32)
            finally:
                hf.padding_side = side
            head = live[0].request
            max_tokens = head.max_tokens or 1024
            temperature = head.temperature if head.temperature is not None else 0.7
            top_p = head.top_p if head.top_p is not None else 1.0
            first_chunk = [True] * len(live)
            closed = [False] * len(live)
            cancel = threading.Event()
            model = rec.model_instance

            def make_iter():
                return batch_stream_generate_text(model, tok, prompts, max_tokens, temp=temperature, top_p=top_p,
                                                  seed=int.from_bytes(os.urandom(4), "little"))

            async with state.engine_lock:
                async for step in iterate_in_thread(make_iter, cancel):
                    for i, (delta_text, finish) in enumerate(step):
                        if closed[i]:
                            continue
                        d = DeltaMessage()
                        if first_chunk[i] and delta_text is not None:
                            d.role = "assistant"
                            first_chunk[i] = False
                        if delta_text is not None:
                            d.content = delta_text
                        await live[i].queue.put(_chunk_sse(live[i].id, model_id, d, finish))
                        if finish:
                            closed[i] = True
            for it in live:
                await it.queue.put(_DONE)
            state.metrics.stream_batches_processed += 1
        except asyncio.CancelledError:
            raise
        except Exception as e:
            log.error("streaming worker: %s", e, exc_info=True)
            for it in items:
                await it.queue.put(_error_sse(str(e)))
                await it.queue.put(_DONE)
            await asyncio.sleep(0.05)


# ------------------------------------------------------------------------------------------
# application
# ------------------------------------------------------------------------------------------
def _load_initial_model(state: ServerState) -> None:
    """main.py:184-216: register the CLI / environment model and load it (errors leave the record in
    ERROR_LOADING and the server up)."""
    from ..utils import load

    cfg = state.config
    if not cfg.model_path:
        log.warning("no initial model path; the registry starts empty")
        return
    rec = InternalModelRecord(id=cfg.model_path, path_or_hf_id=cfg.model_path, status=ModelStatus.LOADING,
                              model_type="causal_lm", adapter_path=cfg.lora_path)
    model_registry[rec.id] = rec
    state.model_id = rec.id
    try:
        try:
            model, tokenizer = load(cfg.model_path, adapter_path=cfg.lora_path,
                                    device=(cfg.devices[0] if cfg.devices else cfg.device))
        except TypeError:
            model, tokenizer = load(cfg.model_path, adapter_path=cfg.lora_path)
        rec.model_instance, rec.tokenizer_instance, rec.status = model, tokenizer, ModelStatus.LOADED
        log.info("loaded %s%s", rec.id, f" with adapter {cfg.lora_path}" if cfg.lora_path else "")
    except Exception as e:
        rec.status = ModelStatus.ERROR_LOADING
        log.error("failed to load %s: %s", rec.id, e, exc_info=True)


def register_model(state: ServerState, model_id: str, model, tokenizer, adapter_path: Optional[str] = None) -> None:
    """Put an already-built model into the registry (tests, embedding the server in another process)."""
    model_registry[model_id] = InternalModelRecord(id=model_id, path_or_hf_id=model_id, status=ModelStatus.LOADED,
                                                   model_type="causal_lm", adapter_path=adapter_path,
                                                   model_instance=model, tokenizer_instance=tokenizer)
    state.model_id = model_id


def create_app(config: Optional[ServerConfig] = None, *, model=None, tokenizer=None,
               model_id: Optional[str] = None) -> FastAPI:
    """Build the FastAPI application.  ``model`` / ``tokenizer`` pre-register a loaded model (used by
    the tests); otherwise ``config.model_path`` is loaded at start-up like the reference does."""
    if config is None:
        cli_args = None
        try:
            from .. import cli as _cli

            cli_args = getattr(_cli, "current_server_args", None)
        except Exception:          # pragma: no cover
            cli_args = None
        config = ServerConfig.from_env(cli_args)
    state = ServerState(config)

    @asynccontextmanager
    async def lifespan(app: FastAPI):
        state.request_queue = asyncio.Queue()
        state.stream_queue = asyncio.Queue()
        state.engine_lock = asyncio.Lock()
        state.stream_slots = asyncio.Semaphore(max(1, config.max_concurrent_streams))
        if model is not None:
            register_model(state, model_id or config.model_path or "model", model, tokenizer, config.lora_path)
        else:
            await asyncio.get_running_loop().run_in_executor(None, _load_initial_model, state)
        if config.scheduler not in ("default", "continuous"):
            log.warning("unknown scheduler %r; using 'default'", config.scheduler)
        rec = _loaded(state.model_id)
        if rec is not None and model is None and config.devices and len(config.devices) > 1:
            # one replica per further GPU (the first device already holds the registry's model); sequences are
            # spread over the replicas by the continuous scheduler, so that mode is implied
            from ..utils import load

            def more():
                return [load(config.model_path, adapter_path=config.lora_path, device=d)[0] for d in config.devices[1:]]
            state.extra_replicas = await asyncio.get_running_loop().run_in_executor(None, more)
            if config.scheduler != "continuous":
                log.info("%d devices: using the continuous scheduler", len(config.devices))
                config.scheduler = "continuous"
        if config.scheduler == "continuous" and rec is not None:
            # admit-on-step over the engine's row-subset steps (server/scheduler.py); every generation route
            # goes through it, the windowed workers are not started
            from .scheduler import ReplicaPool

            state.scheduler = ReplicaPool([rec.model_instance] + state.extra_replicas, rec.tokenizer_instance,
                                          max_slots=config.max_batch_size, metrics=state.metrics,
                                          chunk_tokens=config.chunk_tokens, kv_blocks=config.kv_blocks,
                                          prefix_cache=config.prefix_cache)
            state.scheduler.start()
            state.tasks = []
        else:
            if config.scheduler == "continuous":
                log.warning("scheduler 'continuous' needs a loaded model at start-up; using the windowed workers")
            state.tasks = [asyncio.create_task(batch_processing_worker(state)),
                           asyncio.create_task(streaming_batch_worker(state))]
        try:
            yield
        finally:
            for t in state.tasks:
                t.cancel()
            await asyncio.gather(*state.tasks, return_exceptions=True)
            if state.scheduler is not None:
                await asyncio.get_running_loop().run_in_executor(None, state.scheduler.stop)
                state.scheduler = None

    app = FastAPI(title="mlx_parallm_amd Server", version="0.1.0", lifespan=lifespan,
                  description="Batched generation server for the MI355X decode engine (mlx_parallm-compatible API).")
    app.state.server = state

    def ready_record(model_name: str, not_ready_status: int) -> InternalModelRecord:
        if model_name not in model_registry:
            raise HTTPException(status_code=404, detail=f"Model '{model_name}' not found in registry.")
        rec = model_registry[model_name]
        if rec.status != ModelStatus.LOADED or rec.model_instance is None or rec.tokenizer_instance is None:
            raise HTTPException(status_code=not_ready_status,
                                detail=f"Model '{model_name}' is not currently loaded or ready. Status: {rec.status.value}")
        return rec

    async def wait_for(qr: QueuedRequest, what: str):
        try:
            return await asyncio.wait_for(qr.future, timeout=config.request_timeout_seconds)
        except asyncio.TimeoutError:
            raise HTTPException(status_code=504, detail="Request processing timed out.")
        except HTTPException:
            raise
        except Exception as e:
            raise HTTPException(status_code=500, detail=f"Error processing {what}: {e}")

    @app.get("/health", tags=["General"])
    async def health_check():
        return {"status": "ok"}

    @app.get("/debug/metrics", tags=["Debug"])
    async def debug_metrics():
        return state.metrics.snapshot()

    @app.get("/v1/models", response_model=ModelList, tags=["Models"])
    async def list_models_endpoint():
        return ModelList(data=[rec.to_model_card() for rec in model_registry.values()])

    @app.post("/v1/completions", response_model=CompletionResponse, tags=["Generation"])
    async def create_completion(request: CompletionRequest):
        rec = ready_record(request.model, 409)
        tok = _wrap(rec.tokenizer_instance)
        n_prompt = len(tok.encode(request.prompt))                       # main.py:381-401
        available = config.max_context_length - request.max_tokens
        if n_prompt > available:
            raise HTTPException(status_code=400, detail=(
                f"Prompt too long: {n_prompt} tokens exceeds maximum context ({available} available = "
                f"{config.max_context_length} total - {request.max_tokens} max_tokens). "
                f"Please reduce prompt length or max_tokens."))
        loop = asyncio.get_running_loop()
        if (request.logprobs is not None and request.logprobs > 0) or request.echo is True:
            if request.logprobs is not None and request.logprobs > 20:
                raise HTTPException(status_code=400, detail="logprobs must be <= 20")
            async with state.engine_lock:
                try:
                    return await loop.run_in_executor(None, _exclusive(state, completion_with_logprobs), rec.model_instance,
                                                      tok, request.model, request)
                except (ValueError, NotImplementedError) as e:
                    raise HTTPException(status_code=400, detail=str(e))
        if request.stream:
            if request.n is not None and request.n > 1:
                raise HTTPException(status_code=400,
                                    detail="Streaming with n > 1 is not currently supported for completions.")
            if state.scheduler is not None:
                return StreamingResponse(_scheduled_completion_stream(state, request, tok), media_type="text/event-stream")
            return StreamingResponse(_completion_stream(state, request, rec), media_type="text/event-stream")
        if state.scheduler is not None:
            return await _scheduled_response(state, request, tok, request.model)
        qr = QueuedRequest(request)
        await state.request_queue.put(qr)
        return await wait_for(qr, "request")

    @app.post("/v1/perplexity", response_model=PerplexityResponse, tags=["Analysis"])
    async def compute_perplexity(request: PerplexityRequest):
        rec = ready_record(request.model, 409)
        async with state.engine_lock:
            return await asyncio.get_running_loop().run_in_executor(
                None, _exclusive(state, perplexity_of), rec.model_instance, rec.tokenizer_instance, request.model, request.text)

    @app.post("/v1/chat/completions", response_model=ChatCompletionResponse)
    async def create_chat_completion(request: ChatCompletionRequest):
        rec = ready_record(request.model, 500)
        if request.stream and request.n is not None and request.n > 1:
            raise HTTPException(status_code=400, detail="Streaming with n > 1 is not currently supported.")
        if state.scheduler is not None:
            tok = _wrap(rec.tokenizer_instance)
            if request.stream:
                return StreamingResponse(_scheduled_chat_stream(state, request, tok), media_type="text/event-stream")
            return await _scheduled_response(state, request, tok, request.model)
        if request.stream:
            queued = StreamQueuedChat(request)
            await state.stream_queue.put(queued)

            async def sse():
                async with state.stream_slots:
                    try:
                        while True:
                            chunk = await queued.queue.get()
                            if chunk == _DONE:
                                break
                            yield chunk
                    finally:
                        yield "data: [DONE]\n\n"

            return StreamingResponse(sse(), media_type="text/event-stream")
        qr = QueuedRequest(request)
        await state.request_queue.put(qr)
        return await wait_for(qr, "chat request")

    return app


def _exclusive(state: ServerState, fn: Callable) -> Callable:
    """Run ``fn`` with the engine to itself: under the continuous scheduler that means between two of its steps."""
    def run(*args):
        if state.scheduler is None:
            return fn(*args)
        with state.scheduler.borrow_engine():
            return fn(*args)
    return run


def _request_params(req: AnyRequest) -> Tuple[int, float, float]:
    """Per-request sampling settings under the continuous scheduler (every sequence keeps its own); the chat
    defaults are those of the streaming path (1024 tokens, main.py:709), completions carry theirs."""
    if isinstance(req, ChatCompletionRequest):
        return (int(req.max_tokens or 1024), float(req.temperature if req.temperature is not None else 0.7),
                float(req.top_p if req.top_p is not None else 1.0))
    return int(req.max_tokens), float(req.temperature), float(req.top_p)


def _submit(state: ServerState, tok: TokenizerWrapper, text: str, req: AnyRequest, on_delta: Optional[Callable] = None):
    """Queue one sequence; -> (future resolving to (text, n_prompt, n_completion, finish_reason), n_prompt, sequence).
    ``sequence.cancel()`` frees its KV slot when the caller gives up (timeout, client disconnect)."""
    loop = asyncio.get_running_loop()
    fut: asyncio.Future = loop.create_future()
    ids = _ids_of(tok, text)[0]
    max_tokens, temp, top_p = _request_params(req)
    parts: List[str] = []

    def sink(seq, delta, reason):            # scheduler thread
        if delta:
            parts.append(delta)
        if on_delta is not None and (delta or reason):
            loop.call_soon_threadsafe(on_delta, delta, reason)
        if reason is not None:
            result = ("".join(parts), len(ids), len(seq.generated), reason)
            loop.call_soon_threadsafe(lambda: fut.done() or fut.set_result(result))

    try:
        seq = state.scheduler.submit(ids, max_tokens, temp, top_p, sink)
    except ValueError as e:
        raise HTTPException(status_code=400, detail=str(e))
    return fut, len(ids), seq


async def _scheduled_response(state: ServerState, request: AnyRequest, tok: TokenizerWrapper, model_name: str):
    """Non-streaming completion / chat completion through the continuous scheduler: the n choices are n
    sequences (the zero-width-space variation of the windowed path is not needed: rows sample independently)."""
    n = request.n if request.n is not None else 1
    if not isinstance(n, int) or n <= 0:
        raise HTTPException(status_code=500, detail=f"Error processing request: Parameter 'n' must be a positive integer, got {request.n}")
    try:
        text = prompt_text_of(request, tok)
    except Exception as e:
        raise HTTPException(status_code=500, detail=f"Error processing request: {e}")
    subs = [_submit(state, tok, text, request) for _ in range(n)]
    futs = [sub[0] for sub in subs]
    try:
        results = await asyncio.wait_for(asyncio.gather(*futs), timeout=state.config.request_timeout_seconds)
    except asyncio.TimeoutError:
        for sub in subs:
            sub[2].cancel()                  # abandoned sequences must not keep their KV slots until max_tokens
        raise HTTPException(status_code=504, detail="Request processing timed out.")
    except asyncio.CancelledError:           # the client went away
        for sub in subs:
            sub[2].cancel()
        raise
    if any(r[3] == "error" for r in results):
        raise HTTPException(status_code=500, detail="Error processing request: generation failed")
    usage = CompletionUsage(prompt_tokens=results[0][1], completion_tokens=sum(r[2] for r in results),
                            total_tokens=results[0][1] + sum(r[2] for r in results))
    if isinstance(request, CompletionRequest):
        return CompletionResponse(model=model_name, usage=usage, choices=[
            CompletionChoice(text=r[0], index=i, finish_reason=r[3]) for i, r in enumerate(results)])
    return ChatCompletionResponse(model=model_name, usage=usage, choices=[
        ChatCompletionChoice(index=i, message=ChatMessage(role="assistant", content=r[0].strip()), finish_reason=r[3])
        for i, r in enumerate(results)])


async def _scheduled_events(state: ServerState, request: AnyRequest, tok: TokenizerWrapper) -> AsyncGenerator[Tuple[Optional[str], Optional[str]], None]:
    """(delta, finish_reason) events of one sequence, as they are produced."""
    q: asyncio.Queue = asyncio.Queue()
    seq = _submit(state, tok, prompt_text_of(request, tok), request, on_delta=lambda d, r: q.put_nowait((d, r)))[2]
    try:
        while True:
            delta, reason = await q.get()
            yield delta, reason
            if reason is not None:
                return
    finally:
        if seq.finished is None:             # generator closed early: the client disconnected mid-stream
            seq.cancel()


async def _scheduled_chat_stream(state: ServerState, request: ChatCompletionRequest, tok: TokenizerWrapper) -> AsyncGenerator[str, None]:
    cid = f"chatcmpl-{uuid.uuid4().hex[:28]}"
    first = True
    try:
        async with state.stream_slots:
            async for delta, reason in _scheduled_events(state, request, tok):
                d = DeltaMessage()
                if first and delta is not None:
                    d.role, first = "assistant", False
                if delta is not None:
                    d.content = delta
                yield _chunk_sse(cid, request.model, d, reason)
    except Exception as e:
        log.error("chat stream: %s", e, exc_info=True)
        yield _error_sse(str(e))
    finally:
        yield "data: [DONE]\n\n"


async def _scheduled_completion_stream(state: ServerState, request: CompletionRequest, tok: TokenizerWrapper) -> AsyncGenerator[str, None]:
    rid = f"cmpl-{uuid.uuid4().hex[:29]}"

    def sse(text: str, finish: Optional[str]) -> str:
        c = CompletionResponse(id=rid, created=int(time.time()), model=request.model, usage=None,
                               choices=[CompletionChoice(text=text, index=0, finish_reason=finish)])
        return f"data: {c.model_dump_json(exclude_none=True)}\n\n"

    try:
        async with state.stream_slots:
            async for delta, reason in _scheduled_events(state, request, tok):
                if delta:
                    yield sse(delta, None)
                if reason is not None:
                    yield sse("", reason if reason in ("stop", "length") else "stop")
    except Exception as e:
        log.error("completion stream: %s", e, exc_info=True)
        yield _error_sse(str(e))
    finally:
        yield "data: [DONE]\n\n"


async def _completion_stream(state: ServerState, request: CompletionRequest,
                             rec: InternalModelRecord) -> AsyncGenerator[str, None]:
    """SSE stream of a single completion (main.py:279-355): one ``text_completion`` object per text
    delta, a final one carrying ``finish_reason``, then ``[DONE]``."""
    from ..utils import stream_generate

    rid = f"cmpl-{uuid.uuid4().hex[:29]}"
    tok = _wrap(rec.tokenizer_instance)

    def sse(text: str, finish: Optional[str]) -> str:
        c = CompletionResponse(id=rid, created=int(time.time()), model=request.model, usage=None,
                               choices=[CompletionChoice(text=text, index=0, finish_reason=finish)])
        return f"data: {c.model_dump_json(exclude_none=True)}\n\n"

    cancel = threading.Event()
    n_deltas = 0

    def make_iter():
        return stream_generate(rec.model_instance, tok, request.prompt, max_tokens=request.max_tokens,
                               temp=request.temperature, top_p=request.top_p,
                               seed=int.from_bytes(os.urandom(4), "little"))

    try:
        async with state.stream_slots:
            async with state.engine_lock:
                async for delta in iterate_in_thread(make_iter, cancel):
                    if delta is not None:
                        n_deltas += 1
                        yield sse(delta, None)
        # stream_generate emits one segment per generated token plus the final flush
        yield sse("", "length" if n_deltas > request.max_tokens else "stop")
    except Exception as e:
        log.error("completion stream: %s", e, exc_info=True)
        yield _error_sse(str(e))
    finally:
        yield "data: [DONE]\n\n"


def _module_app() -> FastAPI:
    """``uvicorn mlx_parallm_amd.server.main:app`` (cli.py:48-55 of the reference names the module path)."""
    return create_app()


def __getattr__(name: str):          # build the default app lazily: importing this module must not load a model
    if name == "app":
        global _app
        try:
            return _app
        except NameError:
            _app = _module_app()
            return _app
    raise AttributeError(name)
