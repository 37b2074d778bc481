"""The reference's generation API (``mlx_parallm/utils.py``) over the MI355X decode engine.

Same callables, argument names, defaults and return shapes as the reference; arrays are NumPy
(token ids int32, probabilities float32) instead of ``mx.array``.  Each function cites the
reference lines it mirrors.  Nothing here computes a logit: the model forward, KV append,
attention and sampling run in libmi355_decode.so (``engine.py``); a missing library is an
ImportError, never a fallback.
"""
from __future__ import annotations

import asyncio
import copy
import glob
import importlib
import json
import logging
import os
import time
from collections import OrderedDict
from pathlib import Path
from typing import Any, Callable, Dict, Generator, List, Optional, Tuple, Union

import numpy as np

from .engine import SampleArgs
from .models.base import BatchedKVCache, PagedKVCache, group_of, make_cache_list
from .tokenizer_utils import TokenizerWrapper, load_tokenizer

MODEL_REMAPPING = {"mistral": "llama"}                      # utils.py:33-36

# KV dtype of caches handed out by the pool.  "model" = KV in the model dtype (bandwidth
# optimal; BatchedKVCache semantics, base.py:71-72).  "float32" reproduces the reference's
# PagedKVCache first-allocation quirk (base.py:111-112).
DEFAULT_KV_DTYPE = os.environ.get("MLX_PARALLM_AMD_KV_DTYPE", "model")


class ModelNotFoundError(Exception):                        # utils.py:41-44
    def __init__(self, message):
        self.message = message
        super().__init__(self.message)


def _get_classes(config: dict):
    """utils.py:47-67: (Model, ModelArgs) for ``config["model_type"]``."""
    model_type = config["model_type"]
    model_type = MODEL_REMAPPING.get(model_type, model_type)
    try:
        arch = importlib.import_module(f"mlx_parallm_amd.models.{model_type}")
    except ImportError:
        msg = f"Model type {model_type} not supported."
        logging.error(msg)
        raise ValueError(msg)
    return arch.Model, arch.ModelArgs


def get_model_path(path_or_hf_repo: str, revision: Optional[str] = None) -> Path:
    """utils.py:70-108.  Local directories only: there is no network here, so what the
    reference would download raises ``ModelNotFoundError`` instead."""
    model_path = Path(path_or_hf_repo)
    if not model_path.exists():
        raise ModelNotFoundError(
            f"Model not found for path or HF repo: {path_or_hf_repo}.\n"
            "Only local model directories are supported by this build (no hub download).")
    return model_path


# --------- host-side memoisation (role of utils.py:137-194: tokenisation and chat templating are re-done for
# identical texts by every request of a sampling group, so both are memoised per tokenizer) ---------
class _BoundedMemo:
    """Least-recently-used memo of at most ``limit`` entries.  ``fetch(key, make)`` returns the stored value or
    stores ``make()``; keys are hashable tuples (no string formatting on the hot path)."""

    __slots__ = ("limit", "_entries")

    def __init__(self, limit: int):
        self.limit = int(limit)
        self._entries: "OrderedDict[Any, Any]" = OrderedDict()

    def __len__(self) -> int:
        return len(self._entries)

    def peek(self, key):
        hit = self._entries.get(key)
        if hit is not None:
            self._entries.move_to_end(key)
        return hit

    def put(self, key, value) -> None:
        entries = self._entries
        entries[key] = value
        entries.move_to_end(key)
        while len(entries) > self.limit:
            entries.popitem(last=False)

    def fetch(self, key, make: Callable[[], Any]):
        hit = self.peek(key)
        if hit is None:
            hit = make()
            self.put(key, hit)
        return hit


_token_memo = _BoundedMemo(4096)
_template_memo = _BoundedMemo(2048)


def _tokenizer_tag(tokenizer) -> int:
    return id(getattr(tokenizer, "_tokenizer", tokenizer))


def encode_cached(tokenizer: TokenizerWrapper, text: str) -> List[int]:
    """``tokenizer.encode(text)``, memoised per (tokenizer, text)  (reference: utils.py:163-170)."""
    return _token_memo.fetch((_tokenizer_tag(tokenizer), text), lambda: tokenizer.encode(text))


def _conversation_key(messages: List[Dict[str, Any]]) -> Tuple:
    """Only role and content decide the rendered prompt; anything unhashable (content parts) is keyed by its
    canonical JSON."""
    def atom(v):
        return v if isinstance(v, (str, int, float, bool, type(None))) else json.dumps(v, sort_keys=True, default=str)
    return tuple((atom(m.get("role")), atom(m.get("content"))) for m in messages)


def apply_chat_template_cached(tokenizer: TokenizerWrapper, messages: List[Dict[str, Any]], *,
                               add_generation_prompt: bool = True) -> str:
    """The rendered chat prompt, memoised per (tokenizer, flag, conversation)  (reference: utils.py:178-194)."""
    key = (_tokenizer_tag(tokenizer), bool(add_generation_prompt), _conversation_key(messages))
    return _template_memo.fetch(key, lambda: tokenizer.apply_chat_template(
        messages, tokenize=False, add_generation_prompt=add_generation_prompt))


# --------- KV pool (role of utils.py:199-226) ---------
class _KVPool:
    """Hands out the per-layer cache list of a batch shape and keeps it for the next batch of that shape: the
    device buffers behind a list (one ``mi_kv``) survive, only the row lengths go back to zero.  Lists are keyed
    by everything that decides the allocation: head size, kv heads per layer, batch, cache class and KV dtype."""

    def __init__(self):
        self._pool: Dict[Tuple, List[BatchedKVCache]] = {}

    @staticmethod
    def _recycle(caches: List[BatchedKVCache], batch_size: int, step: Optional[int]) -> None:
        for layer_cache in caches:
            layer_cache.reset(batch_size)
            if step is not None:
                layer_cache.step = step

    def get(self, head_dim: int, kv_heads: List[int], batch_size: int, *, step: Optional[int] = None,
            paged: bool = True, kv_dtype: Optional[str] = None) -> List[BatchedKVCache]:
        dtype_name = (kv_dtype or DEFAULT_KV_DTYPE) if paged else "model"
        key = (int(head_dim), tuple(kv_heads), int(batch_size), bool(paged), dtype_name)
        pooled = self._pool.get(key)
        if pooled is not None:
            self._recycle(pooled, batch_size, step)
            return pooled
        fresh = make_cache_list(PagedKVCache if paged else BatchedKVCache, head_dim, kv_heads, batch_size, step)
        group_of(fresh).kv_dtype = dtype_name
        self._pool[key] = fresh
        return fresh

    def clear(self):
        for caches in self._pool.values():
            g = group_of(caches)
            if g.handle is not None:
                g.handle.close()
        self._pool.clear()


_kv_pool = _KVPool()


def _take(gen, n: int):
    """``enumerate(first n items of gen)``.  The reference writes ``zip(generate_step(...),
    range(max_tokens))``, which pulls one item more than it uses before ``range`` runs out --
    i.e. computes and waits for a step whose tokens are thrown away; this does not."""
    import itertools

    return enumerate(itertools.islice(gen, n))


# --------- generation (utils.py:315-427) ---------
def generate_step(
    prompts,
    model,
    temp: float = 0.0,
    repetition_penalty: Optional[float] = None,
    repetition_context_size: Optional[int] = 20,
    top_p: float = 1.0,
    logit_bias: Optional[Dict[int, float]] = None,
    cache: Optional[List[BatchedKVCache]] = None,
    *,
    seed: Optional[int] = None,
    uniforms_fn: Optional[Callable[[int], np.ndarray]] = None,
    top_logprobs: int = 0,
    return_details: bool = False,
    logprobs_at_temperature: bool = False,
) -> Generator[Tuple[np.ndarray, np.ndarray], None, None]:
    """A generator producing token ids from the given prompts (utils.py:315-427).

    prompts (B, L0) left-padded ids.  The first step feeds the whole prompt (prefill), each
    later step feeds the previous step's tokens straight from device memory.  Yields
    ``(tokens (B,1) int32, probs (B,1) float32)`` per step, ``probs`` being
    ``softmax(logits)[0, tokens]`` as in the reference (utils.py:363).  Like the reference
    (utils.py:420-427) step n+1 is launched before step n's tokens are read back.

    Extensions (keyword only): ``seed`` (Philox key for temp > 0; the counter is the step index of THIS generation,
    so the same seed reproduces the same tokens; the default ``None`` draws a fresh key from ``os.urandom`` per call, so
    repeated calls on one prompt give different samples as the reference's advancing ``mx.random`` state does),
    ``uniforms_fn(step) -> (B,)``
    caller-supplied noise, ``top_logprobs``, ``return_details`` (yield the dict ``tokens / logprobs /
    probs_row0 / top_ids / top_logprobs`` instead), ``logprobs_at_temperature`` (report logprobs under
    ``softmax(logits / temp)``, as the server's logprobs path does, instead of ``softmax(logits)``).
    """
    if repetition_penalty:
        raise NotImplementedError("repetition_penalty not supported.")           # utils.py:366-367
    y = np.asarray(prompts)
    if y.ndim == 1:
        y = y[None]
    y = np.ascontiguousarray(y, dtype=np.int32)
    B = y.shape[0]
    kv_heads = [model.n_kv_heads] * len(model.layers) if isinstance(model.n_kv_heads, int) else model.n_kv_heads
    if cache is None:                                                             # utils.py:390-394
        cache = _kv_pool.get(model.head_dim, kv_heads, B, paged=True)
    handle = model.bind_cache(cache, B, y.shape[1])
    engine = model.engine
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little") >> 1

    def args_for(step: int) -> SampleArgs:
        u = uniforms_fn(step) if (uniforms_fn is not None and temp != 0) else None
        return SampleArgs(temp=temp, top_p=top_p, logit_bias=logit_bias, uniforms=u, seed=seed,
                          top_logprobs=top_logprobs, logprobs_at_temperature=logprobs_at_temperature,
                          stream_position=step)

    def emit(res):
        if return_details:
            return res
        return res["tokens"].reshape(B, 1), res["probs_row0"].reshape(B, 1)

    step = 0
    ticket = engine.step_enqueue(handle, y, args_for(step))                       # y, p = _step(y)
    while True:
        step += 1
        nxt = engine.step_enqueue(handle, None, args_for(step))                   # mx.async_eval(next_y)
        res = engine.step_wait(ticket, B, top_logprobs)                           # mx.eval(y)
        yield emit(res)
        ticket = nxt


def stream_generate(model, tokenizer, prompt: str, max_tokens: int = 100, **kwargs) -> Generator[str, None, None]:
    """utils.py:429-471."""
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    prompt_tokens = np.asarray(tokenizer.encode(prompt), dtype=np.int32)[None, :]
    detokenizer = tokenizer.detokenizer
    detokenizer.reset()
    for _n, (token, _prob) in _take(generate_step(prompt_tokens, model, **kwargs), max_tokens):
        token_item = int(token[0, 0])
        if token_item == tokenizer.eos_token_id:
            break
        detokenizer.add_token(token_item)
        yield detokenizer.last_segment
    detokenizer.finalize()
    yield detokenizer.last_segment


def batch_generate(model, tokenizer, prompts: List[str], max_tokens: int = 100, verbose: bool = False,
                   format_prompts: bool = True, formatter: Optional[Callable] = None, **kwargs) -> List[str]:
    """utils.py:473-543: left-pad, run exactly ``max_tokens`` steps (no early stop), decode, cut at
    the eos / pad strings."""
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    if verbose:
        print("=" * 10)
    if format_prompts:
        prompts_fm = [[{"role": "user", "content": prompt}] for prompt in prompts]
        prompts_fm = [tokenizer.apply_chat_template(p, add_generation_prompt=True, tokenize=False) for p in prompts_fm]
    else:
        prompts_fm = prompts
    tokenizer._tokenizer.padding_side = "left"                                   # utils.py:511-514
    if tokenizer.pad_token is None:
        tokenizer._tokenizer.pad_token = tokenizer.eos_token
        tokenizer._tokenizer.pad_token_id = tokenizer.eos_token_id
    prompts_toks = np.asarray(tokenizer._tokenizer(prompts_fm, padding=True)["input_ids"], dtype=np.int32)
    tic = time.perf_counter()
    output_toks = []
    prompt_time = 0.0
    for n, (tokens, _) in _take(generate_step(prompts_toks, model, **kwargs), max_tokens):
        if n == 0:
            prompt_time = time.perf_counter() - tic
            tic = time.perf_counter()
        output_toks.append(tokens)
    output_toks = np.concatenate(output_toks, axis=1) if output_toks else np.zeros((len(prompts), 0), np.int32)
    responses = [r.split(tokenizer.eos_token)[0].split(tokenizer.pad_token)[0]
                 for r in tokenizer.batch_decode(output_toks.tolist())]
    if verbose:
        gen_time = time.perf_counter() - tic
        prompt_tps = prompts_toks.size / max(prompt_time, 1e-9)
        gen_tps = output_toks.size / max(gen_time, 1e-9)
        print(f"Prompt: {prompt_tps:.3f} tokens-per-sec")
        print(f"Generation: {gen_tps:.3f} tokens-per-sec")
        for prompt, response in zip(prompts, responses):
            print("=" * 10)
            print("Prompt:", prompt)
            print(response)
    return responses


def generate(model, tokenizer, prompt: str, max_tokens: int = 100, verbose: bool = False,
             formatter: Optional[Callable] = None, **kwargs) -> str:
    """utils.py:546-617."""
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    if verbose:
        print("=" * 10)
        print("Prompt:", prompt)
    prompt_tokens = np.asarray(encode_cached(tokenizer, prompt), dtype=np.int32)[None]
    detokenizer = tokenizer.detokenizer
    tic = time.perf_counter()
    detokenizer.reset()
    prompt_time = 0.0
    n = -1
    for n, (token, prob) in _take(generate_step(prompt_tokens, model, **kwargs), max_tokens):
        if n == 0:
            prompt_time = time.perf_counter() - tic
            tic = time.perf_counter()
        if int(token[0, 0]) == tokenizer.eos_token_id:
            break
        detokenizer.add_token(int(token[0, 0]))
        if verbose:
            if formatter:
                detokenizer.finalize()
                formatter(detokenizer.last_segment, float(prob[0, 0]))
            else:
                print(detokenizer.last_segment, end="", flush=True)
    token_count = n + 1
    detokenizer.finalize()
    if verbose:
        gen_time = time.perf_counter() - tic
        print(detokenizer.last_segment, flush=True)
        print("=" * 10)
        if token_count == 0:
            print("No tokens generated for this prompt")
            return
        print(f"Prompt: {prompt_tokens.size / max(prompt_time, 1e-9):.3f} tokens-per-sec")
        print(f"Generation: {(token_count - 1) / max(gen_time, 1e-9):.3f} tokens-per-sec")
    return detokenizer.text


# --------- loading (utils.py:620-747) ---------
def load_config(model_path: Path) -> dict:
    try:
        with open(Path(model_path) / "config.json", "r") as f:
            return json.load(f)
    except FileNotFoundError:
        logging.error(f"Config file not found in {model_path}")
        raise


def _checkpoint_dtype(weights: Dict[str, Any]) -> str:
    import torch

    for k, t in weights.items():
        if k.endswith(".scales"):
            return {torch.float32: "float32", torch.bfloat16: "bfloat16", torch.float16: "float16"}[t.dtype]
    for k, t in weights.items():
        if t.dtype in (torch.float32, torch.bfloat16, torch.float16):
            return {torch.float32: "float32", torch.bfloat16: "bfloat16", torch.float16: "float16"}[t.dtype]
    return "float32"


def load_model(model_path: Path, lazy: bool = False, model_config: dict = {}, *, device: int = 0,
               max_positions: Optional[int] = None, weights_from_rank: Optional[int] = None,
               replicate_stats: Optional[Dict[str, float]] = None):
    """utils.py:630-708: config.json -> glob ``model*.safetensors`` -> Model(args) -> sanitize ->
    quantised modules are those with a ``.scales`` tensor -> unknown tensors filtered.

    ``weights_from_rank`` (keyword only; needs an initialised ``torch.distributed`` group): only that rank reads the
    safetensors files, the other ranks receive the tensors by bucketed broadcast straight into device memory
    (``distributed.replicate_checkpoint``: RCCL over xGMI under backend "nccl") -- SURVEY 8e."""
    from safetensors.torch import load_file

    model_path = Path(model_path)
    config = load_config(model_path)
    config.update(model_config)
    reader = True
    if weights_from_rank is not None:
        import torch.distributed as dist

        reader = (not dist.is_initialized()) or dist.get_rank() == weights_from_rank
    weights: Dict[str, Any] = {}
    if reader:
        weight_files = glob.glob(str(model_path / "model*.safetensors"))
        if not weight_files:
            weight_files = glob.glob(str(model_path / "weight*.safetensors"))   # back-compat, utils.py:659-661
        if not weight_files:
            logging.error(f"No safetensors found in {model_path}")
            raise FileNotFoundError(f"No safetensors found in {model_path}")
        for wf in sorted(weight_files):
            weights.update(load_file(wf))
    if weights_from_rank is not None:
        import torch

        from .distributed import replicate_checkpoint

        dev = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        weights = replicate_checkpoint(weights if reader else None, src=weights_from_rank, device=dev,
                                       stats=replicate_stats)
    model_class, model_args_class = _get_classes(config=config)
    model_args = model_args_class.from_dict(config)
    dtype = _checkpoint_dtype(weights)
    model = model_class(model_args, config=config, device=device, dtype=dtype, max_positions=max_positions)
    if hasattr(model, "sanitize"):
        weights = model.sanitize(weights)
    skipped = model.load_weights(list(weights.items()))
    if skipped > 0:
        logging.warning(f"Filtering out {skipped} unmatched weight tensors during load.")   # utils.py:696-698
    model.finalize()
    model.eval()
    return model


def load_adapters(model, adapter_path: str):
    """mlx-lm ``load_adapters`` as the reference uses it (utils.py:742-744): read
    ``adapter_config.json`` (``fine_tune_type``, ``num_layers``, ``lora_parameters``; written by
    rl_training/lora_init.py:140-153), adapt the LAST ``num_layers`` blocks, load
    ``adapters.safetensors`` (keys ``model.layers.<i>.<proj>.lora_a/lora_b``)."""
    from safetensors.torch import load_file

    adapter_path = Path(adapter_path)
    cfg_file = adapter_path / "adapter_config.json"
    if not cfg_file.exists():
        raise FileNotFoundError(f"The adapter path does not exist: {adapter_path}")
    cfg = json.loads(cfg_file.read_text())
    if cfg.get("fine_tune_type", "lora") != "lora":
        raise NotImplementedError(f"fine_tune_type {cfg.get('fine_tune_type')} not supported (lora only)")
    lp = cfg["lora_parameters"]
    keys = lp.get("keys") or ["self_attn.q_proj", "self_attn.v_proj"]
    weights = load_file(str(adapter_path / "adapters.safetensors"))
    n_layers = len(model.layers)
    for i in range(n_layers - int(cfg["num_layers"]), n_layers):
        for key in keys:
            base = f"model.layers.{i}.{key}"
            a, b = weights.get(base + ".lora_a"), weights.get(base + ".lora_b")
            if a is None or b is None:
                continue
            model.engine.set_lora(i, key, a, b, float(lp["scale"]))
    return model


def load(path_or_hf_repo: str, tokenizer_config={}, model_config={}, adapter_path: Optional[str] = None,
         lazy: bool = False, **engine_kwargs):
    """utils.py:711-747 -> (model, tokenizer)."""
    model_path = get_model_path(path_or_hf_repo)
    model = load_model(model_path, lazy, model_config, **engine_kwargs)
    if adapter_path is not None:
        model = load_adapters(model, adapter_path)
        model.eval()
    tokenizer = load_tokenizer(model_path, tokenizer_config)
    return model, tokenizer


def load_model_and_tokenizer(path_or_hf_repo: str, *, revision: Optional[str] = None, model_config: dict = {}):
    """utils.py:111-132."""
    model_path = get_model_path(path_or_hf_repo, revision=revision)
    return load_model(model_path, lazy=False, model_config=model_config), load_tokenizer(model_path)


# --------- streaming batch API (utils.py:983-1075) ---------
def batch_stream_generate_text(model, tokenizer, prompts_tokens, max_tokens: int, **kwargs
                               ) -> Generator[List[Tuple[Optional[str], Optional[str]]], None, None]:
    """Yields, per step, one ``(text_delta, finish_reason)`` per row; finish_reason is "stop" on
    EOS, "length" at ``max_tokens`` (utils.py:1030-1075)."""
    prompts_tokens = np.asarray(prompts_tokens)
    batch_size = prompts_tokens.shape[0]
    if not isinstance(tokenizer, TokenizerWrapper):
        tokenizer = TokenizerWrapper(tokenizer)
    detokenizers = [copy.deepcopy(tokenizer.detokenizer) for _ in range(batch_size)]
    for d in detokenizers:
        d.reset()
    active = [True] * batch_size
    counts = [0] * batch_size
    eos_token_id = tokenizer.eos_token_id
    step_kwargs = {k: v for k, v in kwargs.items() if k != "repetition_penalty"}    # utils.py:1028
    for _, (ids, _p) in _take(generate_step(prompts_tokens, model, **step_kwargs), max_tokens):
        deltas: List[Tuple[Optional[str], Optional[str]]] = [(None, None)] * batch_size
        any_active = False
        for i in range(batch_size):
            if not active[i]:
                continue
            any_active = True
            token_id = int(ids[i, 0])
            counts[i] += 1
            delta, reason = None, None
            if token_id == eos_token_id:
                active[i] = False
                detokenizers[i].finalize()
                delta, reason = detokenizers[i].last_segment, "stop"
            else:
                detokenizers[i].add_token(token_id)
                delta = detokenizers[i].last_segment
            if active[i] and counts[i] >= max_tokens:
                active[i] = False
                if not reason:
                    detokenizers[i].finalize()
                    final_segment = detokenizers[i].last_segment
                    if final_segment:
                        delta = final_segment
                    reason = "length"
            deltas[i] = (delta, reason)
        yield deltas
        if not any_active:
            break


# --------- server async batch API (utils.py:1087-1346) ---------
async def batch_generate_text(model, tokenizer, prompts: List[str], max_tokens: int = 100, temp: float = 0.7,
                              top_p: float = 1.0, disable_prefix_cache: bool = False,
                              max_context_length: Optional[int] = None, *, seed: Optional[int] = None,
                              stats: Optional[Dict[str, float]] = None) -> List[Tuple[str, int, int]]:
    """-> [(text, n_prompt_tokens, n_completion_tokens)] per prompt (utils.py:1087-1346).

    Tokenise (left pad, truncate to the effective max length), optional shared-prefix prefill,
    per-row EOS / max_tokens bookkeeping, decode.  Differences from the reference, both fixes of
    defects recorded in SURVEY App. C: the common prefix is only split off when it is actually
    prefilled (D2), at least one real token is always left in the suffix, and the process-global
    prefix-KV cache (D1) is not carried over.  Runs in the default executor like the reference
    (utils.py:1345).  ``stats`` (keyword only, optional) receives ``prompt_tokens / prompt_time /
    decode_tokens / decode_time`` for the server's ``/debug/metrics`` tokens-per-second keys."""
    if not prompts:
        return []
    loop = asyncio.get_running_loop()
    tk = tokenizer._tokenizer
    if tk.pad_token_id is None:                                                   # utils.py:1118-1129
        if tk.eos_token_id is not None:
            tk.pad_token_id = tk.eos_token_id
        else:
            tk.pad_token_id = 0
    original_side = tk.padding_side
    tk.padding_side = "left"
    if max_context_length is not None:                                            # utils.py:1135-1168
        effective_max_length = max_context_length
    else:
        effective_max_length = 2048
        cand = getattr(tk, "model_max_length", None)
        try:
            if cand is not None and int(cand) > 0:
                effective_max_length = int(cand)
        except (ValueError, TypeError):
            pass
        effective_max_length = min(effective_max_length, 65536)
    try:
        batch = tk(prompts, return_tensors="np", padding="longest", truncation=True, max_length=effective_max_length)
    finally:
        tk.padding_side = original_side
    ids_np = np.asarray(batch["input_ids"]).astype(np.int64)
    mask_np = np.asarray(batch["attention_mask"])
    n_prompt = [int(np.sum(m)) for m in mask_np]

    def _synchronous_generation():
        B = ids_np.shape[0]
        eos = tokenizer.eos_token_id
        if isinstance(eos, (set, list, tuple)):
            eos = list(eos)[0] if eos else None
        seqs = [ids[mask.astype(bool)].tolist() for ids, mask in zip(ids_np, mask_np)]
        lcp = 0
        if len(seqs) > 1 and not disable_prefix_cache:                            # utils.py:1211-1221
            for pos in range(min(len(s) for s in seqs)):
                if all(s[pos] == seqs[0][pos] for s in seqs):
                    lcp += 1
                else:
                    break
            lcp = min(lcp, min(len(s) for s in seqs) - 1)                         # keep one real token
        kv_heads = [model.n_kv_heads] * len(model.layers)
        suffixes = [s[lcp:] for s in seqs]
        est = max(len(s) for s in suffixes)
        desired = max(256, min(8192, int(((est + max_tokens) // 256 + 1) * 256)))  # utils.py:1234
        caches = _kv_pool.get(model.head_dim, kv_heads, B, step=desired, paged=True)
        if lcp > 0:                                                               # utils.py:1253-1259
            handle = model.bind_cache(caches, B, lcp)
            model.engine.forward(np.tile(np.asarray(seqs[0][:lcp], dtype=np.int32), (B, 1)), handle, want_logits=False)
        pad_id = tokenizer.eos_token_id if tokenizer.eos_token_id is not None else 0
        width = max(len(s) for s in suffixes)
        suffix_batch = np.asarray([[pad_id] * (width - len(s)) + s for s in suffixes], dtype=np.int32)
        generated: List[List[int]] = [[] for _ in range(B)]
        active = [True] * B
        counts = [0] * B
        t_start = time.perf_counter()
        t_first = None
        for step_num, (ids, _) in _take(
                generate_step(suffix_batch, model, cache=caches, temp=temp, top_p=top_p, seed=seed), max_tokens):
            if t_first is None:
                t_first = time.perf_counter()
            any_active = False
            for i in range(B):
                if not active[i]:
                    continue
                any_active = True
                token_id = int(ids[i, 0])
                if token_id == eos or counts[i] >= max_tokens:                    # utils.py:1321-1326
                    active[i] = False
                else:
                    generated[i].append(token_id)
                    counts[i] += 1
            if not any_active:
                break
        if stats is not None:
            t_end = time.perf_counter()
            t_first = t_first if t_first is not None else t_end
            stats.update(prompt_tokens=float(sum(n_prompt)), prompt_time=t_first - t_start,
                         decode_tokens=float(sum(len(g) for g in generated)), decode_time=t_end - t_first)
        return [(tokenizer.decode(generated[i], skip_special_tokens=True), n_prompt[i], len(generated[i]))
                for i in range(B)]

    return await loop.run_in_executor(None, _synchronous_generation)


batch_generate_text_util = batch_generate_text                                   # utils.py:1349


# --------- on-disk formats (utils.py:759-980) live in convert.py; re-exported under the reference's names ---------
from .convert import (MAX_FILE_SIZE_GB, convert, dequantize_model, make_shards, quantize_model, save_config,  # noqa: E402,F401
                      save_weights)
