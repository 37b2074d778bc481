"""Server host logic on CPU: batching rules as pure functions, then the HTTP routes in-process
(httpx ASGI transport) with the device engine replaced by the oracle-backed fake.  The route
tests follow the reference's tests/test_server_basic.py and tests/test_server_batching.py."""
import asyncio
import json
from contextlib import asynccontextmanager

import httpx
import numpy as np
import pytest

from fake_engine import FakeModel
from mlx_parallm_amd import cli, utils
from mlx_parallm_amd.server import main as srv
from mlx_parallm_amd.server.schemas import ChatCompletionRequest, ChatMessage, CompletionRequest
from mlx_parallm_amd.server.state import model_registry
from mlx_parallm_amd.tokenizer_utils import load_tokenizer
from oracle import ref_generate, ref_sample

MODEL_ID = "tiny-server-model"


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    from mlx_parallm_amd.tiny_model import build_tiny_model

    d = tmp_path_factory.mktemp("srv") / "tiny"
    build_tiny_model(d, seed=5, vocab_size=320, hidden_size=32, layers=2, heads=2, kv_heads=2,
                     intermediate_size=64, quantize_model=False, dtype="float32")
    return str(d)


@pytest.fixture()
def parts(tiny):
    utils._kv_pool._pool.clear()
    model_registry.clear()
    return FakeModel(tiny, max_pos=2048), load_tokenizer(tiny)


@asynccontextmanager
async def serving(parts, **cfg):
    config = srv.ServerConfig(model_path=MODEL_ID, batch_timeout=cfg.pop("batch_timeout", 0.05), **cfg)
    app = srv.create_app(config, model=parts[0], tokenizer=parts[1], model_id=MODEL_ID)
    async with app.router.lifespan_context(app):
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://server", timeout=120) as c:
            yield c, app.state.server


def run(coro):
    return asyncio.run(coro)


# ---------------------------------------------------------------- batching rules
def test_collect_batch_drains_waits_and_caps():
    async def go():
        q = asyncio.Queue()
        assert await srv.collect_batch(q, 4, 0.02) == []                       # nothing arrives: empty after the window
        for i in range(6):
            q.put_nowait(i)
        assert await srv.collect_batch(q, 4, 0.5) == [0, 1, 2, 3]               # fast path: no waiting once full
        loop = asyncio.get_running_loop()
        loop.call_later(0.03, q.put_nowait, "late")
        loop.call_later(0.50, q.put_nowait, "too late")
        t0 = loop.time()
        got = await srv.collect_batch(q, 8, 0.15)
        assert got == [4, 5, "late"] and 0.1 < loop.time() - t0 < 0.45          # window counted from the first item
    run(go())


def test_expand_dedup_and_assemble(parts):
    _, tok = parts

    async def go():
        a = srv.QueuedRequest(CompletionRequest(model="m", prompt="same", max_tokens=4))
        b = srv.QueuedRequest(CompletionRequest(model="m", prompt="same", max_tokens=2, n=3))
        c = srv.QueuedRequest(ChatCompletionRequest(model="m", messages=[ChatMessage(role="user", content="hi")]))
        bad = srv.QueuedRequest(CompletionRequest(model="m", prompt="x", n=0))
        ex = srv.expand_requests([a, b, bad, c], srv._wrap(tok))
        assert ex.prompts[:4] == ["same", "same", "same\u200b", "same\u200b\u200b"] and len(ex.prompts) == 5
        assert "hi" in ex.prompts[4] and "<|im_start|>assistant" in ex.prompts[4]
        assert ex.owners == [a, b, b, b, c] and ex.any_n_gt1
        assert isinstance(bad.future.exception(), ValueError)
        uniq, pos = srv.dedup_prompts(ex.prompts)
        assert uniq == ["same", "same\u200b", "same\u200b\u200b", ex.prompts[4]] and pos[0] == [0, 1]
        results = [("t0", 7, 4), ("t1", 7, 1), ("t2", 8, 2), ("t3", 9, 2), (" chat ", 11, 100)]
        srv.assemble_responses(ex, results, batch_max_tokens=4, model_name="m")
        ra, rb, rc = a.future.result(), b.future.result(), c.future.result()
        assert [ch.finish_reason for ch in ra.choices] == ["length"] and ra.usage.total_tokens == 11
        assert [ch.text for ch in rb.choices] == ["t1", "t2", "t3"] and [ch.index for ch in rb.choices] == [0, 1, 2]
        assert [ch.finish_reason for ch in rb.choices] == ["stop", "length", "length"]      # against b's own max_tokens
        assert rb.usage.prompt_tokens == 7 and rb.usage.completion_tokens == 5
        # chat request without max_tokens: judged against the batch's (first request's) limit
        assert rc.choices[0].message.content == "chat" and rc.choices[0].finish_reason == "length"
        assert rc.id.startswith("chatcmpl-req_") and ra.id.startswith("cmpl-req_")
        assert srv.first_request_params(c.request_data) == (100, 0.7, 1.0)
        assert srv.first_request_params(a.request_data) == (4, 0.0, 1.0)
    run(go())


def test_logit_bias_keys_and_cli_flags(parts):
    tok = srv._wrap(parts[1])
    a_id = tok._tokenizer.convert_tokens_to_ids("a")
    assert srv.parse_logit_bias({"17": 2.0, "a": -1.0}, tok) == {17: 2.0, a_id: -1.0}
    assert srv.parse_logit_bias(None, tok) is None
    c = cli.parse_args(["--model-path", "/m", "--port", "1234", "--max-batch-size", "16", "--batch-timeout", "0.2",
                        "--scheduler", "continuous", "--diverse-mode", "true", "--request-timeout-seconds", "5"])
    assert (c.model_path, c.port, c.max_batch_size, c.batch_timeout, c.scheduler, c.diverse_mode) == \
        ("/m", 1234, 16, 0.2, "continuous", True)
    assert cli.parse_args(["--model-path", "/m", "--diverse-mode", "false"]).diverse_mode is False
    assert cli.parse_args(["--model-path", "/m", "--diverse-mode"]).diverse_mode is True
    d = cli.parse_args(["--model-path", "/m"])
    assert (d.host, d.port, d.max_batch_size, d.batch_timeout, d.request_timeout_seconds, d.max_concurrent_streams,
            d.scheduler, d.diverse_mode, d.max_context_length) == ("127.0.0.1", 8000, 8, 0.1, 86400.0, 4, "default", False, 32768)


# ---------------------------------------------------------------- routes
def test_health_models_and_404(parts):
    async def go():
        async with serving(parts) as (c, _):
            assert (await c.get("/health")).json() == {"status": "ok"}
            j = (await c.get("/v1/models")).json()
            assert j["object"] == "list" and any(m["id"] == MODEL_ID and m["status"] == "loaded" for m in j["data"])
            r = await c.post("/v1/completions", json={"model": "definitely-not-loaded", "prompt": "Hello", "max_tokens": 1})
            assert r.status_code == 404
            r = await c.post("/v1/chat/completions", json={"model": "nope", "messages": [{"role": "user", "content": "x"}]})
            assert r.status_code == 404
            m = (await c.get("/debug/metrics")).json()
            assert set(m) == {"batches_processed", "avg_batch_fill_pct", "batch_fill_hist", "queue_depth_last",
                              "stream_batches_processed", "prompt_tps_avg", "prompt_tps_last", "decode_tps_avg",
                              "decode_tps_last", "prompt_tokens_total", "decode_tokens_total"}
    run(go())


def test_greedy_completion_matches_oracle_and_counts_usage(tiny, parts):
    _, tok = parts
    prompt = "Say hello in one word."
    ids = np.asarray(tok.encode(prompt))[None]
    ref = ref_generate.load(tiny, max_pos=2048)
    want = [int(t[0, 0]) for (t, _), _ in zip(ref_generate.generate_step(ids, ref, paged=False), range(8))]
    eos = tok.eos_token_id
    if eos in want:
        want = want[:want.index(eos)]

    async def go():
        async with serving(parts) as (c, _):
            r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": prompt, "max_tokens": 8, "temperature": 0.0})
            assert r.status_code == 200, r.text[:500]
            j = r.json()
            assert len(j["choices"]) == 1 and j["choices"][0]["text"] == tok.decode(want, skip_special_tokens=True)
            assert j["usage"] == {"prompt_tokens": ids.shape[1], "completion_tokens": len(want),
                                  "total_tokens": ids.shape[1] + len(want)}
            assert j["choices"][0]["finish_reason"] == ("length" if len(want) >= 8 else "stop")
            assert j["object"] == "text_completion" and j["model"] == MODEL_ID
    run(go())


def test_n_expansion_for_completions_and_chat(parts):
    async def go():
        async with serving(parts) as (c, _):
            r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "Return exactly one token.", "max_tokens": 6,
                                                      "temperature": 0.7, "top_p": 0.95, "n": 2})
            assert r.status_code == 200, r.text[:500]
            assert [ch["index"] for ch in r.json()["choices"]] == [0, 1]
            r = await c.post("/v1/chat/completions", json={"model": MODEL_ID, "max_tokens": 6, "temperature": 0.7, "top_p": 0.95, "n": 2,
                                                           "messages": [{"role": "user", "content": "Return exactly one word."}]})
            assert r.status_code == 200, r.text[:500]
            j = r.json()
            assert len(j["choices"]) == 2 and j["object"] == "chat.completion"
            assert all(ch["message"]["role"] == "assistant" and isinstance(ch["message"]["content"], str) for ch in j["choices"])
            r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "x", "max_tokens": 2, "n": 0})
            assert r.status_code == 500 and "positive integer" in r.text
    run(go())


def test_concurrent_requests_share_a_batch_and_dedup(parts):
    model, _ = parts

    async def go():
        async with serving(parts, batch_timeout=0.2, max_batch_size=8) as (c, state):
            before = (await c.get("/debug/metrics")).json()["batches_processed"]
            payloads = [{"model": MODEL_ID, "prompt": f"Request {i % 4}: Say hello.", "max_tokens": 5, "temperature": 0.0}
                        for i in range(8)]
            rs = await asyncio.gather(*[c.post("/v1/completions", json=p) for p in payloads])
            assert all(r.status_code == 200 for r in rs)
            texts = [r.json()["choices"][0]["text"] for r in rs]
            assert texts[:4] == texts[4:]                                     # duplicates fan back out
            m = (await c.get("/debug/metrics")).json()
            assert m["batches_processed"] == before + 1 and m["avg_batch_fill_pct"] == 100.0 and m["batch_fill_hist"][9] == 1
            assert m["decode_tokens_total"] > 0 and m["prompt_tokens_total"] > 0 and m["decode_tps_last"] > 0
            # 8 requests, 4 distinct prompts -> one prefill of 4 rows
            prefills = [e for e in model.engine.trace if e[0] == "enqueue" and e[1] != "device-tokens"]
            assert len(prefills) == 1 and prefills[0][1][0] == 4
    run(go())


def test_diverse_mode_disables_dedup(parts):
    model, _ = parts

    async def go():
        async with serving(parts, batch_timeout=0.2, diverse_mode=True) as (c, _):
            p = {"model": MODEL_ID, "prompt": "same prompt", "max_tokens": 3, "temperature": 0.0}
            rs = await asyncio.gather(*[c.post("/v1/completions", json=p) for _ in range(3)])
            assert all(r.status_code == 200 for r in rs)
            prefills = [e for e in model.engine.trace if e[0] == "enqueue" and e[1] != "device-tokens"]
            assert len(prefills) == 1 and prefills[0][1][0] == 3
    run(go())


def test_logprobs_echo_and_scoring_match_oracle(tiny, parts):
    _, tok = parts
    prompt = "Hello world"
    ids = np.asarray(tok._tokenizer([prompt], return_tensors="np")["input_ids"])
    ref = ref_generate.load(tiny, max_pos=2048)
    lg = np.asarray(ref(ids, cache=ref.make_cache(1, paged=False)), dtype=np.float32)[0]        # (L, V)
    lsm = ref_sample.log_softmax(lg)
    want_echo = [float(lsm[i, ids[0, i + 1]]) for i in range(ids.shape[1] - 1)]

    async def go():
        async with serving(parts) as (c, _):
            # the reference's test: echo + logprobs with max_tokens 0 (no decoding)
            r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": prompt, "max_tokens": 0, "temperature": 0.0,
                                                      "top_p": 1.0, "logprobs": 2, "echo": True})
            assert r.status_code == 200, r.text[:500]
            ch = r.json()["choices"][0]
            lp = ch["logprobs"]
            assert ch["text"] == prompt and len(lp["tokens"]) == ids.shape[1] - 1 == len(lp["token_logprobs"])
            np.testing.assert_allclose(lp["token_logprobs"], want_echo, atol=1e-4)
            assert all(len(d) == 2 for d in lp["top_logprobs"]) and lp["text_offset"] == [0] * len(lp["tokens"])
            best = [max(d.values()) for d in lp["top_logprobs"]]
            np.testing.assert_allclose(best, lsm[:-1].max(axis=1), atol=1e-4)
            # echo + generation resumes from the cached prefix; greedy tokens equal the plain path's
            r2 = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": prompt, "max_tokens": 4, "logprobs": 1, "echo": True})
            r3 = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": prompt, "max_tokens": 4, "logprobs": 1})
            a, b = r2.json()["choices"][0], r3.json()["choices"][0]
            assert a["text"] == prompt + b["text"]
            assert a["logprobs"]["tokens"][-len(b["logprobs"]["tokens"]):] == b["logprobs"]["tokens"]
            np.testing.assert_allclose(a["logprobs"]["token_logprobs"][-len(b["logprobs"]["tokens"]):],
                                       b["logprobs"]["token_logprobs"], atol=1e-5)
            assert r3.json()["usage"]["prompt_tokens"] == ids.shape[1]
            # temperature-scaled logprobs: log softmax(logits / T)
            r4 = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": prompt, "max_tokens": 0, "temperature": 0.5,
                                                       "logprobs": 1, "echo": True})
            lsm_t = ref_sample.log_softmax(lg * np.float32(2.0))
            np.testing.assert_allclose(r4.json()["choices"][0]["logprobs"]["token_logprobs"],
                                       [float(lsm_t[i, ids[0, i + 1]]) for i in range(ids.shape[1] - 1)], atol=1e-4)
            # perplexity = exp(mean NLL) of the same teacher-forced logprobs
            r5 = await c.post("/v1/perplexity", json={"model": MODEL_ID, "text": prompt})
            j = r5.json()
            assert r5.status_code == 200 and j["model"] == MODEL_ID and j["token_count"] == ids.shape[1] - 1
            assert abs(j["avg_nll"] + np.mean(want_echo)) < 1e-4 and abs(j["ppl"] - np.exp(-np.mean(want_echo))) < 1e-2
            r6 = await c.post("/v1/perplexity", json={"model": MODEL_ID, "text": "a"})
            assert r6.json() == {"model": MODEL_ID, "token_count": 0, "avg_nll": 0.0, "ppl": 1.0}
    run(go())


def _sse_events(text):
    events = [line[len("data: "):] for line in text.splitlines() if line.startswith("data: ")]
    assert events and events[-1] == "[DONE]"
    return [json.loads(e) for e in events[:-1]]


def test_streaming_chat_is_cobatched_and_completion_stream_finishes(parts):
    async def go():
        async with serving(parts) as (c, state):
            payload = {"model": MODEL_ID, "max_tokens": 6, "temperature": 0.0, "stream": True,
                       "messages": [{"role": "user", "content": "In one sentence, describe a tree."}]}
            ra, rb = await asyncio.gather(c.post("/v1/chat/completions", json=payload), c.post("/v1/chat/completions", json=payload))
            ea, eb = _sse_events(ra.text), _sse_events(rb.text)
            assert ra.status_code == 200 and ea[0]["object"] == "chat.completion.chunk"
            assert ea[0]["choices"][0]["delta"].get("role") == "assistant"
            assert ea[-1]["choices"][0]["finish_reason"] in ("stop", "length")
            text = lambda ev: "".join(e["choices"][0]["delta"].get("content") or "" for e in ev)
            assert text(ea) == text(eb) and ea[0]["id"] != eb[0]["id"]
            assert state.metrics.stream_batches_processed == 1                 # both streams decoded as one batch
            r = await c.post("/v1/chat/completions", json=dict(payload, n=2))
            assert r.status_code == 400
            r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "In one word, greet me.", "max_tokens": 5,
                                                      "temperature": 0.7, "top_p": 0.95, "stream": True})
            ev = _sse_events(r.text)
            assert r.status_code == 200 and ev[-1]["choices"][0]["finish_reason"] in ("stop", "length")
            assert all(e["object"] == "text_completion" for e in ev)
    run(go())


def test_prompt_longer_than_context_is_rejected(parts):
    async def go():
        async with serving(parts, max_context_length=16) as (c, _):
            r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "x" * 40, "max_tokens": 4})
            assert r.status_code == 400 and "Prompt too long" in r.text
    run(go())
