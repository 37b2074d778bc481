"""Continuous (admit-on-step) scheduler on CPU: the slot logic of server/scheduler.py and the server routes in
``--scheduler continuous`` mode, with the device engine replaced by the oracle-backed fake."""
import asyncio
import json
import threading
import time

import httpx
import numpy as np
import pytest

from fake_engine import FakeModel
from mlx_parallm_amd import utils
from mlx_parallm_amd.server import main as srv
from mlx_parallm_amd.server.scheduler import ContinuousScheduler
from mlx_parallm_amd.server.state import model_registry
from mlx_parallm_amd.tokenizer_utils import load_tokenizer
from oracle import ref_generate

MODEL_ID = "tiny-scheduler-model"


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    from mlx_parallm_amd.tiny_model import build_tiny_model

    d = tmp_path_factory.mktemp("sched") / "tiny"
    build_tiny_model(d, seed=6, vocab_size=320, hidden_size=32, layers=2, heads=2, kv_heads=2,
                     intermediate_size=64, quantize_model=False, dtype="float32")
    return str(d)


@pytest.fixture()
def parts(tiny):
    utils._kv_pool._pool.clear()
    model_registry.clear()
    return FakeModel(tiny, max_pos=2048), load_tokenizer(tiny)


def _alone(tiny, ids, n, eos):
    """The oracle's greedy continuation of one sequence on its own: (token ids without EOS, finish reason)."""
    ref = ref_generate.load(tiny, max_pos=2048)
    out = []
    for (t, _), _ in zip(ref_generate.generate_step(np.asarray(ids)[None], ref, paged=False), range(n)):
        if int(t[0, 0]) == eos:
            return out, "stop"
        out.append(int(t[0, 0]))
    return out, "length"


@pytest.mark.parametrize("chunk", [0, 8])
def test_scheduler_sequences_equal_their_solo_runs_and_slots_are_recycled(tiny, parts, chunk):
    model, tok = parts
    sched = ContinuousScheduler(model, tok, max_slots=2, chunk_tokens=chunk)
    sched.start()
    done = {}
    texts = {}
    ev = threading.Event()

    def sink_for(name):
        def sink(seq, delta, reason):
            texts.setdefault(name, []).append(delta or "")
            if reason is not None:
                done[name] = (list(seq.generated), reason)
                if len(done) == 4:
                    ev.set()
        return sink

    prompts = {"a": tok.encode("first request, a long one"), "b": tok.encode("second"), "c": tok.encode("third request"),
               "d": tok.encode("the fourth")}
    limits = {"a": 12, "b": 3, "c": 6, "d": 5}
    sched.submit(prompts["a"], limits["a"], 0.0, 1.0, sink_for("a"))
    sched.submit(prompts["b"], limits["b"], 0.0, 1.0, sink_for("b"))
    time.sleep(0.3)                                   # c and d arrive while a / b are decoding; only 2 slots exist
    sched.submit(prompts["c"], limits["c"], 0.0, 1.0, sink_for("c"))
    sched.submit(prompts["d"], limits["d"], 0.0, 1.0, sink_for("d"))
    assert ev.wait(timeout=120)
    sched.stop()
    eos = tok.eos_token_id
    for name in prompts:
        want, reason = _alone(tiny, prompts[name], limits[name], eos)
        assert done[name] == (want, reason), name
        assert "".join(texts[name]) == tok.decode(want)
    tr = model.engine.trace
    if chunk == 0:
        prefills = [e for e in tr if e[0] == "enqueue_rows" and e[2] != "device-tokens" and e[2][1] > 1]
        assert len(prefills) == 4 and all(len(e[1]) == 1 for e in prefills)       # one unpadded prefill per sequence
        assert not [e for e in tr if e[0] == "enqueue_mixed"]
    else:
        # chunked prefill: prompts enter the cache in chunks of <= 8 tokens inside mixed steps; no prompt is prefilled alone
        mixed = [e for e in tr if e[0] == "enqueue_mixed"]
        assert mixed and all(sum(n for n in e[2] if n > 1) <= chunk and max(e[2]) <= chunk for e in mixed)
        assert not [e for e in tr if e[0] == "enqueue_rows" and e[2] != "device-tokens" and e[2][1] > 1]
        assert sum(sum(n for n, w in zip(e[2], e[3]) if not (n == 1 and w)) for e in mixed) >= sum(len(p) for p in prompts.values()) - 8
        assert any(1 in e[2] and max(e[2]) > 1 for e in mixed)                    # a chunk rode next to a decoding row
        assert sched.mixed_steps == len(mixed) and sched.prefills == 4 and sched.chunks >= 4
    assert any(e[0] == "enqueue_rows" and len(e[1]) == 2 for e in tr)          # two sequences share decode steps
    assert any(e[0] == "enqueue_rows" and e[2] == "device-tokens" for e in tr)  # one-step-ahead on a stable row set
    assert sum(1 for e in tr if e[0] == "reset_row") == 4 and sched.max_rows_seen == 2
    with pytest.raises(ValueError):
        ContinuousScheduler(model, tok, max_slots=1).submit([], 4, 0.0, 1.0, lambda *a: None)


def test_per_row_sampling_parameters(tiny, parts):
    """A greedy and a hot sequence share steps; the greedy one must still equal its solo greedy run."""
    model, tok = parts
    sched = ContinuousScheduler(model, tok, max_slots=2)
    sched.start()
    done, ev = {}, threading.Event()

    def sink(name):
        def f(seq, delta, reason):
            if reason is not None:
                done[name] = list(seq.generated)
                if len(done) == 2:
                    ev.set()
        return f

    p = tok.encode("same prompt for both")
    sched.submit(p, 10, 0.0, 1.0, sink("greedy"))
    sched.submit(p, 10, 1.5, 0.95, sink("hot"))
    assert ev.wait(timeout=120)
    sched.stop()
    want, _ = _alone(tiny, p, 10, tok.eos_token_id)
    assert done["greedy"] == want and done["hot"] != want


def test_replica_pool_spreads_sequences_over_replicas(tiny, parts):
    """One scheduler per model replica (one per GPU in production); new sequences go to the least loaded."""
    from mlx_parallm_amd.server.scheduler import ReplicaPool

    model, tok = parts
    other = FakeModel(tiny, max_pos=2048)
    pool = ReplicaPool([model, other], tok, max_slots=2)
    pool.start()
    done, ev = {}, threading.Event()

    def sink(i):
        def f(seq, delta, reason):
            if reason is not None:
                done[i] = list(seq.generated)
                if len(done) == 4:
                    ev.set()
        return f

    prompts = [tok.encode(f"prompt number {i}") for i in range(4)]
    for i, p in enumerate(prompts):
        pool.submit(p, 6, 0.0, 1.0, sink(i))
    assert ev.wait(timeout=120)
    pool.stop()
    for i, p in enumerate(prompts):
        assert done[i] == _alone(tiny, p, 6, tok.eos_token_id)[0]
    used = [sum(1 for e in m.engine.trace if e[0] == "reset_row") for m in (model, other)]
    assert used == [2, 2]                                  # two sequences each


async def _serving(parts, **cfg):
    config = srv.ServerConfig(model_path=MODEL_ID, scheduler="continuous", **cfg)
    app = srv.create_app(config, model=parts[0], tokenizer=parts[1], model_id=MODEL_ID)
    return app


def _events(text):
    ev = [line[len("data: "):] for line in text.splitlines() if line.startswith("data: ")]
    assert ev and ev[-1] == "[DONE]"
    return [json.loads(e) for e in ev[:-1]]


def test_server_routes_in_continuous_mode(tiny, parts):
    model, tok = parts

    async def go():
        app = await _serving(parts, max_batch_size=3)
        async with app.router.lifespan_context(app):
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://s", timeout=120) as c:
                state = app.state.server
                assert state.scheduler is not None and not state.tasks
                # more concurrent requests than slots: all finish, each equal to its solo greedy run
                prompts = [f"Request {i}: say hello." for i in range(7)]
                rs = await asyncio.gather(*[c.post("/v1/completions", json={
                    "model": MODEL_ID, "prompt": p, "max_tokens": 4 + i, "temperature": 0.0}) for i, p in enumerate(prompts)])
                for i, (p, r) in enumerate(zip(prompts, rs)):
                    assert r.status_code == 200, r.text[:300]
                    ids = np.asarray(tok._tokenizer([p], return_tensors="np")["input_ids"])[0]
                    want, reason = _alone(tiny, ids, 4 + i, tok.eos_token_id)
                    j = r.json()
                    assert j["choices"][0]["text"] == tok.decode(want) and j["choices"][0]["finish_reason"] == reason
                    assert j["usage"] == {"prompt_tokens": len(ids), "completion_tokens": len(want), "total_tokens": len(ids) + len(want)}
                assert state.scheduler.max_rows_seen == 3
                # n choices = n sequences; chat; streams
                r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "two", "max_tokens": 5, "temperature": 0.9,
                                                          "top_p": 0.9, "n": 2})
                assert r.status_code == 200 and [ch["index"] for ch in r.json()["choices"]] == [0, 1]
                r = await c.post("/v1/chat/completions", json={"model": MODEL_ID, "max_tokens": 5, "temperature": 0.0,
                                                               "messages": [{"role": "user", "content": "hi"}]})
                assert r.status_code == 200 and r.json()["choices"][0]["message"]["role"] == "assistant"
                payload = {"model": MODEL_ID, "max_tokens": 6, "temperature": 0.0, "stream": True,
                           "messages": [{"role": "user", "content": "In one sentence, describe a tree."}]}
                ra, rb = await asyncio.gather(c.post("/v1/chat/completions", json=payload), c.post("/v1/chat/completions", json=payload))
                ea, eb = _events(ra.text), _events(rb.text)
                text = lambda ev: "".join(e["choices"][0]["delta"].get("content") or "" for e in ev)
                assert text(ea) == text(eb) and ea[-1]["choices"][0]["finish_reason"] in ("stop", "length")
                assert ea[0]["choices"][0]["delta"].get("role") == "assistant"
                r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "stream me", "max_tokens": 5, "stream": True})
                ev = _events(r.text)
                assert ev[-1]["choices"][0]["finish_reason"] in ("stop", "length") and all(e["object"] == "text_completion" for e in ev)
                # the engine is borrowed between steps for the logprobs path while a long generation runs
                long_task = asyncio.create_task(c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "keep going", "max_tokens": 40}))
                await asyncio.sleep(0.05)
                r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "Hello world", "max_tokens": 0, "logprobs": 1, "echo": True})
                assert r.status_code == 200 and r.json()["choices"][0]["logprobs"]["tokens"]
                r = await c.post("/v1/perplexity", json={"model": MODEL_ID, "text": "Hello world"})
                assert r.status_code == 200 and r.json()["token_count"] > 0
                assert (await long_task).status_code == 200
                r = await c.post("/v1/completions", json={"model": MODEL_ID, "prompt": "x", "max_tokens": 2, "n": 0})
                assert r.status_code == 500
                m = (await c.get("/debug/metrics")).json()
                assert m["decode_tokens_total"] > 0 and m["prompt_tokens_total"] > 0
    asyncio.run(go())


def test_borrower_never_gets_the_engine_while_a_device_fed_step_is_pending(tiny, parts):
    """A borrower (logprobs / echo / perplexity) overwrites the engine's device-resident token feed.  It must only be
    let in when every enqueued step has been read back -- otherwise the step enqueued ahead with tokens=None would
    continue from the borrower's tokens (scheduler.py borrow race)."""
    model, tok = parts
    eng = model.engine
    sched = ContinuousScheduler(model, tok, max_slots=2)
    sched.start()
    done, ev = {}, threading.Event()

    def sink(seq, delta, reason):
        if reason is not None:
            done["a"] = list(seq.generated)
            ev.set()

    p = tok.encode("a sequence that decodes with the next step always enqueued ahead")
    sched.submit(p, 48, 0.0, 1.0, sink)
    borrows, bad = 0, []
    while not ev.is_set():
        with sched.borrow_engine():
            borrows += 1
            if eng._results or sched._step_pending:
                bad.append((dict(eng._results), sched._step_pending))
            eng._last_tokens = np.full((5, 1), 7)          # what a borrower's own steps leave in d_next / last_n
        time.sleep(0.001)
    sched.stop()
    want, _ = _alone(tiny, p, 48, tok.eos_token_id)
    assert borrows >= 3 and not bad, (borrows, bad[:2])
    assert done["a"] == want
    assert any(e[0] == "enqueue_rows" and e[2] == "device-tokens" for e in eng.trace)   # pipelining still happened


def test_cancel_releases_the_slot(tiny, parts):
    """Sequence.cancel() (request timed out / client disconnected): the sequence stops at the next step boundary, its
    sink sees "cancelled", and the only slot goes to the next request instead of decoding to max_tokens."""
    model, tok = parts
    sched = ContinuousScheduler(model, tok, max_slots=1)
    sched.start()
    got, ev_first, ev_done = {}, threading.Event(), threading.Event()

    def sink_a(seq, delta, reason):
        if len(seq.generated) >= 3:
            ev_first.set()
        if reason is not None:
            got["a"] = (len(seq.generated), reason)

    def sink_b(seq, delta, reason):
        if reason is not None:
            got["b"] = (list(seq.generated), reason)
            ev_done.set()

    a = sched.submit(tok.encode("abandoned request"), 1500, 0.0, 1.0, sink_a)
    pb = tok.encode("the next one")
    sched.submit(pb, 4, 0.0, 1.0, sink_b)
    queued = sched.submit(tok.encode("cancelled before admission"), 4, 0.0, 1.0, lambda s, d, r: got.setdefault("q", r))
    queued.cancel()
    assert ev_first.wait(timeout=60)
    a.cancel()
    assert ev_done.wait(timeout=60)
    sched.stop()
    want, reason = _alone(tiny, pb, 4, tok.eos_token_id)
    assert got["a"][1] == "cancelled" and got["a"][0] < 1500
    assert got["b"] == (want, reason) and got.get("q") == "cancelled"
