"""-m gpu: the library entry points the north_star names first -- ``batch_generate`` / ``generate`` / ``stream_generate``
(reference utils.py:473-543, 546-617, 429-471) -- on the HIP path, with the byte-level stub tokenizer of the tiny
checkpoints, against the oracle's restatement of the same host loop: chat template, LEFT padding (pads are attended,
quirk Q1), exactly ``max_tokens`` steps with no early stop for the batch form, stop at EOS for the single-prompt forms,
decode, cut at the eos / pad strings.

Models: the config-1 tiny shape (int4 g64, float32 activations: token ids bit-exact by the north_star bar) and a bf16
GQA model with the pool handing out float32 (PagedKVCache) caches, i.e. the reference's default numerics.
"""
import numpy as np
import pytest

from oracle import ref_generate

pytestmark = pytest.mark.gpu

from mlx_parallm_amd import utils  # noqa: E402

PROMPTS = ["hi", "a considerably longer prompt than the first", "mid-size prompt"]


def _ref_batch_generate(ref, tok, prompts, max_tokens, format_prompts, paged):
    """utils.py:473-543 over the oracle's generate_step."""
    if format_prompts:
        prompts = [tok.apply_chat_template([{"role": "user", "content": p}], add_generation_prompt=True, tokenize=False)
                   for p in prompts]
    tok._tokenizer.padding_side = "left"
    if tok.pad_token is None:
        tok._tokenizer.pad_token = tok.eos_token
        tok._tokenizer.pad_token_id = tok.eos_token_id
    ids = np.asarray(tok._tokenizer(prompts, padding=True)["input_ids"], dtype=np.int64)
    out = []
    for _, (t, _p) in zip(range(max_tokens), ref_generate.generate_step(ids, ref, temp=0.0, paged=paged)):
        out.append(t)
    toks = np.concatenate(out, axis=1)
    texts = [r.split(tok.eos_token)[0].split(tok.pad_token)[0] for r in tok.batch_decode(toks.tolist())]
    return texts, toks, ids


def _ref_generate(ref, tok, prompt, max_tokens, paged):
    """utils.py:546-617: greedy, stop at EOS (the EOS token is not part of the text)."""
    ids = np.asarray(tok.encode(prompt), dtype=np.int64)[None]
    out = []
    for _, (t, _p) in zip(range(max_tokens), ref_generate.generate_step(ids, ref, temp=0.0, paged=paged)):
        if int(t[0, 0]) == tok.eos_token_id:
            break
        out.append(int(t[0, 0]))
    return tok.decode(out), out


@pytest.fixture(params=["llama_q4_f32", "llama_bf16_gqa"])
def pair(request, tiny_dirs, monkeypatch):
    d, cfg = tiny_dirs[request.param]
    utils._kv_pool.clear()
    # the reference's pool hands out PagedKVCache (float32 first allocation, base.py:111-112)
    monkeypatch.setattr(utils, "DEFAULT_KV_DTYPE", "float32")
    model, tok = utils.load(d)
    ref = ref_generate.load(d, max_pos=512)
    yield model, tok, ref
    utils._kv_pool.clear()
    model.engine.close()


@pytest.mark.parametrize("format_prompts", [False, True])
def test_batch_generate_equals_the_oracle_host_loop(pair, format_prompts):
    model, tok, ref = pair
    n = 14
    want, want_toks, ids = _ref_batch_generate(ref, tok, PROMPTS, n, format_prompts, paged=True)
    assert ids.shape[0] == 3 and (ids[0] == tok.eos_token_id).sum() > 0          # really left-padded with pad = eos
    got = utils.batch_generate(model, tok, PROMPTS, max_tokens=n, format_prompts=format_prompts, temp=0.0)
    assert got == want
    # the same through generate_step directly: exactly max_tokens steps, ids bit-exact, the reference's (B,1) shapes
    cache = utils._kv_pool.get(model.head_dim, [model.n_kv_heads] * len(model.layers), 3, paged=True)
    steps = [t for _, (t, p) in zip(range(n), utils.generate_step(ids, model, temp=0.0, cache=cache))]
    assert all(t.shape == (3, 1) for t in steps) and np.array_equal(np.concatenate(steps, axis=1), want_toks)
    assert cache[0].offsets == [ids.shape[1] + n] * 3          # prefill + n one-token passes: one was computed ahead (utils.py:420-427)


def test_generate_and_stream_generate_equal_the_oracle(pair):
    model, tok, ref = pair
    for prompt in ("abc", "The quick brown fox"):
        want_text, want_ids = _ref_generate(ref, tok, prompt, 20, paged=True)
        assert utils.generate(model, tok, prompt, max_tokens=20, temp=0.0) == want_text
        pieces = list(utils.stream_generate(model, tok, prompt, max_tokens=20, temp=0.0))
        assert "".join(pieces) == want_text
        assert len(pieces) == len(want_ids) + 1                                   # one segment per token + the final flush


def test_sampled_batch_generate_is_reproducible_and_differs_from_greedy(pair):
    """temp > 0 through the library API: the Philox stream is keyed by `seed` (caller-visible)."""
    model, tok, _ref = pair
    a = utils.batch_generate(model, tok, PROMPTS, max_tokens=12, format_prompts=False, temp=1.0, top_p=0.9, seed=5)
    b = utils.batch_generate(model, tok, PROMPTS, max_tokens=12, format_prompts=False, temp=1.0, top_p=0.9, seed=5)
    c = utils.batch_generate(model, tok, PROMPTS, max_tokens=12, format_prompts=False, temp=1.0, top_p=0.9, seed=6)
    g = utils.batch_generate(model, tok, PROMPTS, max_tokens=12, format_prompts=False, temp=0.0)
    assert a == b and a != c and a != g
    # no seed: a fresh key per call (the reference's global mx.random state advances between calls; round-2 advisory)
    d1 = utils.batch_generate(model, tok, PROMPTS, max_tokens=12, format_prompts=False, temp=1.0, top_p=0.9)
    d2 = utils.batch_generate(model, tok, PROMPTS, max_tokens=12, format_prompts=False, temp=1.0, top_p=0.9)
    assert d1 != d2
    with pytest.raises(NotImplementedError):
        utils.batch_generate(model, tok, PROMPTS, max_tokens=2, repetition_penalty=1.2)
