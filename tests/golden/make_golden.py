#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- golden input/output vectors of the decode path.

The reference ships NO golden vectors for this path and MLX cannot run here, so these
are produced by the build's own oracle (oracle/, a CPU restatement of the reference source) on
checkpoints made by mlx_parallm_amd.tiny_model.build_tiny_model with fixed seeds.  They pin the
oracle against regressions (tests/test_golden.py, CPU) and give the GPU path a committed target
(tests/test_gpu_golden.py).  "parity unpinned": they are not outputs of MLX.

Each case stores: the build_tiny_model kwargs (JSON), prompts, KV mode, sampling settings + injected
uniforms, and per step: token ids, chosen-token logprobs, the 8 largest logits with their ids and the
top1-top2 margin.  Run:  python tests/golden/make_golden.py
"""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from mlx_parallm_amd.tiny_model import build_tiny_model  # noqa: E402
from oracle import ref_generate  # noqa: E402

OUT = Path(__file__).resolve().parent

CASES = {
    # BASELINE config 1: scripts/build_tiny_model.py defaults (hidden 64, 8 layers, 4 heads, int4 g64, tied,
    # float32 activations, vocab 151936), greedy, batch 1, reference-default PagedKVCache (float32 KV)
    "tiny_default_greedy_b1": dict(model=dict(seed=0, with_tokenizer=False), B=1, L0=12, steps=16, paged=True, temp=0.0, top_p=1.0),
    # left-padded batch of 4 (pads attended, quirk Q1), model-dtype KV
    "tiny_leftpad_b4": dict(model=dict(seed=1, vocab_size=2048, with_tokenizer=False), B=4, L0=10, steps=12, paged=False,
                            temp=0.0, top_p=1.0, pad=True),
    # bf16 GQA llama, both KV modes
    "llama_bf16_gqa_paged": dict(model=dict(seed=2, vocab_size=1024, dtype="bfloat16", quantize_model=False, hidden_size=128,
                                            layers=3, heads=8, kv_heads=2, intermediate_size=256, head_dim=32,
                                            tie_word_embeddings=False, norm_jitter=0.1, with_tokenizer=False),
                                 B=2, L0=9, steps=10, paged=True, temp=0.0, top_p=1.0),
    "llama_q4_bf16_model_kv": dict(model=dict(seed=3, vocab_size=1024, dtype="bfloat16", quantize_model=True, hidden_size=128,
                                              layers=2, heads=8, kv_heads=2, intermediate_size=256, head_dim=32,
                                              tie_word_embeddings=False, norm_jitter=0.1, with_tokenizer=False),
                                   B=2, L0=9, steps=10, paged=False, temp=0.0, top_p=1.0),
    # qwen3-shaped: q/k norms, 5 query heads per kv head, head_dim 64
    "qwen3_bf16_paged": dict(model=dict(seed=4, model_type="qwen3", vocab_size=1024, dtype="bfloat16", quantize_model=False,
                                        hidden_size=128, layers=2, heads=5, kv_heads=1, intermediate_size=256, head_dim=64,
                                        tie_word_embeddings=False, norm_jitter=0.1, with_tokenizer=False),
                             B=3, L0=8, steps=10, paged=True, temp=0.0, top_p=1.0),
    # BASELINE config 3 semantics: top-p 0.9 sampling with logprobs, injected uniforms
    "tiny_top_p": dict(model=dict(seed=5, vocab_size=2048, with_tokenizer=False), B=4, L0=8, steps=12, paged=False,
                       temp=1.0, top_p=0.9),
}


def run_case(name, spec):
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))
    rng = np.random.default_rng(sum(ord(c) for c in name))
    with tempfile.TemporaryDirectory() as d:
        cfg = build_tiny_model(d, **spec["model"])
        ref = ref_generate.load(d, max_pos=256)
        B, L0, steps = spec["B"], spec["L0"], spec["steps"]
        prompts = rng.integers(3, cfg["vocab_size"], size=(B, L0))
        if spec.get("pad"):
            for b in range(B):
                prompts[b, : int(rng.integers(0, L0 // 2))] = 1
        uniforms = rng.random((steps + 2, B)).astype(np.float32)
        toks, lps, top_ids, top_vals, margins = [], [], [], [], []
        gen = ref_generate.generate_step(prompts, ref, temp=spec["temp"], top_p=spec["top_p"], paged=spec["paged"],
                                         uniforms_fn=lambda s: uniforms[s], return_logits=True)
        for (t, _p, logits, lp), _ in zip(gen, range(steps)):
            toks.append(t[:, 0])
            lps.append(lp)
            order = np.argsort(-logits, axis=-1, kind="stable")[:, :8]
            top_ids.append(order)
            tv = np.take_along_axis(logits, order, axis=-1)
            top_vals.append(tv)
            margins.append(tv[:, 0] - tv[:, 1])
        np.savez_compressed(
            OUT / f"{name}.npz", spec=json.dumps(spec), prompts=prompts.astype(np.int32), uniforms=uniforms,
            tokens=np.stack(toks).astype(np.int32), logprobs=np.stack(lps).astype(np.float32),
            top_ids=np.stack(top_ids).astype(np.int32), top_vals=np.stack(top_vals).astype(np.float32),
            margins=np.stack(margins).astype(np.float32))
        print(f"{name}: min top1-top2 margin {np.min(margins):.4f}")


if __name__ == "__main__":
    only = sys.argv[1:]
    for n, s in CASES.items():
        if not only or n in only:
            run_case(n, s)
