#!/usr/bin/env python3
"""Generates tests/golden/serving_*.npz -- oracle SOLO runs for the serving-path parity test at production widths.

The continuous scheduler (reference: mlx_parallm/server/main.py:1404-1726 over utils.py:983-1075) keeps sequences of
different ages in one batch.  The build runs them on a BLOCK-PAGED arena (64-token blocks, per-row block tables) and
takes an arriving prompt in chunks inside the live rows' decode steps (mi_step_enqueue_mixed).  Whatever the schedule,
a sequence must produce what it produces alone: K / V of a token depend only on the tokens at or before it and on its
absolute position (base.py:119-140 stores per row; llama.py:100-117 positions from the row's own offset).  So the
fixture is the oracle's SOLO run of every sequence, and tests/test_gpu_golden_wide.py::test_serving_schedule_* replays
this schedule on the device, teacher-forced with the oracle's tokens:

    rows 0..2   prompts of 1000 / 1015 / 1022 tokens, prefilled one by one, then decoded together for 16 steps
                (rows 1 and 2 cross KV length 1024 = the 16-block boundary and the second 256-key attention round)
    row 3       a 300-token prompt that arrives at step 4 and enters in chunks of 128 / 128 / 44 tokens inside the
                mixed steps 4..6, then decodes next to the others for steps 7..15

at the Mistral-7B (4 query heads per kv head) and Qwen3-14B (5 per kv head, q/k norms; int4 + rank-16 LoRA = config 5)
layer shapes, 2 decoder blocks, both KV modes.  "parity unpinned": outputs of the build's own oracle, not of MLX.
Run (about 10 min on 8 cores):  python tests/golden/make_golden_serving.py [name ...]
"""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import wide_models  # noqa: E402
from oracle import ref_generate, ref_model  # noqa: E402

OUT = Path(__file__).resolve().parent
ADAPTER_SEED = 77

SCHEDULE = dict(prompt_lens=[1000, 1015, 1022, 300], steps=16, arrive_step=4, chunks=[128, 128, 44], block_tokens=64)
CASES = {
    ("mistral-7b", "bf16", 11): [dict(name="serving_mistral_bf16_modelkv", paged=False, prompt_seed=201),
                                 dict(name="serving_mistral_bf16_f32kv", paged=True, prompt_seed=202)],
    ("qwen3-14b", "int4", 14): [dict(name="serving_qwen3_int4_lora_modelkv", paged=False, lora=True, prompt_seed=203),
                                dict(name="serving_qwen3_int4_lora_f32kv", paged=True, lora=True, prompt_seed=204)],
}


def n_tokens(seq: int) -> int:
    """Tokens the schedule samples for sequence `seq`: rows 0..2 their prefill token + `steps`; row 3 the token behind its
    last chunk (step arrive + len(chunks) - 1) + the remaining steps."""
    s = SCHEDULE
    if seq < 3:
        return 1 + s["steps"]
    return s["steps"] - (s["arrive_step"] + len(s["chunks"])) + 1


def prompts(case: dict, vocab: int):
    rng = np.random.default_rng(case["prompt_seed"])
    return [rng.integers(3, vocab, size=n).astype(np.int32) for n in SCHEDULE["prompt_lens"]]


def solo(ref, prompt, n, paged):
    toks, lps, top_ids, top_vals, margins = [], [], [], [], []
    gen = ref_generate.generate_step(prompt[None], ref, temp=0.0, paged=paged, return_logits=True, last_only=True)
    for (t, _p, logits, lp), _ in zip(gen, range(n)):
        toks.append(int(t[0, 0]))
        lps.append(float(lp[0]))
        order = np.argsort(-logits[0], kind="stable")[:8]
        top_ids.append(order)
        top_vals.append(logits[0][order])
        margins.append(float(logits[0][order[0]] - logits[0][order[1]]))
    return dict(tokens=np.asarray(toks, np.int32), logprobs=np.asarray(lps, np.float32),
                top_ids=np.stack(top_ids).astype(np.int32), top_vals=np.stack(top_vals).astype(np.float32),
                margins=np.asarray(margins, np.float32))


def main():
    only = set(sys.argv[1:])
    ref_model.CACHE_F64 = True
    for ck, cases in CASES.items():
        cases = [c for c in cases if not only or c["name"] in only]
        if not cases:
            continue
        with tempfile.TemporaryDirectory() as d:
            cfg = wide_models.build_checkpoint(d, *ck)
            ref = ref_generate.load(d, max_pos=wide_models.MAX_POS)
            if any(c.get("lora") for c in cases):
                ad = Path(d) / "adapter"
                wide_models.build_adapter(ad, cfg, ADAPTER_SEED)
                ref_generate.apply_adapters(ref.w, cfg["num_hidden_layers"], str(ad))
            for case in cases:
                t0 = time.time()
                out = {}
                for i, p in enumerate(prompts(case, cfg["vocab_size"])):
                    for k, v in solo(ref, p, n_tokens(i), case["paged"]).items():
                        out[f"seq{i}_{k}"] = v
                spec = dict(case, family=ck[0], precision=ck[1], model_seed=ck[2], adapter_seed=ADAPTER_SEED, **SCHEDULE)
                np.savez_compressed(OUT / f"{case['name']}.npz", spec=json.dumps(spec), **out)
                mm = min(float(out[f"seq{i}_margins"].min()) for i in range(4))
                print(f"{case['name']}: {time.time() - t0:.0f} s, smallest top1-top2 margin {mm:.5f}", flush=True)


if __name__ == "__main__":
    main()
