#!/usr/bin/env python3
"""Generates tests/golden/wide_*.npz -- oracle outputs at PRODUCTION widths (BASELINE configs 2-5).

The small goldens (make_golden.py) pin the decode path on 64..128-wide models.  These pin it at the real layer
shapes: Mistral-7B (H 4096, 32 q / 8 kv heads x 128, I 14336, V 32000) and Qwen3-14B (H 5120, 40 / 8 heads x 128 with
q/k norms, I 17408, V 151936), truncated to 2 decoder blocks so that the NumPy oracle finishes in minutes; bf16, int4
g64 and int4 + a rank-16 LoRA adapter on q/v; both KV modes (PagedKVCache float32 = the reference's default numerics,
BatchedKVCache = model dtype); batch 8 with a 1024-token prompt decoded to KV length 1100 (the regime of the bench,
including the second 256-key round of the split-KV decode attention), plus a batch-32 and a ragged batch-64 case
(configs 4 and 5).  "parity unpinned": like every golden here these are outputs of the build's own oracle, not of MLX.

Accumulation envelope (round 3; `--envelope`): next to every case, the same run TEACHER-FORCED with the exact oracle's
tokens under the oracle's two float32-accumulating variants (oracle/numerics.py:set_accum -- chunks of 32 combined
sequentially / as a balanced tree; linears through oracle/c/accum_gemm.c, RMSNorm / attention in NumPy).  Committed
per case in tests/golden/envelope/<case>.npz: the variants' logprobs of the oracle's tokens and of its 8 largest
logits, their arg-max ids, and a JSON summary (max / mean |logprob - exact|, seq32-vs-pairwise, id flips).  No HIP
kernel is involved in these numbers: they are what summation order alone does to this model at these widths, and
tests/test_gpu_golden_wide.py takes its tolerances from them (<= 1.5 x the committed spread).
Round 4, ADDED to a case's envelope file: `--envelope --p16 <model-KV case>` = the two orders again with the softmax numerators
rounded to the KV dtype before P.V (keys f32_*_p16: what a 16-bit matrix-core attention with a single P operand computes --
the rounding that explained the device's round-3 excess); `--envelope --x2 <float32-KV case>` = with the float32
activations of every call of more than 16 rows and the operands of the prefill attention rounded to two bf16 terms (keys
f32_*_x2: the arithmetic of the two-term prefill GEMM and the split-operand prefill attention).  Full-depth
cases (`wide_*_full_*`: all 32 / 40 blocks, 2 x 128-token prompts, 16 steps) run from compact weights (ref_model.COMPACT).
`--logits` additionally stores the exact oracle's FULL last-position logits of a few (step, row) pairs of the sampled
case (config 3), against which the test checks the device's inverse-CDF draw.

Checkpoints are NOT committed: tests/wide_models.py rebuilds them from seeds on either side.  Stored per case: the
case spec (JSON), and per step the oracle's token ids, chosen-token logprobs, the 8 largest logits with their ids and
the top1-top2 margin.  Run (about 40 min on 8 cores, < 40 GB):  python tests/golden/make_golden_wide.py [case ...]
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

MAIN = dict(B=8, L0=1024, steps=77, temp=0.0, top_p=1.0)
# checkpoint (family, precision, seed) -> runs on it (adapter runs last: apply_adapters edits the weights in place)
CHECKPOINTS = {
    ("mistral-7b", "bf16", 11): [
        dict(name="wide_mistral_bf16_modelkv", paged=False, **MAIN, prompt_seed=101),           # config 2
        dict(name="wide_mistral_bf16_paged", paged=True, **MAIN, prompt_seed=102),
    ],
    ("mistral-7b", "int4", 12): [
        dict(name="wide_mistral_int4_modelkv", paged=False, **MAIN, prompt_seed=103),
        dict(name="wide_mistral_int4_paged", paged=True, **MAIN, prompt_seed=104),
        dict(name="wide_mistral_int4_topp_paged", paged=True, B=8, L0=1024, steps=24, temp=1.0, top_p=0.9,
             prompt_seed=105),                                                                   # config 3
        dict(name="wide_mistral_int4_lora_modelkv", paged=False, lora=True, **MAIN, prompt_seed=106),
        dict(name="wide_mistral_int4_lora_paged", paged=True, lora=True, **MAIN, prompt_seed=107),
    ],
    ("qwen3-14b", "bf16", 13): [
        dict(name="wide_qwen3_bf16_modelkv", paged=False, **MAIN, prompt_seed=108),
        dict(name="wide_qwen3_bf16_paged", paged=True, **MAIN, prompt_seed=109),
        dict(name="wide_qwen3_bf16_b32_modelkv", paged=False, B=32, L0=160, steps=6, temp=0.0, top_p=1.0,
             prompt_seed=110),                                                                   # config 4 (per-GPU shard x4)
        dict(name="wide_qwen3_bf16_b32_paged", paged=True, B=32, L0=160, steps=6, temp=0.0, top_p=1.0, prompt_seed=111),
    ],
    ("qwen3-14b", "int4", 14): [
        dict(name="wide_qwen3_int4_modelkv", paged=False, **MAIN, prompt_seed=112),
        dict(name="wide_qwen3_int4_paged", paged=True, **MAIN, prompt_seed=113),
        dict(name="wide_qwen3_int4_lora_modelkv", paged=False, lora=True, **MAIN, prompt_seed=114),
        dict(name="wide_qwen3_int4_lora_paged", paged=True, lora=True, **MAIN, prompt_seed=115),
        dict(name="wide_qwen3_int4_lora_b64_modelkv", paged=False, lora=True, B=64, L0=96, steps=6, temp=0.0,
             top_p=1.0, ragged=True, prompt_seed=116),                                           # config 5
        dict(name="wide_qwen3_int4_lora_b64_paged", paged=True, lora=True, B=64, L0=96, steps=6, temp=0.0, top_p=1.0,
             ragged=True, prompt_seed=117),
    ],
}
# int8 (north_star: "int4/int8-quantised weights"; nn.quantize with config["quantization"]["bits"] = 8, utils.py:679-690)
CHECKPOINTS[("mistral-7b", "int8", 15)] = [
    dict(name="wide_mistral_int8_modelkv", paged=False, **MAIN, prompt_seed=118),
    dict(name="wide_mistral_int8_paged", paged=True, **MAIN, prompt_seed=119),
]
# an f16 model at production width (the f16 instantiations of every kernel; in the float32-KV mode: the exact [hi | lo] bf16
# copy of the f16 weights on the float32-activation matrix-core paths)
# config 5 as BASELINE words it: top-p sampling at 64 sequences on the adapted int4 Qwen3-14B (V = 151936), ragged prompts;
# float32-KV, where the draw can be checked against the oracle's cumulative distribution (full logits of six pairs)
CHECKPOINTS[("qwen3-14b", "int4", 14)].append(
    dict(name="wide_qwen3_int4_lora_b64_topp_paged", paged=True, lora=True, B=64, L0=96, steps=6, temp=1.0, top_p=0.9,
         ragged=True, prompt_seed=122))
CHECKPOINTS[("mistral-7b", "f16", 16)] = [
    dict(name="wide_mistral_f16_modelkv", paged=False, **MAIN, prompt_seed=120),
    dict(name="wide_mistral_f16_paged", paged=True, **MAIN, prompt_seed=121),
]
# FULL DEPTH (round 4): every decoder block of the real models (llama.py:228 / qwen3.py:182 loop over all layers) -- 32 for
# Mistral-7B, 40 for Qwen3-14B -- at a context the CPU oracle can afford: depth is where accumulation error grows, and the
# envelope above had only been shown 2 blocks deep.  The checkpoints (14.5 GB bf16 / 8.3 GB int4) are drawn per tensor
# (wide_models.build_checkpoint_streamed) and the oracle keeps them in their storage format (ref_model.COMPACT,
# oracle/c/exact_gemm.c).  The 4th key element is the number of blocks.
FULL = dict(B=2, L0=128, steps=16, temp=0.0, top_p=1.0)
CHECKPOINTS[("mistral-7b", "bf16", 31, 32)] = [
    dict(name="wide_mistral_bf16_full_modelkv", paged=False, **FULL, prompt_seed=131),
    dict(name="wide_mistral_bf16_full_paged", paged=True, **FULL, prompt_seed=132),
]
CHECKPOINTS[("qwen3-14b", "int4", 32, 40)] = [
    dict(name="wide_qwen3_int4_full_modelkv", paged=False, **FULL, prompt_seed=133),
    dict(name="wide_qwen3_int4_full_paged", paged=True, **FULL, prompt_seed=134),
]
CHECKPOINTS[("mistral-7b", "int4", 33, 32)] = [
    dict(name="wide_mistral_int4_full_modelkv", paged=False, **FULL, prompt_seed=135),
    dict(name="wide_mistral_int4_full_paged", paged=True, **FULL, prompt_seed=136),
]
CHECKPOINTS[("qwen3-14b", "bf16", 34, 40)] = [
    dict(name="wide_qwen3_bf16_full_modelkv", paged=False, **FULL, prompt_seed=137),
    dict(name="wide_qwen3_bf16_full_paged", paged=True, **FULL, prompt_seed=138),
]
ADAPTER_SEED = 77


def load_ref(d, ck):
    """-> (cfg, oracle model) of checkpoint key ck built in directory d; full-depth keys load compact."""
    full = len(ck) > 3
    ref_model.COMPACT = full
    cfg = wide_models.build_checkpoint(d, *ck)
    return cfg, ref_generate.load(d, max_pos=wide_models.MAX_POS, compact=full)


def run_case(ref, cfg, ck, case):
    t0 = time.time()
    B, steps = case["B"], case["steps"]
    prompts = wide_models.prompts_for(case, cfg["vocab_size"])
    uniforms = np.random.default_rng(case["prompt_seed"] + 1000).random((steps + 2, B)).astype(np.float32)
    toks, lps, top_ids, top_vals, margins = [], [], [], [], []
    gen = ref_generate.generate_step(prompts, ref, temp=case["temp"], top_p=case["top_p"], paged=case["paged"],
                                     uniforms_fn=lambda s: uniforms[s], return_logits=True, last_only=True)
    for (t, _p, logits, lp), _ in zip(gen, range(steps)):
        toks.append(t[:, 0])
        lps.append(lp)
        order = np.argsort(-logits, axis=-1, kind="stable")[:, :8]
        top_ids.append(order)
        tv = np.take_along_axis(logits, order, axis=-1)
        top_vals.append(tv)
        margins.append(tv[:, 0] - tv[:, 1])
    spec = dict(case, family=ck[0], precision=ck[1], model_seed=ck[2], adapter_seed=ADAPTER_SEED,
                layers=(ck[3] if len(ck) > 3 else wide_models.LAYERS))
    np.savez_compressed(
        OUT / f"{case['name']}.npz", spec=json.dumps(spec), uniforms=uniforms,
        tokens=np.stack(toks).astype(np.int32), logprobs=np.stack(lps).astype(np.float32),
        top_ids=np.stack(top_ids).astype(np.int32), top_vals=np.stack(top_vals).astype(np.float32),
        margins=np.stack(margins).astype(np.float32))
    print(f"{case['name']}: {time.time() - t0:.0f} s, min top1-top2 margin {np.min(margins):.5f}, "
          f"margins <= 0.13: {int((np.stack(margins) <= 0.13).sum())} of {steps * B}", flush=True)


FULL_LOGITS = {"wide_mistral_int4_topp_paged": [(0, 0), (0, 5), (1, 2), (7, 7), (12, 3), (23, 1)],    # (step, row)
               "wide_qwen3_int4_lora_b64_topp_paged": [(0, 0), (0, 63), (1, 17), (2, 40), (4, 5), (5, 31)]}


def log_softmax64(lg):
    x = lg.astype(np.float64)
    x = x - x.max(axis=-1, keepdims=True)
    return x - np.log(np.exp(x).sum(axis=-1, keepdims=True))


def teacher_forced(ref, cfg, case, g, want_full=()):
    """Feed the golden's tokens; -> per step: logprob of the golden token, logprobs at the golden's top-8 ids, arg-max id
    (+ full logits of the (step, row) pairs in want_full)."""
    B, steps = case["B"], case["steps"]
    cache = ref.make_cache(B, paged=case["paged"])
    y = wide_models.prompts_for(case, cfg["vocab_size"])
    lp_tok, lp_top, amax, full = [], [], [], {}
    T = np.float64(case["temp"]) if case["temp"] != 0 else np.float64(1.0)
    for s in range(steps):
        logits = ref(y, cache=cache, last_only=True)[:, -1, :]
        lsm = log_softmax64(logits)                       # the sampler reports log_softmax(logits) (utils.py:345-364), not /temp
        tok = g["tokens"][s].astype(np.int64)
        lp_tok.append(lsm[np.arange(B), tok])
        lp_top.append(np.take_along_axis(lsm, g["top_ids"][s].astype(np.int64), axis=-1))
        amax.append(np.argmax(logits, axis=-1))
        for (ss, b) in want_full:
            if ss == s:
                full[(s, b)] = logits[b].astype(np.float32)
        y = tok[:, None]
    return np.stack(lp_tok), np.stack(lp_top), np.stack(amax), full


def run_envelope(ref, cfg, ck, case, p16=False, x2=False):
    """p16 (round 4, `--envelope --p16`): the two orders again with the softmax numerators rounded to the KV dtype before P.V
    (numerics.SDPA_P16 -- a 16-bit matrix-core attention), ADDED to the case's committed envelope file as
    f32_seq32_p16 / f32_pairwise_p16; only meaningful for the model-dtype KV cases."""
    from oracle import numerics

    path = OUT / f"{case['name']}.npz"
    g = np.load(path)
    B, steps = case["B"], case["steps"]
    # the golden's own log-probabilities at its top-8 ids (log Z from the chosen token where it is among them, else recomputed)
    res = {}
    t0 = time.time()
    for mode in ("f32_seq32", "f32_pairwise"):
        numerics.set_accum(mode)
        numerics.set_sdpa_p16(p16)
        numerics.set_x_split2(x2)
        try:
            res[mode + ("_p16" if p16 else "") + ("_x2" if x2 else "")] = teacher_forced(ref, cfg, case, g)
        finally:
            numerics.set_accum("exact")
            numerics.set_sdpa_p16(False)
            numerics.set_x_split2(False)
        print(f"  {case['name']} {mode}{' p16' if p16 else ''}: {time.time() - t0:.0f} s", flush=True)
    # exact reference values for the same quantities: chosen-token logprob is stored; top-8 logprobs = top_vals - logZ, with
    # logZ recovered from the greedy rows (token = top-1) or re-derived from the variants' own gap (sampled rows: see below)
    lp_exact = g["logprobs"].astype(np.float64)
    greedy = case["temp"] == 0.0
    epath = OUT / "envelope" / f"{case['name']}.npz"
    if p16 or x2:
        prev = np.load(epath)
        out = {k: prev[k] for k in prev.files}
        summ = json.loads(str(prev["summary"]))
    else:
        out = dict(spec=str(g["spec"]))
        summ = dict(case=case["name"], steps=steps, B=B)
    for mode, (lp_tok, lp_top, amax, _f) in res.items():
        d = np.abs(lp_tok - lp_exact)
        summ[mode] = dict(max_lp=float(d.max()), mean_lp=float(d.mean()),
                          id_flips=int((amax != g["top_ids"][:, :, 0]).sum()))
        if greedy:
            logz = g["top_vals"][:, :, 0].astype(np.float64) - lp_exact                   # token = top-1
            top_exact = g["top_vals"].astype(np.float64) - logz[:, :, None]
            summ[mode]["max_top8_lp"] = float(np.abs(lp_top - top_exact).max())
        out[f"lp_{mode}"] = lp_tok.astype(np.float32)
        out[f"top_lp_{mode}"] = lp_top.astype(np.float32)
        out[f"argmax_{mode}"] = amax.astype(np.int32)
    if not (p16 or x2):
        a, b = res["f32_seq32"], res["f32_pairwise"]
        summ["seq32_vs_pairwise"] = dict(max_lp=float(np.abs(a[0] - b[0]).max()), mean_lp=float(np.abs(a[0] - b[0]).mean()),
                                         max_top8_lp=float(np.abs(a[1] - b[1]).max()), id_diffs=int((a[2] != b[2]).sum()))
        summ["oracle_margins_le_1e-2"] = int((g["margins"] <= 1e-2).sum())
    out["summary"] = json.dumps(summ)
    (OUT / "envelope").mkdir(exist_ok=True)
    np.savez_compressed(epath, **out)
    print(f"envelope {json.dumps(summ)}", flush=True)


def run_full_logits(ref, cfg, ck, case):
    g = np.load(OUT / f"{case['name']}.npz")
    pairs = FULL_LOGITS[case["name"]]
    _a, _b, _c, full = teacher_forced(ref, cfg, case, g, want_full=pairs)
    np.savez_compressed(OUT / f"{case['name']}_logits.npz", pairs=np.asarray(pairs, np.int32),
                        logits=np.stack([full[tuple(p)] for p in pairs]), spec=str(g["spec"]))
    print(f"{case['name']}: full logits of {len(pairs)} (step, row) pairs stored", flush=True)


def main():
    args = sys.argv[1:]
    envelope, logits, p16, x2 = "--envelope" in args, "--logits" in args, "--p16" in args, "--x2" in args
    only = set(a for a in args if not a.startswith("--"))
    if envelope or logits:
        for ck, cases in CHECKPOINTS.items():
            cases = [c for c in cases if (not only or c["name"] in only) and (envelope or c["name"] in FULL_LOGITS)]
            if not cases:
                continue
            with tempfile.TemporaryDirectory() as d:
                cfg, ref = load_ref(d, ck)
                adapted = False
                for case in cases:
                    if case.get("lora") and not adapted:
                        ad = Path(d) / "adapter"
                        wide_models.build_adapter(ad, cfg, ADAPTER_SEED)
                        ref_generate.apply_adapters(ref.w, cfg["num_hidden_layers"], str(ad))
                        adapted = True
                    assert bool(case.get("lora")) == adapted, "adapter runs must come last"
                    if logits and case["name"] in FULL_LOGITS:
                        ref_model.CACHE_F64 = len(ck) == 3
                        run_full_logits(ref, cfg, ck, case)
                        ref_model.CACHE_F64 = False
                    if envelope:
                        if (not p16 or not case["paged"]) and (not x2 or case["paged"]):
                            run_envelope(ref, cfg, ck, case, p16, x2)
        return
    ref_model.CACHE_F64 = True
    for ck, cases in CHECKPOINTS.items():
        cases = [c for c in cases if not only or c["name"] in only]
        if not cases:
            continue
        with tempfile.TemporaryDirectory() as d:
            t0 = time.time()
            cfg, ref = load_ref(d, ck)
            print(f"checkpoint {ck}: built + loaded in {time.time() - t0:.0f} s", flush=True)
            adapted = False
            for case in cases:
                if case.get("lora") and not adapted:
                    ad = Path(d) / "adapter"
                    wide_models.build_adapter(ad, cfg, ADAPTER_SEED)
                    ref_generate.apply_adapters(ref.w, cfg["num_hidden_layers"], str(ad))
                    adapted = True
                assert bool(case.get("lora")) == adapted, "adapter runs must come last"
                run_case(ref, cfg, ck, case)


if __name__ == "__main__":
    main()
