#!/usr/bin/env python3
"""bench.py -- decode tokens/sec of the MI355X batched-decode engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (the whole model forward for one new token per sequence +
sampling) over one batch of B sequences per GPU.  Workload at N=1: BASELINE.json configs[1] --
Mistral-7B-Instruct-v0.1 shape, bf16, batch 8, greedy decode -- with synthetic token ids and
random-init weights (no network for checkpoints), the prompt of `--context` tokens prefilled
(timed separately) so that decode starts at KV length 1024.  N > 1: one process per GPU, each
holding a full replica (rank 0 generates the weights, RCCL broadcast over xGMI) and its own
batch shard; no collective inside the timed region except the bracketing barriers ("weak").

`python bench.py --gpus N` typed as a plain command (no torchrun environment) launches its own N rank processes
before anything touches a GPU and relays rank 0's line (mlx_parallm_amd.distributed.self_launch).

Rank 0 prints ONE JSON line.  The headline leg (`value`, `ms_per_step`, `roofline`, `prefill_tokens_per_sec`) runs the
REFERENCE'S numerics: float32 KV = PagedKVCache semantics, what `generate_step` / `batch_generate` really run on
(utils.py:392 + base.py:111-112), the mode in which ids bit-exact / logprobs <= 1e-3 against the oracle are asserted.  The
same workload with KV in the model dtype (BatchedKVCache semantics, narrower arithmetic than the reference's on this path)
is timed next to it and reported under `fast_mode` (`--kv-dtype model` swaps the two; the float32 leg then appears as
`reference_numerics`).  `roofline` is for the dominant kernel (the fused gate|up SwiGLU
weight-streaming kernel): algorithmic bytes per launch / its mean launch time from HIP events
on the engine's stream, measured in a second instrumented pass of the same K steps.
`cpu_baseline` is the oracle's C restatement (oracle/c) timed on this box's host cores on a
bounded sample (see the "sample" field); it is a reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SHAPES = {
    "mistral-7b": dict(model_type="mistral", hidden_size=4096, num_hidden_layers=32, num_attention_heads=32,
                       num_key_value_heads=8, intermediate_size=14336, vocab_size=32000, rms_norm_eps=1e-5,
                       rope_theta=10000.0, tie_word_embeddings=False, max_position_embeddings=32768),
    "qwen3-14b": dict(model_type="qwen3", hidden_size=5120, num_hidden_layers=40, num_attention_heads=40,
                      num_key_value_heads=8, head_dim=128, intermediate_size=17408, vocab_size=151936,
                      rms_norm_eps=1e-6, rope_theta=1000000.0, tie_word_embeddings=False,
                      max_position_embeddings=40960),
    "tiny": dict(model_type="llama", hidden_size=64, num_hidden_layers=8, num_attention_heads=4,
                 num_key_value_heads=4, intermediate_size=128, vocab_size=151936, rms_norm_eps=1e-6,
                 rope_theta=10000.0, tie_word_embeddings=True, max_position_embeddings=4096),
}


def tensor_specs(cfg):
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    yield "model.embed_tokens.weight", (V, H), "mat"
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        yield p + "self_attn.q_proj.weight", (nh * D, H), "mat"
        yield p + "self_attn.k_proj.weight", (nkv * D, H), "mat"
        yield p + "self_attn.v_proj.weight", (nkv * D, H), "mat"
        yield p + "self_attn.o_proj.weight", (H, nh * D), "mat"
        if cfg["model_type"] == "qwen3":
            yield p + "self_attn.q_norm.weight", (D,), "norm"
            yield p + "self_attn.k_norm.weight", (D,), "norm"
        yield p + "mlp.gate_proj.weight", (I, H), "mat"
        yield p + "mlp.up_proj.weight", (I, H), "mat"
        yield p + "mlp.down_proj.weight", (H, I), "mat"
        yield p + "input_layernorm.weight", (H,), "norm"
        yield p + "post_attention_layernorm.weight", (H,), "norm"
    yield "model.norm.weight", (H,), "norm"
    if not cfg["tie_word_embeddings"]:
        yield "lm_head.weight", (V, H), "mat"


def load_synthetic(engine, cfg, seed, quant_bits, rank, world, dist, stats=None, device=None):
    """Random-init weights N(0, 0.02^2) (norm weights = 1) generated on rank 0, replicated to the other ranks in a few
    LARGE flat buckets -- one RCCL broadcast per ~1 GiB bucket (a ring over point-to-point xGMI links is per-link
    bound, so few large collectives) -- optionally MLX-affine quantised (group 64) and handed
    to the engine as views into the bucket.  ``engine`` None (dry run): everything except the engine calls.
    ``stats``: dict that receives broadcast_seconds / broadcast_bytes / broadcast_buckets."""
    import torch

    from mlx_parallm_amd import distributed as D
    from mlx_parallm_amd.quant import quantize

    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev)
    specs = list(tensor_specs(cfg))
    kinds = {name: kind for name, _shape, kind in specs}
    index = {name: i for i, (name, _s, _k) in enumerate(specs)}
    nbytes, bsec, bbytes = 0, 0.0, 0
    buckets = D.plan_buckets([(n, sh) for n, sh, _k in specs], elem_size=2, bucket_bytes=1 << 30)
    for bucket in buckets:
        flat = torch.empty(D.bucket_numel(bucket), dtype=torch.bfloat16, device=dev)
        views = {name: flat[off:off + n].view(shape) for name, shape, off, n in bucket}
        if rank == 0:
            for name, t in views.items():
                if kinds[name] == "norm":
                    t.fill_(1.0)
                else:
                    gen.manual_seed(seed * 1000003 + index[name])
                    t.copy_(torch.randn(t.shape, generator=gen, device=dev, dtype=torch.float32) * 0.02)
        if world > 1:
            sec, nb = D.broadcast_bucket(flat, src=0)
            bsec, bbytes = bsec + sec, bbytes + nb
        for name, t in views.items():
            if kinds[name] == "mat" and quant_bits:
                packed, scales, biases = quantize(t, 64, quant_bits)
                base = name[: -len(".weight")]
                if engine is not None:
                    engine.set_tensor(base + ".weight", packed)
                    engine.set_tensor(base + ".scales", scales)
                    engine.set_tensor(base + ".biases", biases)
                nbytes += packed.numel() * 4 + scales.numel() * 4
            else:
                if engine is not None:
                    engine.set_tensor(name, t)
                nbytes += t.numel() * 2
        del views, flat
    if dev.type == "cuda":
        torch.cuda.synchronize()
    if engine is not None:
        engine.finalize()
    if stats is not None:
        stats.update(broadcast_seconds=round(bsec, 4), broadcast_bytes=int(bbytes), broadcast_buckets=len(buckets) if world > 1 else 0)
    return nbytes


def streamed_weight_bytes(cfg, quant_bits):
    """W of SURVEY §8d / App. B: every matmul weight read once per decode step (not the embedding table)."""
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    per_layer = (nh + 2 * nkv) * D * H + H * nh * D + 3 * I * H
    params = cfg["num_hidden_layers"] * per_layer + V * H
    bpp = 2.0 if not quant_bits else quant_bits / 8.0 + 4.0 / 64.0
    return params * bpp, params


def numbers(cfg, quant_bits, B, K, W, ctx, leg, kv_dtype, kern, world):
    """Algorithmic bytes and rates of one timed leg: the step, and the launches of linear `kern`."""
    w_bytes, n_params = streamed_weight_bytes(cfg, quant_bits)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    S_mid = ctx + W + K // 2
    bpp = 2.0 if not quant_bits else quant_bits / 8.0 + 4.0 / 64.0
    kvb = 2 if kv_dtype == "model" else 4
    ab = 2 if kv_dtype == "model" else 4                    # activation bytes (float32 after layer 0 in PagedKVCache mode)
    step_bytes = (w_bytes + B * cfg["num_hidden_layers"] * 2 * nkv * D * S_mid * kvb
                  + B * cfg["num_hidden_layers"] * 2 * nkv * D * kvb + B * cfg["vocab_size"] * 4)
    kern_bytes = {
        "gemv_gate_up": 2 * I * H * bpp + B * H * ab + B * I * ab,
        "gemv_down": I * H * bpp + B * I * ab + 2 * B * H * ab,
        "gemv_qkv": (nh + 2 * nkv) * D * H * bpp + B * H * ab + B * (nh + 2 * nkv) * D * ab,
        "gemv_o": H * nh * D * bpp + B * nh * D * ab + 2 * B * H * ab,
        "gemv_head": cfg["vocab_size"] * H * bpp + B * H * ab + B * cfg["vocab_size"] * 4,
    }.get(kern, 0.0)
    ms_per_step = leg["elapsed"] / K * 1e3
    avg_ms = leg["total_ms"] / max(leg["n_launch"], 1)
    achieved = kern_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # prefill: algorithmic flops = 2 W T + NL 2 S^2 Hq D per sequence (causal), whatever the implementation
    # spends on top (the float32 leg multiplies the exact three-way split of x: 3 x the MFMA work for the same flops)
    pf_flops = B * (2.0 * n_params * ctx + cfg["num_hidden_layers"] * 2.0 * ctx * ctx * nh * D)
    return dict(ms_per_step=ms_per_step, value=world * B * K / leg["elapsed"], step_bytes=step_bytes,
                kern_bytes=kern_bytes, avg_ms=avg_ms, achieved=achieved,
                prefill=world * B * ctx / leg["t_prefill"], pf_tflops=pf_flops / leg["t_prefill"] / 1e12, pf_flops=pf_flops,
                rank_min=B * K / leg["own_max"], rank_max=B * K / leg["own_min"])


def apply_lora(engine, cfg, layers, seed):
    """A ~ U(-1/sqrt(K), 1/sqrt(K)), B ~ N(0, 0.01^2), rank 16, scale 10 (lora_init.py:68-72 defaults)."""
    import torch

    H, nh, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    g = torch.Generator().manual_seed(seed + 99)
    for li in range(cfg["num_hidden_layers"] - layers, cfg["num_hidden_layers"]):
        for key, n in (("self_attn.q_proj", nh * D), ("self_attn.v_proj", nkv * D)):
            a = (torch.rand((H, 16), generator=g) * 2 - 1) / (H ** 0.5)
            b = torch.randn((16, n), generator=g) * 0.01
            engine.set_lora(li, key, a, b, 10.0)


OTHER_CONFIGS = [
    # BASELINE.json configs 3 / 4 / 5 at their per-GPU shard (configs 4 / 5 shard 32 / 64 sequences over 4 / 8 GPUs by the
    # sequence = 8 per GPU at full scale; the FULL batch on ONE GPU is the harder single-GPU statement and what the
    # round-2 profiles measured, so that is what runs here)
    dict(config="3: Mistral-7B int4-g64, batch 8, top-p 0.9 / T=1 sampling with logprobs", workload="mistral-7b-int4", batch=8,
         lora=0, kv_modes=("model", "float32"), mixed=False),
    dict(config="3b (north_star: int4 / int8-quantised weights): Mistral-7B int8-g64, batch 8, top-p 0.9 / T=1 sampling with logprobs",
         workload="mistral-7b-int8", batch=8, lora=0, kv_modes=("model",), mixed=False),
    dict(config="4: Qwen3-14B bf16, batch 32 (all of config 4's sequences on one GPU), greedy", workload="qwen3-14b-bf16",
         batch=32, lora=0, kv_modes=("model",), mixed=False),
    dict(config="5: Qwen3-14B int4-g64 + rank-16 LoRA on q/v of the last 8 layers, batch 64 (all of config 5's sequences on "
                "one GPU), top-p sampling; decode steps, and mixed prefill + decode steps", workload="qwen3-14b-int4", batch=64,
         lora=8, kv_modes=("model",), mixed=True),
]


def other_config_legs(args, seed):
    """Short legs of the other BASELINE configurations, on the driver's clock: same timing protocol as the headline leg
    (prefill untimed here, W warm-up + K timed steps, then K instrumented steps for the dominant linear)."""
    import numpy as np

    from mlx_parallm_amd.engine import Engine, SampleArgs

    res = []
    K, W, ctx = args.steps, min(args.warmup, 4), args.context
    for oc in OTHER_CONFIGS:
        family, prec = oc["workload"].rsplit("-", 1)
        qb = {"int4": 4, "int8": 8}.get(prec, 0)
        cfg = dict(SHAPES[family])
        if qb:
            cfg["quantization"] = {"group_size": 64, "bits": qb}
        B = oc["batch"]
        a = argparse.Namespace(**vars(args))
        a.batch, a.warmup, a.no_prefill_timing, a.workload, a.lora = B, W, True, oc["workload"], oc["lora"]
        chunk = max(16, (96 if qb == 4 else 256) - (B - 1) - 1)
        cap = ctx + 2 * (K + W) + 2 * chunk + 136
        t0 = time.perf_counter()
        eng = Engine(cfg, device=0, max_positions=max(cap, 2048), act_dtype="bfloat16")
        load_synthetic(eng, cfg, seed, qb, 0, 1, None)
        if oc["lora"]:
            apply_lora(eng, cfg, oc["lora"], seed)
        t_load = time.perf_counter() - t0
        rng = np.random.default_rng(seed + 23)
        prompts = rng.integers(0, cfg["vocab_size"], size=(B, ctx)).astype(np.int32)
        sample = SampleArgs(temp=1.0, top_p=0.9, seed=seed) if qb else SampleArgs(temp=0.0)
        entry = {"config": oc["config"], "workload": oc["workload"], "batch_per_gpu": B, "context": ctx, "steps": K, "warmup": W,
                 "load_seconds": round(t_load, 2)}
        for kvm in oc["kv_modes"]:
            leg = decode_leg(eng, cfg, a, kvm, prompts, sample, None, 1, cap)
            n = numbers(cfg, qb, B, K, W, ctx, leg, kvm, "gemv_gate_up", 1)
            entry["kv_" + ("model_dtype" if kvm == "model" else "float32")] = {
                "value": round(n["value"], 2), "unit": "tokens/s", "ms_per_step": round(n["ms_per_step"], 4),
                "step_bytes": int(n["step_bytes"]),
                "step_hbm_frac": round(n["step_bytes"] / (n["ms_per_step"] * 1e-3) / 8e12, 4),
                "roofline": {"bound": "hbm", "kernel": "gemv_gate_up", "achieved": round(n["achieved"], 1), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(n["achieved"] / 8000.0, 4), "traffic": None,
                             "bytes_per_launch": int(n["kern_bytes"]), "avg_launch_ms": round(n["avg_ms"], 5),
                             "launches": leg["n_launch"]},
            }
        if oc["mixed"]:
            a.chunk, a.kv_dtype, a.mode = chunk, "model", "mixed"
            m = mixed_leg(eng, cfg, a, sample, None, 1)
            f, u = m["fused"], m["unfused"]
            entry["mixed_prefill_decode"] = {
                "what": f"{B - 1} live sequences decoding at KV length ~{ctx} while the next request's prompt enters {chunk} tokens "
                        "per step in the same pass over the weights (mi_step_enqueue_mixed, block-paged KV)",
                "value": round(f["decode_tokens"] / f["elapsed"], 2), "unit": "tokens/s (decode rows)",
                "ms_per_step": round(f["elapsed"] / K * 1e3, 4),
                "prefill_tokens_per_sec_in_the_same_steps": round(f["prompt_tokens"] / f["elapsed"], 1),
                "fused_over_unfused": round(u["elapsed"] / f["elapsed"], 3),
            }
        eng.close()
        res.append(entry)
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="mistral-7b-bf16",
                    choices=["mistral-7b-bf16", "mistral-7b-int4", "mistral-7b-int8", "qwen3-14b-bf16", "qwen3-14b-int4",
                             "qwen3-14b-int8", "tiny-bf16"])
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--kv-dtype", default="float32", choices=["model", "float32"],
                    help="KV mode of the HEADLINE leg: float32 = the reference's PagedKVCache numerics (default); the other mode is "
                         "still timed and reported, under fast_mode (model-dtype KV) / reference_numerics (float32 KV)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill-timing", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the leg in the other KV mode")
    ap.add_argument("--lora", type=int, default=0, metavar="LAYERS",
                    help="apply rank-16 LoRA adapters (scale 10) to q_proj / v_proj of the last LAYERS blocks "
                         "(BASELINE config 5 uses 8; SURVEY §8d)")
    ap.add_argument("--greedy", action="store_true", help="greedy decode also for the int4 workloads (A/B of the sampler)")
    ap.add_argument("--profile-kernel", default="gemv_gate_up")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (A/B experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse N > 1 on a single-GPU box or on CPU")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--mode", default="decode", choices=["decode", "mixed"],
                    help="decode: the headline metric (K decode steps of a full batch).  mixed: BASELINE config 5's 'mixed "
                         "prefill + decode' -- every step decodes the live rows AND ingests a chunk of an arriving prompt in "
                         "the same pass over the weights (mi_step_enqueue_mixed, block-paged KV), timed next to the unfused "
                         "schedule (decode step, then the chunk on its own)")
    ap.add_argument("--chunk", type=int, default=0,
                    help="--mode mixed: prompt tokens ingested per step; 0 = what the live rows leave of 256 rows per step (int4: "
                         "96, the row limit of its weight-streaming kernel)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default workload only: skip the short legs of BASELINE configs 3 / 4 / 5 (`other_configs` of the line)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the multi-process protocol without an engine (launch, rendezvous, bucketed weight "
                         "broadcast, barrier / max-over-ranks timing) -- the only mode that runs without a GPU; "
                         "prints a line with value = null")
    return ap.parse_args(argv)


def decode_leg(engine, cfg, args, kv_dtype, prompts, sample, dist, world, cap):
    """Prefill (timed separately, after one untimed pass on a cache of its own) + W warm-up steps + EXACTLY K timed
    steps bracketed by barrier + device sync, MAX over ranks; then K more steps with the dominant kernel's launches
    bracketed by HIP events on the engine's own stream."""
    import torch

    B, K, W = args.batch, args.steps, args.warmup
    kv = engine.new_kv(B, capacity=cap, kv_dtype=kv_dtype)
    if not args.no_prefill_timing:
        kv_warm = engine.new_kv(B, capacity=cap, kv_dtype=kv_dtype)
        engine.step_wait(engine.step_enqueue(kv_warm, prompts, sample), B)
        kv_warm.close()
    engine.sync()
    t0 = time.perf_counter()
    engine.step_wait(engine.step_enqueue(kv, prompts, sample), B)
    t_prefill = time.perf_counter() - t0

    def run_steps(n):
        last = None
        for _ in range(n):
            last = engine.step_enqueue(kv, None, sample)               # tokens stay on the device
        return last

    last = run_steps(W)
    if last is not None:
        engine.step_wait(last, B)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    engine.sync()
    t0 = time.perf_counter()
    last = run_steps(K)
    engine.step_wait(last, B)
    engine.sync()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                  # this rank's K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    own_min = own_max = own
    if world > 1:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed, t_prefill, own, -own], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, t_prefill, own_max, own_min = float(tt[0].item()), float(tt[1].item()), float(tt[2].item()), -float(tt[3].item())
    engine.profile_select(args.profile_kernel)
    last = run_steps(K)
    engine.step_wait(last, B)
    n_launch, total_ms = engine.profile_read()
    engine.profile_select(None)
    kv.close()
    return dict(elapsed=elapsed, t_prefill=t_prefill, n_launch=n_launch, total_ms=total_ms, own_min=own_min, own_max=own_max)


def mixed_leg(engine, cfg, args, sample, dist, world):
    """Steady state of a serving batch: B slots, B - 1 of them decoding at context ~args.context, one slot taking in
    the prompt of the next request `--chunk` tokens per step; when a prompt is complete it starts decoding and the
    oldest sequence leaves (its row is reset for the next arrival).  Returns timings of K such steps, fused (one
    mi_step_enqueue_mixed per step) and unfused (a decode step of the live rows + the chunk as a step of its own)."""
    import numpy as np
    import torch

    B, K, W, ctx, C = args.batch, args.steps, args.warmup, args.context, args.chunk
    rng = np.random.default_rng(args.seed + 5)
    V = cfg["vocab_size"]
    out = {}
    for fused in (True, False):
        kv = engine.new_paged_kv(B, block_tokens=64, max_tokens_per_row=ctx + K + W + 2 * C + 64, kv_dtype=args.kv_dtype)
        # B - 1 live sequences with `ctx` tokens each (prefilled 8 rows at a time), row B - 1 is the arrival slot
        live = list(range(B - 1))
        last = np.zeros(B, dtype=np.int32)
        for i in range(0, B - 1, 8):
            rows = live[i:i + 8]
            p = rng.integers(0, V, size=(len(rows), ctx)).astype(np.int32)
            res = engine.step_wait(engine.step_enqueue_rows(kv, rows, p, sample), len(rows))
            last[rows] = res["tokens"]
        arriving, pos = B - 1, 0
        prompt = rng.integers(0, V, size=ctx).astype(np.int32)

        def one_step():
            nonlocal arriving, pos, prompt, live
            chunk = prompt[pos:pos + C]
            done = pos + len(chunk) >= len(prompt)
            if fused:
                t = engine.step_enqueue_mixed(kv, live + [arriving], [[int(last[r])] for r in live] + [chunk], [1] * len(live) + [int(done)], sample)
                res = engine.step_wait(t, len(live) + int(done))
                last[live] = res["tokens"][:len(live)]
                if done:
                    last[arriving] = res["tokens"][-1]
            else:
                t1 = engine.step_enqueue_rows(kv, live, [[int(last[r])] for r in live], sample)
                t2 = engine.step_enqueue_mixed(kv, [arriving], [chunk], [int(done)], sample)
                last[live] = engine.step_wait(t1, len(live))["tokens"]
                r2 = engine.step_wait(t2, int(done))
                if done:
                    last[arriving] = r2["tokens"][-1]
            pos += len(chunk)
            if done:                                    # the new sequence joins, the oldest leaves and frees its row
                oldest = live.pop(0)
                live.append(arriving)
                kv.reset_row(oldest)
                arriving, pos = oldest, 0
                prompt = rng.integers(0, V, size=ctx).astype(np.int32)
            return len(chunk)

        for _ in range(W):
            one_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); engine.sync()
        t0 = time.perf_counter()
        ptoks = sum(one_step() for _ in range(K))
        engine.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        out["fused" if fused else "unfused"] = dict(elapsed=el, prompt_tokens=ptoks, decode_tokens=(B - 1) * K, kv_stats=kv.stats())
        kv.close()
    return out


def main():
    args = parse_args()
    if os.environ.get("WORLD_SIZE") is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched a GPU yet, and
        # the ranks are fresh child processes (never an exec of a process that initialised HIP).
        from mlx_parallm_amd.distributed import self_launch

        sys.exit(self_launch([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus,
                             local_ranks=[0] * args.gpus if args.same_device else None))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` plainly, or under "
                         "torch.distributed.run --nproc-per-node N with the same N")
    if args.same_device:
        local_rank = 0
    have_gpu = torch.cuda.is_available() if not args.dry_run else (args.backend == "nccl" and torch.cuda.is_available())
    if not args.dry_run and not have_gpu:
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU backend (use --dry-run to rehearse the "
                         "multi-process protocol on a CPU-only host)")
    if have_gpu:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    family, prec = args.workload.rsplit("-", 1)
    quant_bits = {"int4": 4, "int8": 8}.get(prec, 0)
    cfg = dict(SHAPES[family])
    if quant_bits:
        cfg["quantization"] = {"group_size": 64, "bits": quant_bits}
    B, ctx, K, W = args.batch, args.context, args.steps, args.warmup
    cap = ctx + 2 * (K + W) + 8
    bstats = {}
    ranks_seen = world
    if world > 1:                                   # every rank reports in: "RCCL ranks seen" of the output line
        dev = "cuda" if args.backend == "nccl" else "cpu"
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    if args.dry_run:
        t0 = time.perf_counter()
        wdev = torch.device("cuda", local_rank) if have_gpu else torch.device("cpu")
        load_synthetic(None, cfg, args.seed, quant_bits, rank, world, dist, stats=bstats, device=wdev)
        t_load = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))                                     # stands in for the K steps: rank r takes longer
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if (args.backend == "nccl") else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        if rank == 0:
            print(json.dumps({"metric": "decode_tokens_per_sec", "value": None, "unit": "tokens/s", "n_gpus": world,
                              "steps": K, "warmup": W, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dry_run": True, "data": "synthetic", "ranks_seen": ranks_seen,
                              "backend": args.backend, "load_seconds": round(t_load, 2),
                              "rehearsal_barrier_seconds": round(elapsed, 4), **bstats,
                              "config": {"workload": f"{family} shape ({args.workload}), DRY RUN: no engine, no decode",
                                         "batch_per_gpu": B, "global_batch": B * world,
                                         "parallelism": f"dp{world} (batch-sharded replicas, no collective in the decode step)"}}),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from mlx_parallm_amd.engine import Engine, SampleArgs

    if args.mode == "mixed":
        if args.chunk <= 0:
            args.chunk = max(16, (96 if quant_bits == 4 else 256) - (B - 1) - 1)
        cap = ctx + K + W + 2 * args.chunk + 128
    engine = Engine(cfg, device=local_rank, max_positions=max(cap, 2048), act_dtype="bfloat16")
    for kv_ in args.opt:
        k_, v_ = kv_.split("=")
        engine.set_option(k_, int(v_))
    t0 = time.perf_counter()
    load_synthetic(engine, cfg, args.seed, quant_bits, rank, world, dist, stats=bstats)
    if args.lora:
        apply_lora(engine, cfg, args.lora, args.seed)
    t_load = time.perf_counter() - t0

    rng = np.random.default_rng(args.seed + 17 * rank)                 # each rank decodes its own shard
    prompts = rng.integers(0, cfg["vocab_size"], size=(B, ctx)).astype(np.int32)
    # greedy for the bf16 configurations (BASELINE configs 2, 4); top-p 0.9 at temperature 1 with
    # log-probabilities for the int4 configuration (config 3).  `greedy` names the per-step sampler either way.
    greedy = SampleArgs(temp=1.0, top_p=0.9, seed=args.seed) if (quant_bits and not args.greedy) else SampleArgs(temp=0.0)

    if args.mode == "mixed":
        m = mixed_leg(engine, cfg, args, greedy, dist, world)
        if rank == 0:
            f, u = m["fused"], m["unfused"]
            w_bytes, _ = streamed_weight_bytes(cfg, quant_bits)
            print(json.dumps({
                "metric": "decode_tokens_per_sec", "value": round(world * f["decode_tokens"] / f["elapsed"], 2), "unit": "tokens/s",
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(f["elapsed"] / K * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16" if not quant_bits else f"int{quant_bits}-g64 weights, bf16 activations",
                "data": "synthetic token ids; random-init weights N(0,0.02^2)",
                "config": {"workload": f"{family} shape ({args.workload}), MIXED prefill + decode: {B - 1} live sequences/GPU decoding at KV length ~{ctx} "
                                       f"while the prompt of the next request enters the cache {args.chunk} tokens per step in the SAME pass "
                                       f"over the weights (mi_step_enqueue_mixed, block-paged KV, 64-token blocks)"
                                       + (f", rank-16 LoRA on q/v of the last {args.lora} layers" if args.lora else ""),
                           "batch_per_gpu": B, "global_batch": B * world, "context": ctx, "chunk_tokens": args.chunk,
                           "kv_dtype": args.kv_dtype, "parallelism": f"dp{world} (batch-sharded replicas, no collective in the step)"},
                "prefill_tokens_per_sec_in_the_same_steps": round(world * f["prompt_tokens"] / f["elapsed"], 1),
                "unfused_schedule": {"what": "the same K steps as a decode step of the live rows followed by the chunk as a step of its own (two passes over the weights)",
                                     "decode_tokens_per_sec": round(world * u["decode_tokens"] / u["elapsed"], 2),
                                     "ms_per_step": round(u["elapsed"] / K * 1e3, 4)},
                "fused_over_unfused": round(u["elapsed"] / f["elapsed"], 3),
                "weight_bytes_per_step": int(w_bytes), "kv_arena": f["kv_stats"], "load_seconds": round(t_load, 2), "ranks_seen": ranks_seen,
            }), flush=True)
        engine.close()
        if world > 1:
            dist.destroy_process_group()
        return

    head = decode_leg(engine, cfg, args, args.kv_dtype, prompts, greedy, dist, world, cap)
    other_mode = "float32" if args.kv_dtype == "model" else "model"
    other = None if args.no_second_leg else decode_leg(engine, cfg, args, other_mode, prompts, greedy, dist, world, cap)

    w_bytes, n_params = streamed_weight_bytes(cfg, quant_bits)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    kern = args.profile_kernel

    def leg_numbers(leg, kv_dtype):
        return numbers(cfg, quant_bits, B, K, W, ctx, leg, kv_dtype, kern, world)

    def pmc_traffic(kv_dtype):
        """HBM bytes per launch of the dominant kernel.  NOT measured in this run: read from the committed PMC passes
        of this same command (profiles/, separate rocprofv3 --pmc runs, as the guide prescribes): 2 x FETCH_SIZE
        (gfx950 counts 64 B per 128-B request on wide coalesced streams, MI355X_MICROARCH.md HBM section) + WRITE_SIZE,
        both reported in KiB.  null for any other workload / kernel / batch than the one that was profiled."""
        if args.workload != "mistral-7b-bf16" or kern != "gemv_gate_up" or B != 8 or args.lora:
            return None, None
        # model-dtype KV: the M <= 8 kernel with the fused RMSNorm; float32 KV: the split-K kernel on float32 activations
        want = "gemv_mfma_gu8_kernel<bf16,MB=8>" if kv_dtype == "model" else "skinny_kernel<bf16,0,1,true,true>"
        for tag in (("round4_bf16kv", "round3_bf16kv", "round2", "round1") if kv_dtype == "model" else ("round4_f32kv", "round3_f32kv")):
            vals = {}
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                f = ROOT / "profiles" / f"{tag}_pmc_{ctr}.csv"
                if not f.exists():
                    break
                for line in f.read_text().splitlines():
                    if line.startswith(want + ","):
                        vals[ctr] = float(line.rsplit(",", 1)[1]) * 1024.0
            if len(vals) == 2:
                return int(2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]), \
                    f"static reference from profiles/{tag}_pmc_{{FETCH,WRITE}}_SIZE.csv (separate rocprofv3 --pmc passes of this command; 2xFETCH_SIZE+WRITE_SIZE), not measured in this run"
        return None, None

    def roofline_of(num, kv_dtype):
        traffic, src = pmc_traffic(kv_dtype)
        return {
            "bound": "hbm", "kernel": kern, "achieved": round(num["achieved"], 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(num["achieved"] / 8000.0, 4),
            "frac_of_achievable": round(num["achieved"] / 6290.0, 4),   # of the 6.29 TB/s a streaming kernel reaches (MI355X_MICROARCH.md, HBM)
            "traffic": traffic, "traffic_source": src,
            "bytes_per_launch": int(num["kern_bytes"]), "avg_launch_ms": round(num["avg_ms"], 5),
        }

    if rank == 0:
        hn = leg_numbers(head, args.kv_dtype)
        kv_names = {"model": "bf16 (BatchedKVCache semantics)", "float32": "float32 (PagedKVCache semantics, the reference's default)"}
        out = {
            "metric": "decode_tokens_per_sec", "value": round(hn["value"], 2), "unit": "tokens/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(hn["ms_per_step"], 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not quant_bits else f"int{quant_bits}-g64 weights, bf16 activations",
            "data": "synthetic token ids; random-init weights N(0,0.02^2)",
            "config": {
                "workload": f"{family} shape ({args.workload}), batch {B}/GPU, "
                            + ("top-p 0.9 / T=1 sampling with logprobs" if (quant_bits and not args.greedy) else "greedy decode") + f" from KV length {ctx}"
                            + (f", rank-16 LoRA on q/v of the last {args.lora} layers" if args.lora else ""),
                "batch_per_gpu": B, "global_batch": B * world, "context": ctx,
                "kv_dtype": kv_names[args.kv_dtype],
                "numerics": ("the reference's (PagedKVCache float32; oracle parity asserted in this mode)" if args.kv_dtype == "float32"
                             else "fast mode (16-bit KV and activations); the reference-numerics leg is under reference_numerics"),
                "parallelism": f"dp{world} (batch-sharded replicas, no collective in the decode step)",
            },
            "roofline": dict(roofline_of(hn, args.kv_dtype), launches=head["n_launch"]),
            "step_bytes": int(hn["step_bytes"]),
            "step_hbm_frac": round(hn["step_bytes"] / (hn["ms_per_step"] * 1e-3) / 8e12, 4),
            "step_hbm_frac_of_achievable": round(hn["step_bytes"] / (hn["ms_per_step"] * 1e-3) / 6.29e12, 4),
            "prefill_tokens_per_sec": round(hn["prefill"], 1),
            "prefill_roofline": {"bound": "mfma", "achieved": round(hn["pf_tflops"], 1), "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": round(hn["pf_tflops"] / 2500.0, 4), "flops_per_gpu": hn["pf_flops"],
                                 "what": "whole prefill call (B x context tokens, all kernels, host-timed) against the dense bf16 MFMA peak"},
            "load_seconds": round(t_load, 2),
            "ranks_seen": ranks_seen,
            # each rank's own K steps, timed before it waits at the closing barrier: a slow rank shows here, not only in the max
            "per_rank_tokens_per_sec": {"min": round(hn["rank_min"], 2), "max": round(hn["rank_max"], 2)},
        }
        out.update(bstats)
        if other is not None:
            on = leg_numbers(other, other_mode)
            out["reference_numerics" if other_mode == "float32" else "fast_mode"] = {
                "what": f"the same workload and timing protocol with KV {kv_names[other_mode]}"
                        + ("; ids bit-exact / logprobs <= 1e-3 vs the oracle are asserted in THIS mode (tests/test_gpu_engine.py, test_gpu_golden_wide.py)"
                           if other_mode == "float32" else
                           "; an extra of this build, NOT the reference's arithmetic on this path (every activation and the caches are 16-bit: "
                           "parity there is 'same id wherever the oracle's margin exceeds a few ulps of a 16-bit logit', DESIGN 2)"),
                "value": round(on["value"], 2), "unit": "tokens/s", "ms_per_step": round(on["ms_per_step"], 4),
                "prefill_tokens_per_sec": round(on["prefill"], 1),
                "prefill_mfma_frac": round(on["pf_tflops"] / 2500.0, 4),
                "step_bytes": int(on["step_bytes"]),
                "step_hbm_frac": round(on["step_bytes"] / (on["ms_per_step"] * 1e-3) / 8e12, 4),
                "roofline": dict(roofline_of(on, other_mode), launches=other["n_launch"]),
            }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c_ref

            nl_s, steps_s = 2, 6
            sec, head_sec, nth = c_ref.bench_decode(H, nh, nkv, D, I, cfg["vocab_size"], nl_s, B, ctx, steps_s)
            per_step = (sec - head_sec) / steps_s / nl_s * cfg["num_hidden_layers"] + head_sec / steps_s
            out["cpu_baseline"] = {
                "value": round(B / per_step, 3), "unit": "tokens/s", "cores": nth, "kind": "port",
                "sample": (f"oracle/c restatement, {nl_s} of {cfg['num_hidden_layers']} decoder blocks + lm_head, "
                           f"{steps_s} decode steps at KV length {ctx}, batch {B}, bf16 weights; block time scaled "
                           f"x{cfg['num_hidden_layers']}/{nl_s}; not MLX (unavailable offline)"),
            }
        if world == 1 and not args.no_other_configs and args.workload == "mistral-7b-bf16" and B == 8 and not args.lora:
            engine.close()
            engine = None
            out["other_configs"] = other_config_legs(args, args.seed)
        print(json.dumps(out), flush=True)
    if engine is not None:
        engine.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
