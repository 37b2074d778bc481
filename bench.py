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

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused gate|up SwiGLU
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


def load_synthetic(engine, cfg, seed, quant_bits, rank, world, dist):
    """Random-init weights N(0, 0.02^2) (norm weights = 1) generated on rank 0's GPU, replicated with
    RCCL broadcasts, optionally MLX-affine quantised (group 64), handed to the engine."""
    import torch

    from mlx_parallm_amd.quant import quantize

    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev)
    nbytes = 0
    for idx, (name, shape, kind) in enumerate(tensor_specs(cfg)):
        if kind == "norm":
            t = torch.ones(shape, dtype=torch.bfloat16, device=dev)
        else:
            t = torch.empty(shape, dtype=torch.bfloat16, device=dev)
            if rank == 0:
                gen.manual_seed(seed * 1000003 + idx)
                t.copy_(torch.randn(shape, generator=gen, device=dev, dtype=torch.float32) * 0.02)
        if world > 1:
            dist.broadcast(t, src=0)
        if kind == "mat" and quant_bits:
            packed, scales, biases = quantize(t, 64, quant_bits)
            base = name[: -len(".weight")]
            engine.set_tensor(base + ".weight", packed)
            engine.set_tensor(base + ".scales", scales)
            engine.set_tensor(base + ".biases", biases)
            nbytes += packed.numel() * 4 + scales.numel() * 4
        else:
            engine.set_tensor(name, t)
            nbytes += t.numel() * 2
        del t
    torch.cuda.synchronize()
    engine.finalize()
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="mistral-7b-bf16",
                    choices=["mistral-7b-bf16", "mistral-7b-int4", "mistral-7b-int8", "qwen3-14b-bf16", "qwen3-14b-int4",
                             "qwen3-14b-int8", "tiny-bf16"])
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--kv-dtype", default="model", choices=["model", "float32"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill-timing", action="store_true")
    ap.add_argument("--lora", type=int, default=0, metavar="LAYERS",
                    help="apply rank-16 LoRA adapters (scale 10) to q_proj / v_proj of the last LAYERS blocks "
                         "(BASELINE config 5 uses 8; SURVEY §8d)")
    ap.add_argument("--greedy", action="store_true", help="greedy decode also for the int4 workloads (A/B of the sampler)")
    ap.add_argument("--profile-kernel", default="gemv_gate_up")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (A/B experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse N > 1 on a single-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    from mlx_parallm_amd.engine import Engine, SampleArgs

    family, prec = args.workload.rsplit("-", 1)
    quant_bits = {"int4": 4, "int8": 8}.get(prec, 0)
    cfg = dict(SHAPES[family])
    if quant_bits:
        cfg["quantization"] = {"group_size": 64, "bits": quant_bits}
    B, ctx, K, W = args.batch, args.context, args.steps, args.warmup
    cap = ctx + 2 * (K + W) + 8
    engine = Engine(cfg, device=local_rank, max_positions=max(cap, 2048), act_dtype="bfloat16")
    for kv_ in args.opt:
        k_, v_ = kv_.split("=")
        engine.set_option(k_, int(v_))
    t0 = time.perf_counter()
    load_synthetic(engine, cfg, args.seed, quant_bits, rank, world, dist)
    if args.lora:
        # A ~ U(-1/sqrt(K), 1/sqrt(K)), B ~ N(0, 0.01^2), rank 16, scale 10 (lora_init.py:68-72 defaults)
        H, nh, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
        D = cfg.get("head_dim") or H // nh
        g = torch.Generator().manual_seed(args.seed + 99)
        for li in range(cfg["num_hidden_layers"] - args.lora, cfg["num_hidden_layers"]):
            for key, n in (("self_attn.q_proj", nh * D), ("self_attn.v_proj", nkv * D)):
                a = (torch.rand((H, 16), generator=g) * 2 - 1) / (H ** 0.5)
                b = torch.randn((16, n), generator=g) * 0.01
                engine.set_lora(li, key, a, b, 10.0)
    t_load = time.perf_counter() - t0

    rng = np.random.default_rng(args.seed + 17 * rank)                 # each rank decodes its own shard
    prompts = rng.integers(0, cfg["vocab_size"], size=(B, ctx)).astype(np.int32)
    kv = engine.new_kv(B, capacity=cap, kv_dtype=args.kv_dtype)
    # greedy for the bf16 configurations (BASELINE configs 2, 4); top-p 0.9 at temperature 1 with
    # log-probabilities for the int4 configuration (config 3).  `greedy` names the per-step sampler either way.
    greedy = SampleArgs(temp=1.0, top_p=0.9, seed=args.seed) if (quant_bits and not args.greedy) else SampleArgs(temp=0.0)

    # ---- prefill (timed separately: "prefill tok/s"); one untimed pass first, on a cache of its own, so that the
    # timed pass does not pay the first-launch costs of the prefill kernels (code object load, LDS attributes)
    if not args.no_prefill_timing:
        kv_warm = engine.new_kv(B, capacity=cap, kv_dtype=args.kv_dtype)
        engine.step_wait(engine.step_enqueue(kv_warm, prompts, greedy), B)
        kv_warm.close()
    engine.sync()
    t0 = time.perf_counter()
    ticket = engine.step_enqueue(kv, prompts, greedy)
    engine.step_wait(ticket, B)
    t_prefill = time.perf_counter() - t0

    def run_steps(n):
        last = None
        for _ in range(n):
            last = engine.step_enqueue(kv, None, greedy)               # tokens stay on the device
        return last

    last = run_steps(W)
    if last is not None:
        engine.step_wait(last, B)
    # ---- timed region: exactly K steps, barrier + device sync on both sides, max over ranks
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    engine.sync()
    t0 = time.perf_counter()
    last = run_steps(K)
    engine.step_wait(last, B)
    engine.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- dominant kernel: mean launch duration from HIP events on the engine's stream
    kern = args.profile_kernel
    engine.profile_select(kern)
    last = run_steps(K)
    engine.step_wait(last, B)
    n_launch, total_ms = engine.profile_read()
    engine.profile_select(None)

    w_bytes, n_params = streamed_weight_bytes(cfg, quant_bits)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    D = cfg.get("head_dim") or H // nh
    kvb = 2 if args.kv_dtype == "model" else 4
    S_mid = ctx + W + K // 2
    step_bytes = (w_bytes + B * cfg["num_hidden_layers"] * 2 * nkv * D * S_mid * kvb
                  + B * cfg["num_hidden_layers"] * 2 * nkv * D * kvb + B * cfg["vocab_size"] * 4)
    bpp = 2.0 if not quant_bits else quant_bits / 8.0 + 4.0 / 64.0
    kern_bytes = {
        "gemv_gate_up": 2 * I * H * bpp + B * H * 2 + B * I * 2,
        "gemv_down": I * H * bpp + B * I * 2 + 2 * B * H * 2,
        "gemv_qkv": (nh + 2 * nkv) * D * H * bpp + B * H * 2 + B * (nh + 2 * nkv) * D * 2,
        "gemv_o": H * nh * D * bpp + B * nh * D * 2 + 2 * B * H * 2,
        "gemv_head": cfg["vocab_size"] * H * bpp + B * H * 2 + B * cfg["vocab_size"] * 4,
    }.get(kern, 0.0)

    def pmc_traffic():
        """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/, own
        rocprofv3 --pmc runs of this same command): 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request
        on wide coalesced streams, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both reported in KiB."""
        if args.workload != "mistral-7b-bf16" or kern != "gemv_gate_up" or B != 8:
            return None
        want = "gemv_mfma_kernel<bf16,dense,MB=8,swiglu>"
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = ROOT / "profiles" / f"round1_pmc_{ctr}.csv"
            if not f.exists():
                return None
            for line in f.read_text().splitlines():
                if line.startswith(want + ","):
                    vals[ctr] = float(line.rsplit(",", 1)[1]) * 1024.0
        if len(vals) != 2:
            return None
        return int(2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"])

    if rank == 0:
        ms_per_step = elapsed / K * 1e3
        value = world * B * K / elapsed
        avg_ms = total_ms / max(n_launch, 1)
        achieved = kern_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "decode_tokens_per_sec", "value": round(value, 2), "unit": "tokens/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not quant_bits else f"int{quant_bits}-g64 weights, bf16 activations",
            "data": "synthetic token ids; random-init weights N(0,0.02^2)",
            "config": {
                "workload": f"{family} shape ({args.workload}), batch {B}/GPU, "
                            + ("top-p 0.9 / T=1 sampling with logprobs" if quant_bits else "greedy decode") + f" from KV length {ctx}"
                            + (f", rank-16 LoRA on q/v of the last {args.lora} layers" if args.lora else ""),
                "batch_per_gpu": B, "global_batch": B * world, "context": ctx,
                "kv_dtype": "bf16" if args.kv_dtype == "model" else "float32 (PagedKVCache quirk mode)",
                "parallelism": f"dp{world} (batch-sharded replicas, no collective in the decode step)",
            },
            "roofline": {
                "bound": "hbm", "kernel": kern, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4),
                "frac_of_achievable": round(achieved / 6290.0, 4),      # of the 6.29 TB/s a streaming kernel reaches (MI355X_MICROARCH.md, HBM)
                "traffic": pmc_traffic(),
                "traffic_source": "profiles/round1_pmc_{FETCH,WRITE}_SIZE.csv (separate rocprofv3 --pmc passes; 2xFETCH_SIZE+WRITE_SIZE)",
                "bytes_per_launch": int(kern_bytes), "avg_launch_ms": round(avg_ms, 5), "launches": n_launch,
            },
            "step_bytes": int(step_bytes),
            "step_hbm_frac": round(step_bytes / (ms_per_step * 1e-3) / 8e12, 4),
            "step_hbm_frac_of_achievable": round(step_bytes / (ms_per_step * 1e-3) / 6.29e12, 4),
            "prefill_tokens_per_sec": round(world * B * ctx / t_prefill, 1),
            "load_seconds": round(t_load, 2),
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
        print(json.dumps(out), flush=True)
    kv.close()
    engine.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
