"""``python -m mlx_parallm_amd.dp_generate --model-path DIR --prompts-file F --gpus N``: ``batch_generate`` over the GPUs
of one node (the north_star's scaling scheme -- the reference itself has no multi-device path).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Rank 0 reads the checkpoint and
replicates it with a few large bucketed broadcasts; every rank then runs ``utils.batch_generate`` on its contiguous
share of the prompts with its own engine and KV cache -- no collective inside generation -- and rank 0 gathers the
responses in prompt order and prints them as one JSON object.  Typed as a plain command it launches its own rank
processes (``distributed.self_launch``); under ``torch.distributed.run`` it uses the ranks it is given.

``--prompts-file``: a JSON list of strings, or plain text with one prompt per line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path
from typing import List, Optional, Sequence


def read_prompts(path: str) -> List[str]:
    text = Path(path).read_text()
    try:
        data = json.loads(text)
        if isinstance(data, list) and all(isinstance(p, str) for p in data):
            return data
    except json.JSONDecodeError:
        pass
    return [ln for ln in text.splitlines() if ln.strip()]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="mlx_parallm_amd.dp_generate")
    p.add_argument("--model-path", required=True)
    p.add_argument("--adapter-path", default=None)
    p.add_argument("--prompts-file", required=True)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--max-tokens", type=int, default=100)
    p.add_argument("--temp", type=float, default=0.0)
    p.add_argument("--top-p", type=float, default=1.0)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no-format", action="store_true", help="feed the prompts as they are (no chat template)")
    p.add_argument("--kv-dtype", default=None, choices=["model", "float32"])
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    p.add_argument("--same-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0")
    return p


def run_rank(ns) -> Optional[dict]:
    import torch

    from . import utils
    from .distributed import init_distributed, sharded_batch_generate

    if ns.same_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = init_distributed(backend=ns.backend if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if ns.kv_dtype:
        utils.DEFAULT_KV_DTYPE = ns.kv_dtype
    prompts = read_prompts(ns.prompts_file)
    stats: dict = {}
    t0 = time.perf_counter()
    torch.cuda.set_device(local_rank)
    model = utils.load_model(Path(ns.model_path), device=local_rank, weights_from_rank=0 if world > 1 else None,
                             replicate_stats=stats)
    if ns.adapter_path:
        utils.load_adapters(model, ns.adapter_path)
    tok = utils.load_tokenizer(Path(ns.model_path))
    t_load = time.perf_counter() - t0

    def gen(mine: List[str]) -> List[str]:
        return utils.batch_generate(model, tok, mine, max_tokens=ns.max_tokens, format_prompts=not ns.no_format,
                                    temp=ns.temp, top_p=ns.top_p, seed=ns.seed + rank)

    t0 = time.perf_counter()
    responses = sharded_batch_generate(gen, prompts)
    t_gen = time.perf_counter() - t0
    model.engine.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return None
    return {"responses": responses, "n_prompts": len(prompts), "n_gpus": world, "load_seconds": round(t_load, 3),
            "generate_seconds": round(t_gen, 3), **stats}


def main(argv: Optional[Sequence[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    ns = build_parser().parse_args(argv)
    if os.environ.get("WORLD_SIZE") is None and ns.gpus > 1:
        from .distributed import self_launch          # (this process has not touched a GPU)

        return self_launch([sys.executable, "-m", "mlx_parallm_amd.dp_generate"] + argv, ns.gpus,
                           local_ranks=[0] * ns.gpus if ns.same_device else None)
    out = run_rank(ns)
    if out is not None:
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
