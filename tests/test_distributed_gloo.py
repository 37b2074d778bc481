"""N > 1 path on CPU: two processes over gloo exercise the batch sharding, the bucketed weight
broadcast and the rank-ordered gather.  No GPU, no RCCL here; on the GPU box the same
code runs with backend "nccl"."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from mlx_parallm_amd import distributed as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 32, 64, 65):
        for world in (1, 2, 3, 4, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert D.shard_range(32, 1, 4) == (8, 16) and D.shard_range(64, 7, 8) == (56, 64)     # BASELINE cfg 4 / 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # weights: rank 0 holds the real values, the others zeros
    g = torch.Generator().manual_seed(0)
    ref = {f"w{i}": torch.randn((17, 5 + i), generator=g) for i in range(6)}
    ref["ids"] = torch.arange(11, dtype=torch.int32)
    mine = {k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in ref.items()}
    D.broadcast_tensors(mine, src=0, bucket_bytes=512)               # small buckets -> several collectives
    for k in ref:
        assert torch.equal(mine[k], ref[k]), k
    # data-parallel generate: each rank sees only its contiguous share
    prompts = [f"p{i}" for i in range(7)]
    seen = []

    def gen(batch):
        seen.extend(batch)
        return [f"{p}@{rank}" for p in batch]

    res = D.sharded_batch_generate(gen, prompts)
    s, e = D.shard_range(len(prompts), rank, world)
    assert seen == prompts[s:e]
    if rank == 0:
        assert res == ["p0@0", "p1@0", "p2@0", "p3@0", "p4@1", "p5@1", "p6@1"]
    else:
        assert res is None
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    torch.distributed.destroy_process_group()


def test_two_process_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _replicate_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_distributed(backend="gloo")
    g = torch.Generator().manual_seed(3)
    ckpt = {"a.weight": torch.randn((33, 7), generator=g).to(torch.bfloat16),
            "b.weight": torch.randint(-2 ** 31, 2 ** 31 - 1, (5, 9), generator=g, dtype=torch.int64).to(torch.int32).view(torch.uint32),
            "b.scales": torch.randn((5, 3), generator=g).to(torch.bfloat16),
            "n.weight": torch.randn((11,), generator=g)}
    stats = {}
    got = D.replicate_checkpoint(ckpt if rank == 0 else None, src=0, device=torch.device("cpu"), bucket_bytes=256,
                                 stats=stats)
    assert sorted(got) == sorted(ckpt)
    for k, t in ckpt.items():
        want = t.view(torch.int32) if t.dtype == torch.uint32 else t
        assert got[k].shape == t.shape and torch.equal(got[k], want), k
    assert stats["broadcast_buckets"] >= 3 and stats["broadcast_bytes"] >= sum(t.numel() * t.element_size() for t in ckpt.values())
    open(os.path.join(out_dir, f"rep{rank}"), "w").write("ok")
    torch.distributed.destroy_process_group()


def test_checkpoint_replication_over_gloo(tmp_path):
    """utils.load_model(weights_from_rank=0): the reading rank's tensors arrive on every rank by bucketed broadcast."""
    mp.spawn(_replicate_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"rep{r}").exists() for r in range(2))


def test_plan_buckets():
    specs = [("a", (1000,)), ("b", (10, 100)), ("big", (5000,)), ("c", (3,))]
    buckets = D.plan_buckets(specs, elem_size=2, bucket_bytes=4096)
    assert [[e[0] for e in b] for b in buckets] == [["a", "b"], ["big"], ["c"]]       # an oversized tensor gets its own
    for b in buckets:
        assert all(off % 128 == 0 for _, _, off, _ in b)
        assert D.bucket_numel(b) >= b[-1][2] + b[-1][3]


def test_bench_self_launches_two_ranks_as_a_plain_command():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts the two ranks itself (before any GPU
    call), they rendezvous over gloo on 127.0.0.1, replicate the synthetic weights in buckets, run the barrier /
    max-over-ranks protocol, and rank 0's single JSON line comes back through the parent.  --dry-run: no engine (this
    host has no GPU); the same path with the engine is tests/test_gpu_bench_contract.py."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "tiny-bf16",
                          "--steps", "3", "--warmup", "1", "--dry-run"], cwd=str(root), env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["dry_run"] is True and j["value"] is None
    tiny = 151936 * 64 + 8 * (4 * 64 * 64 + 3 * 128 * 64 + 2 * 64) + 64              # tied embedding: no lm_head
    assert j["broadcast_buckets"] == 1 and j["broadcast_bytes"] >= 2 * tiny
    assert j["rehearsal_barrier_seconds"] >= 0.02                                     # the slower rank (rank 1) sets the time
    # a failing rank takes the launch down instead of leaving the others at a collective
    bad = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "tiny-bf16"],
                         cwd=str(root), env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "needs an MI355X" in bad.stderr


def test_self_launch_refuses_under_a_gpu_tool_preload(monkeypatch, capsys):
    """rocprofv3 preloads a library that initialises the GPU before main(); starting rank processes from such a process is
    the exec this pool forbids.  self_launch must refuse (exit code 2, a message) instead of forking (round-2 advisory)."""
    from mlx_parallm_amd import distributed

    assert distributed.gpu_preload_in_environment({}) == ""
    assert distributed.gpu_preload_in_environment({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    rc = distributed.self_launch([sys.executable, "-c", "raise SystemExit(0)"], 2)
    assert rc == 2 and "refusing" in capsys.readouterr().err
