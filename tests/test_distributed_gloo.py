"""N > 1 path on CPU: two processes over gloo exercise the batch sharding, the bucketed weight
broadcast and the rank-ordered gather.  No GPU, no RCCL here; on the GPU box the same
code runs with backend "nccl"."""
import os
import socket

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
