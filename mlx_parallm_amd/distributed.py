"""Batch-sharded data parallelism over the GPUs of one node.

The reference has no multi-device support at all (no ``mx.distributed`` / NCCL call anywhere), so
nothing here mirrors reference code; it is the north_star's scaling scheme: one process per GPU
(``torch.distributed``, backend "nccl" = RCCL over xGMI), every rank holds a FULL weight replica
and its own KV cache, the requests of a batch are split contiguously across ranks, and the decode
step contains NO collective.  The only exchanges are

  * one bucketed broadcast of the weights from the loading rank at start-up, and
  * a gather of the (tiny) generated token lists / strings at the end of a call.

xGMI is point-to-point (7 links x ~153 GB/s per GPU) and a ring broadcast is per-link bound, so
weights go out in a few LARGE buckets rather than tensor by tensor.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: Optional[str] = None):
    """Initialise the default process group when WORLD_SIZE > 1.  backend None -> "nccl" (RCCL) if a
    GPU is visible, else "gloo".  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist

    rank, world, local_rank = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) of rank's share; the first n % world ranks get one item more."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    s, e = shard_range(len(items), rank, world)
    return list(items[s:e])


def broadcast_tensors(tensors: Dict[str, "object"], src: int = 0, bucket_bytes: int = 1 << 30) -> None:
    """In-place broadcast of same-shaped tensors that every rank has already allocated: tensors are
    packed by dtype into flat buckets of up to ``bucket_bytes`` and each bucket is one collective."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    by_dtype: Dict[object, List[Tuple[str, object]]] = {}
    for name in sorted(tensors):
        t = tensors[name]
        by_dtype.setdefault((t.dtype, t.device), []).append((name, t))
    for (dtype, device), items in by_dtype.items():
        bucket: List[Tuple[str, object]] = []
        size = 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.empty(sum(t.numel() for _, t in bucket), dtype=dtype, device=device)
            if dist.get_rank() == src:
                off = 0
                for _, t in bucket:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            dist.broadcast(flat, src=src)
            off = 0
            for _, t in bucket:
                if dist.get_rank() != src:
                    t.copy_(flat[off:off + t.numel()].reshape(t.shape))
                off += t.numel()
            bucket, size = [], 0

        for name, t in items:
            nbytes = t.numel() * t.element_size()
            if bucket and size + nbytes > bucket_bytes:
                flush()
            bucket.append((name, t))
            size += nbytes
        flush()


def gather_in_rank_order(local_items: Sequence, dst: int = 0) -> Optional[List]:
    """Concatenation of every rank's list in rank order on ``dst`` (None elsewhere)."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_items)
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(list(local_items), out, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged: List = []
    for part in out:
        merged.extend(part)
    return merged


def sharded_batch_generate(generate_fn: Callable[[List[str]], List[str]], prompts: Sequence[str],
                           dst: int = 0) -> Optional[List[str]]:
    """Data-parallel ``batch_generate`` (utils.py:473): each rank runs ``generate_fn`` on its
    contiguous share of ``prompts``; rank ``dst`` gets all responses in the original order."""
    import torch.distributed as dist

    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    mine = shard_list(prompts, rank, world)
    local = generate_fn(mine) if mine else []
    if len(local) != len(mine):
        raise RuntimeError("generate_fn must return one response per prompt")
    return gather_in_rank_order(local, dst=dst)
