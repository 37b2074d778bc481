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


# ------------------------------------------------------------------------------------------------------------------
# weight replication without staging copies: every rank carves the SAME flat buckets, the loading rank fills them,
# one collective per bucket, and the tensors handed to mi_engine_set_tensor are views into the bucket
def plan_buckets(specs: Iterable[Tuple[str, Tuple[int, ...]]], elem_size: int, bucket_bytes: int = 1 << 30):
    """specs: (name, shape) in load order -> list of buckets, each a list of (name, shape, offset_elems, numel);
    a bucket closes when the next tensor would push it past ``bucket_bytes`` (a larger tensor gets a bucket of its own).
    Offsets are multiples of 128 elements so that every view is 256-byte aligned for 16-bit types."""
    buckets, cur, off = [], [], 0
    for name, shape in specs:
        n = 1
        for d in shape:
            n *= int(d)
        if cur and (off + n) * elem_size > bucket_bytes:
            buckets.append(cur)
            cur, off = [], 0
        cur.append((name, tuple(int(d) for d in shape), off, n))
        off += (n + 127) // 128 * 128
    if cur:
        buckets.append(cur)
    return buckets


def bucket_numel(bucket) -> int:
    name, shape, off, n = bucket[-1]
    return off + (n + 127) // 128 * 128


def broadcast_bucket(flat, src: int = 0) -> Tuple[float, int]:
    """One collective for one flat bucket (RCCL over xGMI under "nccl").  -> (seconds, bytes); (0, 0) single process."""
    import time

    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0, 0
    if flat.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(flat, src=src)
    if flat.is_cuda:
        torch.cuda.synchronize()
    return time.perf_counter() - t0, flat.numel() * flat.element_size()


# ------------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def gpu_preload_in_environment(env: Optional[Dict[str, str]] = None) -> str:
    """Non-empty (the offending setting) when a profiler / tool library that initialises the GPU is preloaded into this
    process: rocprofv3 sets LD_PRELOAD to its tool library and ROCP_TOOL_LIBRARIES / ROCPROFILER_* for it."""
    env = os.environ if env is None else env
    pre = env.get("LD_PRELOAD", "")
    if any(k in pre for k in ("rocprof", "roctracer", "rocprofiler")):
        return "LD_PRELOAD=" + pre
    for key in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR", "ROCP_TOOL_LIB"):
        if env.get(key):
            return f"{key}={env[key]}"
    # HSA_TOOLS_LIB is loaded by hsa_init, which a launcher that never touches the GPU never calls: a debug agent or
    # tracer named there has not initialised anything in THIS process -- worth a note, not a refusal
    if env.get("HSA_TOOLS_LIB"):
        import sys

        print(f"note: HSA_TOOLS_LIB={env['HSA_TOOLS_LIB']} is set; the rank processes will load it at hsa_init", file=sys.stderr)
    return ""


def self_launch(argv: Sequence[str], nprocs: int, *, local_ranks: Optional[Sequence[int]] = None,
                extra_env: Optional[Dict[str, str]] = None) -> int:
    """Run ``argv`` (a full command line) as ``nprocs`` rank processes of one node and relay rank 0's stdout.

    This is what ``python -m torch.distributed.run --nproc-per-node N`` does for the driver, for the case where the
    user typed the plain command (``python bench.py --gpus N``).  The CALLER MUST NOT HAVE TOUCHED THE GPU: children are
    fresh processes started with subprocess (never an exec of a GPU-initialised process), each with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set.  Rank 0's stdout goes to this process's stdout line by line
    (the JSON line of bench.py), every other stream to stderr.  Returns the largest exit code; if one rank fails the
    others are terminated (they would otherwise wait at a collective for ever)."""
    import subprocess
    import sys
    import threading
    import time

    why = gpu_preload_in_environment()
    if why:
        # rocprofv3 (and anything else that preloads a HIP tool library) initialises the GPU in THIS process before
        # main() runs; starting the ranks from here would be an exec out of a GPU-initialised process, which this
        # pool forbids.  Profile a multi-GPU run by putting the rank program itself after `--`.
        sys.stderr.write(f"self_launch: refusing to start rank processes from a process with a GPU tool preloaded ({why}); "
                         "put the rank program itself after the profiler's `--` (RANK / WORLD_SIZE / MASTER_* set by hand) "
                         "or profile --gpus 1\n")
        return 2

    port = free_port()
    procs = []
    for r in range(nprocs):
        env = dict(os.environ)
        env.update(RANK=str(r), WORLD_SIZE=str(nprocs), LOCAL_RANK=str(local_ranks[r] if local_ranks else r),
                   LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))

    def relay(p, rank):
        for line in p.stdout:
            if rank == 0:
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(f"[rank {rank}] {line}")

    threads = [threading.Thread(target=relay, args=(p, r), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    codes: List[Optional[int]] = [None] * nprocs
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):                # exactly the children started above, by handle
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return max(abs(int(c)) for c in codes)


def replicate_checkpoint(weights: Optional[Dict[str, "object"]], src: int = 0, device=None,
                         bucket_bytes: int = 1 << 30, stats: Optional[Dict[str, float]] = None) -> Dict[str, "object"]:
    """Every rank ends up with the checkpoint tensors of rank ``src`` (which read them from disk; the others pass None):
    the manifest (names, shapes, dtypes) travels as a small object, the bytes as one broadcast per flat bucket and
    dtype, straight into device memory (``device``: a torch.device; default cuda:<current> when available, else CPU).
    Returned tensors are views into the buckets -- ``mi_engine_set_tensor`` copies out of them."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(weights or {})
    rank = dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    manifest = [[(k, tuple(t.shape), str(t.dtype)) for k, t in sorted(weights.items())]] if rank == src else [None]
    dist.broadcast_object_list(manifest, src=src)
    by_dtype: Dict[str, List[Tuple[str, Tuple[int, ...]]]] = {}
    for name, shape, dt in manifest[0]:
        by_dtype.setdefault(dt, []).append((name, shape))
    out: Dict[str, "object"] = {}
    secs, nbytes, nb = 0.0, 0, 0
    for dt, items in by_dtype.items():
        tdt = getattr(torch, dt.split(".")[-1])
        wire = torch.int32 if tdt == torch.uint32 else tdt          # (collectives do not take uint32)
        esz = torch.empty(0, dtype=wire).element_size()
        for bucket in plan_buckets(items, esz, bucket_bytes):
            flat = torch.empty(bucket_numel(bucket), dtype=wire, device=device)
            if rank == src:
                for name, shape, off, n in bucket:
                    flat[off:off + n].copy_(weights[name].reshape(-1).view(wire) if weights[name].dtype != wire
                                            else weights[name].reshape(-1))
            s, b = broadcast_bucket(flat, src=src)
            secs, nbytes, nb = secs + s, nbytes + b, nb + 1
            for name, shape, off, n in bucket:
                out[name] = flat[off:off + n].view(shape)
    if stats is not None:
        stats.update(broadcast_seconds=round(secs, 4), broadcast_bytes=int(nbytes), broadcast_buckets=nb)
    return out
