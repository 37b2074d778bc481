#!/usr/bin/env python3
"""How long the HOST takes to enqueue one decode step (all its launches) against how long the GPU takes to run it:
if the two are close, the step is launch-bound and a hipGraph would pay.  Uses bench.py's synthetic loader."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import bench
from mlx_parallm_amd.engine import Engine, SampleArgs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mistral-7b-int4")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--context", type=int, default=1024)
    args = ap.parse_args()
    family, prec = args.workload.rsplit("-", 1)
    qb = {"int4": 4, "int8": 8}.get(prec, 0)
    cfg = dict(bench.SHAPES[family])
    if qb:
        cfg["quantization"] = {"group_size": 64, "bits": qb}
    torch.cuda.set_device(0)
    eng = Engine(cfg, device=0, max_positions=4096, act_dtype="bfloat16")
    bench.load_synthetic(eng, cfg, 0, qb, 0, 1, None)
    B = args.batch
    kv = eng.new_kv(B, capacity=args.context + 400, kv_dtype="model")
    sa = SampleArgs(temp=0.0)
    prompts = np.random.default_rng(0).integers(0, cfg["vocab_size"], size=(B, args.context)).astype(np.int32)
    eng.step_wait(eng.step_enqueue(kv, prompts, sa), B)
    for _ in range(8):
        eng.step_wait(eng.step_enqueue(kv, None, sa), B)
    eng.sync()
    n = 3                                         # the result ring has 4 slots: enqueue 3 steps ahead, then drain
    t_host, t_all = 0.0, 0.0
    reps = 20
    for _ in range(reps):
        t0 = time.perf_counter()
        tickets = [eng.step_enqueue(kv, None, sa) for _ in range(n)]
        t1 = time.perf_counter()
        for t in tickets:
            eng.step_wait(t, B)
        t2 = time.perf_counter()
        t_host += t1 - t0
        t_all += t2 - t0
    print(f"{args.workload} B={B}: host enqueue {t_host / (reps * n) * 1e3:.3f} ms/step, end to end {t_all / (reps * n) * 1e3:.3f} ms/step")


if __name__ == "__main__":
    main()
