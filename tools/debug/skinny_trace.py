#!/usr/bin/env python3
"""Where the time of a gemm_skinny launch goes (debug build only).

Build the library with -DMI_SK_TRACE (tools/debug/build_trace_lib.sh), then on a GPU box:

    MLX_PARALLM_AMD_LIB=$PWD/mlx_parallm_amd/csrc/alt/libmi355_trace.so \
        python tools/debug/skinny_trace.py --workload mistral-7b-int4 [--kv float32]

Runs bench.py's decode leg (4 timed steps), dumps the per-workgroup wall-clock stamps (100 MHz s_memrealtime) of the
last launches and prints, per linear of the decode step, the average of:
  ramp    first workgroup's entry -> last workgroup's entry
  stage_norm / w0..w7 / stage   entry -> row statistics known / wave w has written its share of the first activation
          chunk / the chunk is complete (barrier)
  stream  the K slice (weights streamed, MFMA)
  publish partial tile stored write-through + drained
  count   arrival counter
  combine (last arriver) the ksplit partials read back and added
  epi     (last arriver) epilogue
  total   first entry -> last exit; `to next` = last exit -> first entry of the NEXT skinny launch (whatever runs between)
"""
import argparse
import ctypes as C
import struct
import sys
from collections import OrderedDict, defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

TICK_US = 0.01          # s_memrealtime: 100 MHz


def read_dump(path):
    recs = []
    with open(path, "rb") as f:
        (n,) = struct.unpack("l", f.read(8))
        for _ in range(n):
            hdr = struct.unpack("12i", f.read(48))
            grid = hdr[4]
            import numpy as np

            st = np.frombuffer(f.read(grid * 128), dtype=np.uint64).reshape(grid, 16).astype(np.int64)
            recs.append((hdr, st))
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mistral-7b-int4")
    ap.add_argument("--kv", default="model")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/skinny_trace.bin")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value, passed on to bench.py")
    args = ap.parse_args()

    import bench

    argv = ["--workload", args.workload, "--no-cpu-baseline", "--steps", str(args.steps), "--warmup", "2", "--batch", str(args.batch),
            "--no-prefill-timing"]
    argv += ["--no-second-leg"] if args.kv == "model" else []
    for o in args.opt:
        argv += ["--opt", o]
    sys.argv = ["bench.py"] + argv
    bench.main()
    from mlx_parallm_amd import _lib as L

    lib = C.CDLL(str(L.LIB_PATH))
    if not hasattr(lib, "mi_debug_sk_trace_dump"):
        raise SystemExit("this build has no trace (compile gemm_skinny.hip with -DMI_SK_TRACE)")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    rc = lib.mi_debug_sk_trace_dump(str(args.out).encode())
    if rc:
        raise SystemExit(f"dump failed: {rc}")
    report(args.out, args.steps)


def report(path, steps):
    import numpy as np

    recs = read_dump(path)
    # launches of one decode step: signature sequence repeats; take the steps of the TIMED region = the `steps` steps in front of
    # the last `steps` (instrumented) ones of the last leg
    sig = lambda h: (h[1], h[2], h[5], h[8], h[7], h[9], h[6])      # N, K, epi, pro, qb, act, M
    last_sig = sig(recs[-1][0])
    ends = [i for i, (h, _) in enumerate(recs) if sig(h) == last_sig]
    # step length = distance between consecutive occurrences of the last launch (lm_head)
    per = ends[-1] - ends[-2]
    hi = ends[-1] + 1 - steps * per
    lo = hi - steps * per
    sel = recs[lo:hi]
    print(f"{len(recs)} launches in the dump, {per} skinny launches per decode step, analysing launches {lo}..{hi - 1}")
    rows = OrderedDict()
    acc = defaultdict(lambda: defaultdict(list))
    for j, (h, st) in enumerate(sel):
        pos = j % per
        layer_pos = pos if pos >= per - 1 else pos % 4           # q|k|v, o, gate|up, down x layers, then lm_head
        key = ("head" if pos == per - 1 else ("qkv", "o", "gate_up", "down")[layer_pos], h[1], h[2], h[3], h[4], h[10])
        if j < per:
            rows.setdefault(key, None)
        s0 = st[:, 0].min()
        a = acc[key]
        a["ramp"].append((st[:, 0].max() - s0) * TICK_US)
        a["stage"].append(np.mean(st[:, 2] - st[:, 0]) * TICK_US)
        a["stage_norm"].append(np.mean(st[:, 1] - st[:, 0]) * TICK_US)
        for w in range(8):
            a[f"w{w}"].append(np.mean(st[:, 8 + w] - st[:, 0]) * TICK_US)
        a["stream"].append(np.mean(st[:, 3] - st[:, 2]) * TICK_US)
        if h[3] > 1:
            a["publish"].append(np.mean(st[:, 4] - st[:, 3]) * TICK_US)
            a["count"].append(np.mean(st[:, 5] - st[:, 4]) * TICK_US)
            lastm = st[:, 6] >= st[:, 5]
            lastm &= st[:, 5] >= s0
            a["combine"].append(np.mean((st[:, 6] - st[:, 5])[lastm]) * TICK_US if lastm.any() else 0.0)
            a["epi"].append(np.mean((st[:, 7] - st[:, 6])[lastm]) * TICK_US if lastm.any() else 0.0)
            end = max(st[:, 5].max(), st[:, 7][lastm].max() if lastm.any() else 0)
        else:
            a["publish"].append(0.0); a["count"].append(0.0); a["combine"].append(0.0)
            a["epi"].append(np.mean(st[:, 7] - st[:, 3]) * TICK_US)
            end = st[:, 7].max()
        a["total"].append((end - s0) * TICK_US)
        a["last_stream_end"].append((st[:, 3].max() - s0) * TICK_US)
        if j + 1 < len(sel):
            a["to_next"].append((sel[j + 1][1][:, 0].min() - end) * TICK_US)
    cols = ["ramp", "stage_norm", "w0", "w1", "w2", "w3", "w4", "w5", "w6", "w7", "stage", "stream", "last_stream_end", "publish", "count", "combine", "epi", "total", "to_next"]
    print(f"{'linear':<10}{'N':>7}{'K':>7}{'ksplit':>7}{'grid':>6}{'mt':>3} " + " ".join(f"{c:>7}" for c in cols))
    for key in rows:
        a = acc[key]
        print(f"{key[0]:<10}{key[1]:>7}{key[2]:>7}{key[3]:>7}{key[4]:>6}{key[5]:>3} " + " ".join(f"{np.mean(a[c]):>7.2f}" for c in cols))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    else:
        main()
