#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSVs kept under profiles/.

  summarize_prof.py stats   <dir> <out.csv> "<header comment>"     (--kernel-trace --stats run)
  summarize_prof.py pmc     <dir> <COUNTER> <out.csv> "<header comment>"   (--pmc COUNTER run)
  summarize_prof.py timeline <dir> <out.csv> "<header comment>"    (--kernel-trace run: one decode step, dispatch by dispatch)

Kernel names are shortened: namespaces dropped, the template arguments of the GEMV / GEMM / attention
kernels spelled out (dtype, weight kind, rows per call, epilogue).
"""
import csv
import glob
import re
import sys
from collections import defaultdict


_TY = {"DF16b": "bf16", "DF16_": "f16", "f": "float", "i": "int", "j": "uint32"}


def _demangle_own(name: str):
    """rocprofv3's CSV writer leaves some gfx950 names mangled (its demangler does not know the DF16b = __bf16
    code); this handles the kernels of this library: _ZN2mi12_GLOBAL__N_1<len><name>I<template args>E..."""
    m = re.match(r"_ZN2mi(?:12_GLOBAL__N_1)?(\d+)", name)
    if not m:
        return None
    ln = int(m.group(1))
    base = name[m.end():m.end() + ln]
    rest = name[m.end() + ln:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while rest and not rest.startswith("E"):
            mm = re.match(r"L([bij])(\d+)E", rest)
            if mm:
                args.append(("true" if mm.group(2) == "1" else "false") if mm.group(1) == "b" else mm.group(2))
                rest = rest[mm.end():]
                continue
            for code, ty in _TY.items():
                if rest.startswith(code):
                    args.append(ty)
                    rest = rest[len(code):]
                    break
            else:
                return base
    return base + ("<" + ", ".join(args) + ">" if args else "")


def pretty(name: str) -> str:
    own = _demangle_own(name)
    if own:
        name = own.replace("bf16", "__bf16")
    if "_Accum" in name:   # rocprofv3's failed demangle of a <__bf16, true, ...> instantiation
        base = name.split("<")[0].replace("void ", "").replace("mi::(anonymous namespace)::", "").replace("mi::", "")
        return {"gemm_tile_kernel": "gemm_tile_kernel<bf16,swiglu>"}.get(base, base + "<bf16,...>")
    n = name.replace("void ", "").replace("mi::(anonymous namespace)::", "").replace("mi::", "")
    n = re.sub(r"\(.*\)$", "", n).replace(" [clone .kd]", "").replace(".kd", "")
    m = re.match(r"gemv_mfma_kernel<(\w+), (true|false), (\d+), (true|false)(?:, (\d+))?(?:, (\d+))?(?:, (true|false))?>", n)
    if m:
        at = {"__bf16": "bf16", "_Float16": "f16"}.get(m.group(1), m.group(1))
        return "gemv_mfma_kernel<%s,%s,MB=%s,%s%s>" % (at, "int4" if m.group(2) == "true" else "dense", m.group(3),
                                                        "swiglu" if m.group(4) == "true" else "plain",
                                                        ",dbuf" if m.group(7) == "true" else "")
    m = re.match(r"gemv_mfma_gu8_kernel<(\w+), (\d+)", n)
    if m:
        at = {"__bf16": "bf16", "_Float16": "f16"}.get(m.group(1), m.group(1))
        return "gemv_mfma_gu8_kernel<%s,MB=%s>" % (at, m.group(2))
    m = re.match(r"gemm_tile_kernel<(\w+), (true|false)>", n)
    if m:
        at = {"__bf16": "bf16", "_Float16": "f16"}.get(m.group(1), m.group(1))
        return "gemm_tile_kernel<%s,%s>" % (at, "swiglu" if m.group(2) == "true" else "plain")
    return n.replace("__bf16", "bf16").replace("_Float16", "f16").replace(" ", "")


def find(d, suffix):
    fs = sorted(glob.glob(f"{d}/**/*{suffix}", recursive=True))
    if not fs:
        sys.exit(f"no *{suffix} under {d}")
    return fs


def stats(d, out, header):
    agg = defaultdict(lambda: [0, 0.0])
    for f in find(d, "_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            k = pretty(r["Name"])
            agg[k][0] += int(r["Calls"])
            agg[k][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as o:
        for h in header.split("\\n"):
            o.write(f"# {h}\n")
        o.write("kernel,calls,total_ms,avg_us,percent\n")
        for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{k},{c},{ns / 1e6:.3f},{ns / c / 1e3:.2f},{100 * ns / tot:.2f}\n")


def pmc(d, counter, out, header):
    agg = defaultdict(lambda: [0, 0.0])
    for f in find(d, "_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = pretty(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    with open(out, "w") as o:
        for h in header.split("\\n"):
            o.write(f"# {h}\n")
        o.write("kernel,dispatches,avg_KiB_per_dispatch\n" if counter.endswith("_SIZE") else "kernel,dispatches,avg_per_dispatch\n")
        for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{k},{c},{v / c:.1f}\n")


def timeline(d, out, header, marker="sample_kernel", skip_last=70):
    """One steady-state decode step from a --kernel-trace run: the dispatches between the last two `marker` kernels, each
    with its start offset, duration and the idle gap in front of it (us).  Rows: the embedding, layers 0, 1 and the
    last layer in full, the head + sampler; plus per-kernel sums over the whole step."""
    rows = []
    for f in find(d, "_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), pretty(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
    if len(marks) < 3:
        sys.exit("timeline: fewer than three sampler dispatches in the trace")
    # bench.py ends with an instrumented pass (64 steps with HIP events around the dominant kernel, which open gaps of
    # their own): take a step of the TIMED region in front of it
    back = skip_last if len(marks) > skip_last + 2 else 1
    # of the eight steps in front of that point, the one with the shortest span: a single step can carry a host-side pause
    # of the profiler (seen: 3.5 ms between two kernels of one step of a 9-ms step) that says nothing about the engine
    best = None
    for bk in range(back, min(back + 8, len(marks) - 1)):
        a, b = marks[-bk - 1] + 1, marks[-bk] + 1
        cand = rows[a:b]
        if cand and (best is None or cand[-1][1] - cand[0][0] < best[-1][1] - best[0][0]):
            best = cand
    step = list(best)
    t0 = step[0][0]
    span = (step[-1][1] - t0) / 1e3
    busy = sum(e - s_ for s_, e, _ in step) / 1e3
    agg = defaultdict(lambda: [0, 0.0])
    for s_, e, k in step:
        agg[k][0] += 1
        agg[k][1] += (e - s_) / 1e3
    with open(out, "w") as o:
        for h in header.split("\\n"):
            o.write(f"# {h}\n")
        o.write(f"# one decode step: {len(step)} dispatches, {span:.1f} us from first start to last end, {busy:.1f} us inside kernels, "
                f"{span - busy:.1f} us between them\n")
        o.write("# per kernel over the step: " + "; ".join(f"{k} x{c} = {t:.1f} us" for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])) + "\n")
        o.write("index,kernel,start_us,duration_us,gap_before_us\n")
        n = len(step)
        per_layer = max(1, (n - 4) // 32)
        keep = set(range(0, min(n, 1 + 2 * per_layer + 1))) | set(range(max(0, n - per_layer - 4), n))
        prev_end = None
        for i, (s_, e, k) in enumerate(step):
            gap = 0.0 if prev_end is None else (s_ - prev_end) / 1e3
            prev_end = e
            if i in keep:
                o.write(f"{i},{k},{(s_ - t0) / 1e3:.2f},{(e - s_) / 1e3:.2f},{gap:.2f}\n")


def rename(path):
    """Re-apply pretty() to the first column of an existing summary (merging rows that collapse)."""
    lines = open(path).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    out = head + [body[0]]
    for l in body[1:]:
        k, rest = l.rsplit(",", len(body[0].split(",")) - 1)[0], l.rsplit(",", len(body[0].split(",")) - 1)[1:]
        out.append(",".join([pretty(k)] + rest))
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "rename":
        for f in sys.argv[2:]:
            rename(f)
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
    else:
        sys.exit(__doc__)
