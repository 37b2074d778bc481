#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in libmi355_decode.so, from the compiler's own report.

The Makefile compiles each source with -Rpass-analysis=kernel-resource-usage and keeps the remarks next to the object
(csrc/*.res).  This script prints them as one table and FAILS when a kernel that the engine launches spills to scratch:
a spill inside a weight-streaming loop costs far more than the registers it saves, and a later edit can bring
one back silently (it did once in round 3: an index computed per thread instead of per chunk).

    python tools/kernel_resources.py [--out profiles/roundN_kernel_resources.txt]

Instantiations listed in UNUSED are compiled but never launched (host-side plans do not select them).
"""
import argparse
import glob
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "mlx_parallm_amd", "csrc")
CXXFILT = "c++filt"

# (regex on the demangled name, why it may spill)
UNUSED = [
    (r"skinny_kernel<[^,]+, 4, 8, true, false>", "int4 SwiGLU at 128 rows: skinny_plan() runs it as four 32-row slabs"),
    (r"attn_decode_kernel<", "the vector-ALU decode attention (variant 1): kept for A/B runs and head_dim 16 / 32 in float32; "
                             "the engine's default is attn_decode_mfma_kernel"),
]

FIELDS = {
    "TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
    "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds",
}


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"file": os.path.basename(path)[:-4], "name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    return rows


def demangle(names):
    # the system c++filt predates the bf16 / _Float16 manglings: hand them over as vendor types
    pre = [n.replace("DF16b", "u4bf16").replace("DF16_", "u3f16") for n in names]
    try:
        out = subprocess.run([CXXFILT], input="\n".join(pre), capture_output=True, text=True, check=True).stdout.split("\n")
    except Exception:
        return names
    res = []
    for o in out[:len(names)]:
        o = o.replace("(anonymous namespace)::", "").replace("mi::", "").replace("void ", "")
        res.append(re.sub(r"\(.*$", "", o))
    return res


def collect():
    rows = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.res"))):
        rows += parse(f)
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["kernel"] = d
    return rows


def spilling(rows):
    bad = []
    for r in rows:
        if r.get("scratch", 0) > 0 and not any(re.search(rx, r["kernel"]) for rx, _ in UNUSED):
            bad.append(r)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    a = ap.parse_args()
    rows = collect()
    if not rows:
        sys.exit("no csrc/*.res files: build the library first (make -C mlx_parallm_amd/csrc)")
    lines = ["%-14s %5s %5s %5s %7s %4s %7s  %s" % ("source", "vgpr", "agpr", "sgpr", "scratch", "occ", "lds", "kernel")]
    for r in rows:
        lines.append("%-14s %5d %5d %5d %7d %4d %7d  %s" % (r["file"], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1),
                                                          r.get("occ", -1), r.get("lds", -1), r["kernel"]))
    bad = spilling(rows)
    lines.append("")
    lines.append("%d kernels; spilling and launched by the engine: %d" % (len(rows), len(bad)))
    for rx, why in UNUSED:
        lines.append("not launched (may spill): %s -- %s" % (rx, why))
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)
    if bad:
        sys.exit("SPILLS: " + ", ".join(r["kernel"] for r in bad))


if __name__ == "__main__":
    main()
