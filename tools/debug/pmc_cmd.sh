#!/bin/bash
# Counter passes over ANY single-process GPU command (one rocprofv3 --pmc pass per counter group, never combined with a
# trace domain) + one --kernel-trace --stats pass:
#   tools/debug/pmc_cmd.sh <tag> "<group1 counters>;<group2 counters>;..." <kernel-name regex> -- python3 <script> [args]
# Prints, per kernel matching the regex, the average of every counter per dispatch and the average duration.
set -o pipefail
TAG=$1; GROUPS_=$2; REGEX=$3; shift 3
[ "$1" = "--" ] && shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
IFS=';' read -ra GRP <<< "$GROUPS_"
i=0
for G in "${GRP[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $G --output-format csv -d "$OUT/p$i" -o run -- "$@" > "$OUT/p$i.out" 2> "$OUT/p$i.err" || { echo "pass $i failed"; tail -3 "$OUT/p$i.err"; }
  i=$((i+1))
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o run -- "$@" > "$OUT/kt.out" 2> "$OUT/kt.err" || echo "kernel-trace pass failed"
python3 - "$OUT" "$REGEX" <<'PY' | tee "$ROOT/gpurun_out/pmc_${TAG}.txt"
import csv, glob, re, sys
from collections import defaultdict
out, rx = sys.argv[1], re.compile(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if rx.search(k):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if rx.search(k):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(set(acc) | set(dur)):
    print(k[:110])
    if dur[k]:
        d = sorted(dur[k]); print(f"   dispatches {len(d)}  avg {sum(d)/len(d):.1f} us  median {d[len(d)//2]:.1f} us  min {d[0]:.1f}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
rm -rf "$OUT"/p*/ "$OUT"/kt/
