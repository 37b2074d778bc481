#!/bin/bash
# Kernel-time summary of an arbitrary bench.py invocation (run from the repo root through gpurun):
#   tools/profile_cmd.sh <tag> <bench.py arguments...>
set -eo pipefail
# gpus_guard: bench.py --gpus N > 1 makes bench.py a LAUNCHER (subprocess of rank processes); under rocprofv3 the
# profiler's preloaded library has already initialised the GPU in it, so that would be an exec out of a GPU-initialised
# process (forbidden on this pool).  Profile one rank: put the rank program itself after `--`.
prev=""
for a in "$@"; do
  case "$prev $a" in
    "--gpus 1") ;;
    "--gpus "*) echo "$0: refusing --gpus $a under rocprofv3 (see the comment in this script)" >&2; exit 2 ;;
  esac
  case "$a" in --gpus=1) ;; --gpus=*) echo "$0: refusing $a under rocprofv3" >&2; exit 2 ;; esac
  prev="$a"
done
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/bench.py "$@" --no-cpu-baseline --no-other-configs > "$OUT/bench.json" 2> "$OUT/stats.err"
python3 "$ROOT/tools/summarize_prof.py" stats "$OUT/stats" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" "rocprofv3 --kernel-trace --stats -- python3 bench.py $*"
grep '^{' "$OUT/bench.json" > "$ROOT/gpurun_out/${TAG}_bench.json"
rm -rf "$OUT/stats"
