#!/bin/bash
# One rocprofv3 --pmc pass per counter (never combined with trace domains) over a short bench run:
#   tools/profile_pmc.sh <tag> "<COUNTER> <COUNTER> ..." [bench.py arguments]
# Writes gpurun_out/<tag>_pmc_<COUNTER>.csv (per-kernel averages; tools/summarize_prof.py pmc).
set -o pipefail
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
TAG=$1; CTRS=$2; shift 2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-second-leg --no-prefill-timing $*"
for CTR in $CTRS; do
  if timeout -k 10 300 rocprofv3 --pmc $CTR --output-format csv -d "$OUT/pmc_$CTR" -o run -- $CMD > "$OUT/pmc_$CTR.out" 2> "$OUT/pmc_$CTR.err"; then
    python3 "$ROOT/tools/summarize_prof.py" pmc "$OUT/pmc_$CTR" $CTR "$ROOT/gpurun_out/${TAG}_pmc_$CTR.csv" \
      "rocprofv3 --pmc $CTR -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-second-leg --no-prefill-timing $* (own pass; average per dispatch)" || echo "summary failed for $CTR"
  else
    echo "counter $CTR: rocprofv3 failed (see $OUT/pmc_$CTR.err)"; tail -3 "$OUT/pmc_$CTR.err"
  fi
  rm -rf "$OUT/pmc_$CTR"
done
ls -la "$ROOT/gpurun_out/" | grep "${TAG}_pmc" || true
