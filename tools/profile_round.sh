#!/bin/bash
# Produces the profiles/ summaries on a GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh <tag> [bench.py arguments ...]     e.g. round2   |   round2_int4 --workload mistral-7b-int4
# Three separate rocprofv3 runs of the bench command: kernel trace + stats, then one --pmc pass per counter
# (PMC passes never share a run with other trace domains).
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
TAG=${1:-round1}
shift || true
EXTRA="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd /tmp
STATS_CMD="python3 $ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-other-configs $EXTRA"
PMC_CMD="python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-second-leg $EXTRA"

timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $STATS_CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
python3 "$ROOT/tools/summarize_prof.py" stats "$OUT/stats" "$ROOT/gpurun_out/${TAG}_bench_kernel_stats.csv" \
  "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline $EXTRA   (B=8, KV 1024; per KV mode: prefill x2 + 8 warm-up + 64 timed + 64 instrumented decode steps; both KV modes unless --no-second-leg)\ngemm_tile / rmsnorm_rows / split3 / attn_prefill kernels are the prefill, everything else the decode steps\nthe bench line printed by this same run is kept next to this file (*_bench_under_rocprof.json); the profiler lengthens bench.py's event bracket around the dominant kernel by ~10 %, not the kernel"
python3 "$ROOT/tools/summarize_prof.py" timeline "$OUT/stats" "$ROOT/gpurun_out/${TAG}_step_timeline.csv" \
  "one decode step of: python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline $EXTRA (rocprofv3 --kernel-trace; the last KV mode that ran)"
grep '^{' "$OUT/bench_under_rocprof.json" > "$ROOT/gpurun_out/${TAG}_bench_under_rocprof.json"

for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $CTR --output-format csv -d "$OUT/pmc_$CTR" -o run -- $PMC_CMD > "$OUT/pmc_$CTR.out" 2> "$OUT/pmc_$CTR.err"
  python3 "$ROOT/tools/summarize_prof.py" pmc "$OUT/pmc_$CTR" $CTR "$ROOT/gpurun_out/${TAG}_pmc_$CTR.csv" \
    "rocprofv3 --pmc $CTR -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-second-leg $EXTRA (own pass; $CTR in KiB per dispatch as rocprofv3 reports it)\ngfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> hbm_bytes ~ 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM)"
done
rm -rf "$OUT/stats" "$OUT"/pmc_*/   # raw traces are large; the summaries are what travels back
ls -la "$ROOT/gpurun_out/"
