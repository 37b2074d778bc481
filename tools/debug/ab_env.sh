# A/B an environment setting on one box: tools/debug/ab_env.sh VAR=value [bench args...]
set -e
KV=$1; shift
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 64 --warmup 8 --no-prefill-timing "$@" > gpurun_out/ab_base_$rep.json 2>> gpurun_out/ab.err
  env $KV python bench.py --no-cpu-baseline --steps 64 --warmup 8 --no-prefill-timing "$@" > gpurun_out/ab_alt_$rep.json 2>> gpurun_out/ab.err
done
