set -e
for ns in 0 3 2; do
  MI_ATTN_NSPLIT=$ns python bench.py --no-cpu-baseline --steps 64 --warmup 8 --no-prefill-timing > gpurun_out/r2_ns_$ns.json 2>> gpurun_out/r2_ns.err
done
