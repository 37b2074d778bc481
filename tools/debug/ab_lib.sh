# A/B two builds of the library on one box: tools/debug/ab_lib.sh <alt .so name under csrc/alt> [bench args...]
set -e
ALT=$1; shift
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 64 --warmup 8 --no-prefill-timing "$@" > gpurun_out/ab_base_$rep.json 2>> gpurun_out/ab.err
  MLX_PARALLM_AMD_LIB=$PWD/mlx_parallm_amd/csrc/alt/$ALT python bench.py --no-cpu-baseline --steps 64 --warmup 8 --no-prefill-timing "$@" > gpurun_out/ab_alt_$rep.json 2>> gpurun_out/ab.err
done
