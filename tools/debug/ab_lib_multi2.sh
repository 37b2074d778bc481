# A/B the shipped library against csrc/alt/<name>.so on several workloads, decode only, two repetitions each:
#   tools/debug/ab_lib_multi2.sh libmi355_noslp.so
ALT=$PWD/mlx_parallm_amd/csrc/alt/$1
run() { python bench.py --no-cpu-baseline --no-second-leg --no-prefill-timing --no-other-configs --steps 32 --warmup 4 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"; }
for wl in "--workload qwen3-14b-int4 --lora 8 --batch 64" "--workload mistral-7b-int4" "--workload mistral-7b-int4 --batch 32" "--workload qwen3-14b-int4" "--workload mistral-7b-int8"; do
  for rep in 1 2; do
    echo "base  [$wl] $(run $wl)"
    echo "alt   [$wl] $(MLX_PARALLM_AMD_LIB=$ALT run $wl)"
  done
done
