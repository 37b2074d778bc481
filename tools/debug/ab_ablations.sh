# gate|up launch time (ms) and step time under each timing-only ablation build (tools/debug/build_ablation_libs.sh)
run() { python bench.py --no-cpu-baseline --no-second-leg --no-prefill-timing --no-other-configs --steps 32 --warmup 4 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('step ms', j['ms_per_step'], ' gate|up ms', j['roofline']['avg_launch_ms'])"; }
for wl in "--workload qwen3-14b-int4 --lora 8 --batch 64" "--workload mistral-7b-int4" "--workload qwen3-14b-int4 --batch 32"; do
  echo "base      [$wl] $(run $wl)"
  for lib in nofma nounpack nostage nosx nofma_nounpack_nostage_nosx; do
    echo "$lib [$wl] $(MLX_PARALLM_AMD_LIB=$PWD/mlx_parallm_amd/csrc/alt/libmi355_abl_$lib.so run $wl)"
  done
done
