# A/B environment settings on several int4 workloads (decode only): tools/debug/ab_env_multi.sh "VAR=a" "VAR=b" ...
run() { python bench.py --no-cpu-baseline --no-second-leg --no-prefill-timing --no-other-configs --steps 32 --warmup 4 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], 'gate|up', j['roofline']['avg_launch_ms'])"; }
for wl in "--workload qwen3-14b-int4 --lora 8 --batch 64" "--workload mistral-7b-int4" "--workload mistral-7b-int4 --batch 32" "--workload qwen3-14b-int4" "--workload mistral-7b-int4 --batch 64"; do
  for v in "$@"; do
    echo "$v [$wl] $(env $v bash -c "$(declare -f run); run $wl")"
  done
done
