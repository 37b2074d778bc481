# RMSNorm hand-over between launches (engine option norm_handover) on / off at larger decode batches
run() { python bench.py --no-cpu-baseline --no-second-leg --no-prefill-timing --no-other-configs --steps 32 --warmup 4 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(j['value'], j['ms_per_step'])"; }
for wl in "--workload qwen3-14b-int4 --lora 8 --batch 64" "--workload qwen3-14b-bf16 --batch 32" "--workload mistral-7b-bf16 --batch 32" "--workload mistral-7b-int4 --batch 32" "--workload mistral-7b-bf16 --batch 64" "--workload mistral-7b-int4"; do
  echo "handover on  [$wl] $(run $wl)"
  echo "handover off [$wl] $(run $wl --opt norm_handover=0)"
done
