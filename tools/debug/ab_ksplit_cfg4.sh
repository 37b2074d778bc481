# split-K sweep of the Qwen3-14B bf16 linears at 32 rows (config-4 shard): MI_SKINNY_FORCE="N:K:ksplit"
run() { python bench.py --workload qwen3-14b-bf16 --batch 32 --no-cpu-baseline --no-second-leg --no-prefill-timing --no-other-configs --steps 32 --warmup 4 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], 'kernel ms', j['roofline']['avg_launch_ms'])"; }
echo "plan: $(run)"
for ks in 1 2 3 4; do echo "gate|up ksplit $ks: $(MI_SKINNY_FORCE=34816:5120:$ks run)"; done
for ks in 2 4 6 8; do echo "down ksplit $ks: $(MI_SKINNY_FORCE=5120:17408:$ks run --profile-kernel gemv_down)"; done
for ks in 2 3 4 6; do echo "qkv ksplit $ks: $(MI_SKINNY_FORCE=7168:5120:$ks run --profile-kernel gemv_qkv)"; done
for ks in 2 4 6 8; do echo "o ksplit $ks: $(MI_SKINNY_FORCE=5120:5120:$ks run --profile-kernel gemv_o)"; done
