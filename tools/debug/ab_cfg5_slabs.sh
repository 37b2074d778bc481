set -e
B="python bench.py --workload qwen3-14b-int4 --lora 8 --batch 64 --steps 32 --warmup 4 --no-cpu-baseline --no-second-leg --no-prefill-timing"
for v in "MI_SKINNY_SLABS=0" "MI_SKINNY_SLABS=1" "MI_SKINNY_SLABS=1 MI_SKINNY_FORCE=34816:5120:1" "MI_SKINNY_SLABS=1 MI_SKINNY_FORCE=34816:5120:2" "MI_SKINNY_SLABS=1 MI_SKINNY_FORCE=34816:5120:3"; do
  echo "== $v"; env $v $B | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
done
