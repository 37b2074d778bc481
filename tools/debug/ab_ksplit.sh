# A/B split-K plans on one box: MI_SKINNY_FORCE="N:K:ksplit,..." overrides the cost model for the linears named
set -e
run() { # name workload env...
  n=$1; w=$2; shift 2
  env "$@" python bench.py --workload $w --no-cpu-baseline --steps 64 --warmup 8 --no-prefill-timing > gpurun_out/r2_ks_$n.json 2>> gpurun_out/r2_ks.err
}
run f_base mistral-7b-bf16 X=1
run f_gu4 mistral-7b-bf16 MI_SKINNY_FORCE=28672:4096:4
run f_gu3 mistral-7b-bf16 MI_SKINNY_FORCE=28672:4096:3
run f_qkv5 mistral-7b-bf16 MI_SKINNY_FORCE=6144:4096:5
run f_down8 mistral-7b-bf16 MI_SKINNY_FORCE=4096:14336:8
run f_head2 mistral-7b-bf16 MI_SKINNY_FORCE=32000:4096:2
run f_base2 mistral-7b-bf16 X=1
