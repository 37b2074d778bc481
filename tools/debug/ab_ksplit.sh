set -e
run() { # name workload env...
  n=$1; w=$2; shift 2
  env "$@" python bench.py --workload $w --no-cpu-baseline --steps 64 --warmup 8 --no-second-leg --no-prefill-timing --opt norm_handover=${HO:-1} > gpurun_out/r2_ks_$n.json 2>> gpurun_out/r2_ks.err
}
run q_new qwen3-14b-int4 X=1
run q_gu3only qwen3-14b-int4 MI_SKINNY_FORCE=7168:5120:3,5120:5120:4,5120:17408:4
HO=0 run q_new_noho qwen3-14b-int4 X=1
run m_new mistral-7b-int4 X=1
run m_o6 mistral-7b-int4 MI_SKINNY_FORCE=4096:4096:6
run q8_new qwen3-14b-int8 X=1
run q8_gu2 qwen3-14b-int8 MI_SKINNY_FORCE=34816:5120:2
run q8_gu3 qwen3-14b-int8 MI_SKINNY_FORCE=34816:5120:3
