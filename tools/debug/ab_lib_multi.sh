# A/B two builds over several workloads on one box: tools/debug/ab_lib_multi.sh <alt .so under csrc/alt>
set -e
ALT=$1
i=0
while read -r line; do
  i=$((i+1))
  python bench.py --no-cpu-baseline --steps 32 --warmup 4 --no-prefill-timing --no-second-leg $line > gpurun_out/abm_base_$i.json 2>> gpurun_out/abm.err
  MLX_PARALLM_AMD_LIB=$PWD/mlx_parallm_amd/csrc/alt/$ALT python bench.py --no-cpu-baseline --steps 32 --warmup 4 --no-prefill-timing --no-second-leg $line > gpurun_out/abm_alt_$i.json 2>> gpurun_out/abm.err
  echo "$i $line" >> gpurun_out/abm_cases.txt
done <<'CASES'
--workload mistral-7b-bf16
--workload mistral-7b-bf16 --kv-dtype float32
--workload mistral-7b-int4
--workload mistral-7b-bf16 --batch 32
--workload qwen3-14b-int4 --lora 8 --batch 64
--workload qwen3-14b-bf16
CASES
