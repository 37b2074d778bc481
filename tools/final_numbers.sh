#!/bin/bash
# The measurement set quoted in DESIGN.md / README.md, one box, final build.  Output: gpurun_out/final_<name>.json
set -e
b() { n=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/final_$n.json 2>> gpurun_out/final.err; echo "$n done"; }
python bench.py > gpurun_out/final_default.json 2>> gpurun_out/final.err; echo default done
b int4 --workload mistral-7b-int4
b int8 --workload mistral-7b-int8
b q_bf16 --workload qwen3-14b-bf16
b q_int4 --workload qwen3-14b-int4
b m_b32 --batch 32 --no-second-leg
b m_b64 --batch 64 --no-second-leg
b q_b32 --workload qwen3-14b-bf16 --batch 32 --no-second-leg
b q4_lora_b64 --workload qwen3-14b-int4 --lora 8 --batch 64 --no-second-leg
b mixed_m --mode mixed
b mixed_q --mode mixed --workload qwen3-14b-bf16 --batch 32
b mixed_q4 --mode mixed --workload qwen3-14b-int4 --lora 8 --batch 64
