#!/bin/bash
# All of round 3's committed profiles, one box (run from the repo root through gpurun; ~15 min):
#   kernel stats + step timeline + FETCH_SIZE / WRITE_SIZE passes per configuration -> gpurun_out/round3*_*.csv / .json
set -o pipefail
bash tools/profile_round.sh round3_bf16kv --no-second-leg
bash tools/profile_round.sh round3_f32kv --kv-dtype float32 --no-second-leg
bash tools/profile_round.sh round3_int4 --workload mistral-7b-int4 --no-second-leg
bash tools/profile_round.sh round3_int8 --workload mistral-7b-int8 --no-second-leg
bash tools/profile_round.sh round3_cfg4 --workload qwen3-14b-bf16 --batch 32 --no-second-leg
bash tools/profile_round.sh round3_cfg5 --workload qwen3-14b-int4 --lora 8 --batch 64 --no-second-leg
echo "profiles done"
