#!/bin/bash
# All of round 4's committed profiles, one box (run from the repo root through gpurun; ~15 min):
#   kernel stats + step timeline + FETCH_SIZE / WRITE_SIZE passes per configuration -> gpurun_out/round4*_*.csv / .json
# The bench's default KV mode is float32 (the reference's numerics) since round 4; every leg names its mode.
set -o pipefail
bash tools/profile_round.sh round4_f32kv --kv-dtype float32 --no-second-leg
bash tools/profile_round.sh round4_bf16kv --kv-dtype model --no-second-leg
bash tools/profile_round.sh round4_int4 --workload mistral-7b-int4 --kv-dtype model --no-second-leg
bash tools/profile_round.sh round4_int8 --workload mistral-7b-int8 --kv-dtype model --no-second-leg
bash tools/profile_round.sh round4_cfg4 --workload qwen3-14b-bf16 --batch 32 --kv-dtype model --no-second-leg
bash tools/profile_round.sh round4_cfg5 --workload qwen3-14b-int4 --lora 8 --batch 64 --kv-dtype model --no-second-leg
echo "profiles done"
