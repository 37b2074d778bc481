"""Replays tests/test_gpu_fullsize.py::test_alternative_launch_structures_agree[mistral-7b-int4] and prints the greedy
token matrices of the two launch structures (single launches vs in-launch-seam pairs).  Evidence for DESIGN 7a: the
run of 2026-10-04 19:20 (gpurun_out/r2_t27.log) had tokens[5][2] = 8802 (single) vs 11657 (paired); tokens is [prefill + 5 steps][8 rows]."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import bench  # noqa: E402
from mlx_parallm_amd.engine import Engine  # noqa: E402
from test_gpu_fullsize import _greedy  # noqa: E402

cfg = dict(bench.SHAPES["mistral-7b"])
cfg["quantization"] = {"group_size": 64, "bits": 4}
eng = Engine(cfg, device=0, max_positions=2048, act_dtype="bfloat16")
bench.load_synthetic(eng, cfg, 0, 4, 0, 1, None)
rng = np.random.default_rng(6)
p = rng.integers(0, cfg["vocab_size"], size=(8, 300)).astype(np.int32)
base, _ = _greedy(eng, p, 5)
for rep in range(3):
    single, ls = _greedy(eng, p, 5, skinny_gemm=0)
    paired, lp = _greedy(eng, p, 5, skinny_gemm=0, fused_gemv_pairs=3)
    print(f"rep {rep}: single[5] = {single[5].tolist()}")
    print(f"rep {rep}: paired[5] = {paired[5].tolist()}  equal: {np.array_equal(single, paired) and np.array_equal(ls, lp)}")
print("single (all steps):\n", single)
print("base[5]   =", base[5].tolist())
eng.close()
