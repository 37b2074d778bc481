"""Prefill GEMM throughput per linear (HIP events around every launch of one family), 8 x 1024 rows, Mistral-7B shapes
with a reduced number of layers.  A/B builds:  MLX_PARALLM_AMD_LIB=/path/to/other.so python tools/debug/gemm_ab.py"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench
from mlx_parallm_amd.engine import Engine

family = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
kvd = sys.argv[2] if len(sys.argv) > 2 else "model"
cfg = dict(bench.SHAPES[family]); cfg["num_hidden_layers"] = 6
eng = Engine(cfg, device=0, max_positions=2048, act_dtype="bfloat16")
bench.load_synthetic(eng, cfg, 0, 0, 0, 1, None)
H, I = cfg["hidden_size"], cfg["intermediate_size"]
nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
D = cfg.get("head_dim") or H // nh
BL = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 1024)      # batch, prompt length
p = np.random.default_rng(0).integers(0, cfg["vocab_size"], size=BL).astype(np.int32)
rows = p.size
flops = {"gemv_qkv": 2.0 * rows * (nh + 2 * nkv) * D * H, "gemv_o": 2.0 * rows * H * nh * D,
         "gemv_gate_up": 2.0 * rows * 2 * I * H, "gemv_down": 2.0 * rows * I * H}
mult = 3.0 if kvd == "float32" else 1.0
print("lib:", os.environ.get("MLX_PARALLM_AMD_LIB", "default"), "kv:", kvd)
for fam in ("gemv_gate_up", "gemv_qkv", "gemv_o", "gemv_down", "attn", "rope_append", "embed"):
    best = 1e9
    for it in range(3):
        kv = eng.new_kv(BL[0], capacity=BL[1] + 16, kv_dtype=kvd)
        eng.profile_select(fam)
        eng.forward(p, kv, want_logits=False)
        n, ms = eng.profile_read()
        eng.profile_select(None)
        kv.close()
        best = min(best, ms / max(n, 1))
    print(f"  {fam:13s} {best*1e3:8.1f} us per launch  {flops.get(fam, 0.0) * mult / (best * 1e-3) / 1e12:7.1f} TFLOP/s (MFMA work)", flush=True)
