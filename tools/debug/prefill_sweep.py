"""Prefill throughput against the number of prompt rows (one sequence of L tokens), Mistral-7B bf16 shape: where does
the tile GEMM stop paying?  Run with / without MI_GEMM_TILE128=1."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench
from mlx_parallm_amd.engine import Engine, SampleArgs

wl = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b-bf16"
family, prec = wl.rsplit("-", 1)
q = {"int4": 4, "int8": 8}.get(prec, 0)
cfg = dict(bench.SHAPES[family])
if q:
    cfg["quantization"] = {"group_size": 64, "bits": q}
eng = Engine(cfg, device=0, max_positions=4096, act_dtype="bfloat16")
bench.load_synthetic(eng, cfg, 0, q, 0, 1, None)
for kv_ in sys.argv[2:]:                                   # engine options key=value
    k_, v_ = kv_.split("=")
    eng.set_option(k_, int(v_))
    print("option", k_, v_)
rng = np.random.default_rng(0)
g = SampleArgs(temp=0.0)
for L in (32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048):
    p = rng.integers(0, cfg["vocab_size"], size=(1, L)).astype(np.int32)
    ts = []
    for it in range(4):
        kv = eng.new_kv(1, capacity=L + 8, kv_dtype="model")
        eng.sync(); t0 = time.perf_counter()
        eng.step_wait(eng.step_enqueue(kv, p, g), 1)
        ts.append(time.perf_counter() - t0)
        kv.close()
    t = min(ts[1:])
    print(f"L={L:5d}: {t*1e3:8.2f} ms  {L/t:9.0f} tok/s", flush=True)
