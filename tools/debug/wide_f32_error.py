"""Where does the float32-KV (PagedKVCache) mode lose accuracy at production width?  Runs on the GPU box: oracle
(NumPy, host cores) vs engine on the wide Mistral checkpoint, logits of a prefill and of decode steps, under
different engine options."""
import sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import wide_models
from oracle import ref_generate, ref_model
from mlx_parallm_amd import utils

ref_model.CACHE_F64 = True
fam = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
L0s = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["8", "64", "300"])]
d = tempfile.mkdtemp()
cfg = wide_models.build_checkpoint(d, fam, prec, 11)
ref = ref_generate.load(d, max_pos=2048)
model = utils.load_model(d, max_positions=2048)
eng = model.engine
rng = np.random.default_rng(0)
B = 4
for L0 in L0s:
    p = rng.integers(3, cfg["vocab_size"], size=(B, L0)).astype(np.int32)
    cache = ref.make_cache(B, paged=True)
    t0 = time.time()
    want = ref(p, cache=cache, last_only=True)[:, -1]
    nxt = np.argmax(want, -1)[:, None]
    want2 = ref(nxt, cache=cache, last_only=True)[:, -1]
    print(f"L0={L0}: oracle {time.time()-t0:.1f}s", flush=True)
    for name, opts in [("default", {}), ("generic gemv", {"force_generic_gemv": 1}),
                       ("no skinny", {"skinny_gemm": 0}), ("valu attn", {"decode_attention_mfma": 0}),
                       ("unfused attn", {"fused_decode_attention": 0}), ("no defer", {"defer_norm": 0}),
                       ("all exact", {"force_generic_gemv": 1, "fused_decode_attention": 0, "prefill_gemm": 0})]:
        for k, v in opts.items():
            eng.set_option(k, v)
        kv = eng.new_kv(B, capacity=L0 + 8, kv_dtype="float32")
        got = eng.forward(p, kv)
        got2 = eng.forward(nxt.astype(np.int32), kv)
        kv.close()
        for k in opts:
            eng.set_option(k, {"force_generic_gemv": 0, "skinny_gemm": 1, "decode_attention_mfma": 1, "fused_decode_attention": 1,
                               "prefill_gemm": 1, "defer_norm": 1}[k])
        e1, e2 = np.abs(got - want), np.abs(got2 - want2)
        print(f"  {name:14s} prefill max|dlogit| {e1.max():.2e} rms {np.sqrt((e1**2).mean()):.2e}   decode max {e2.max():.2e} rms {np.sqrt((e2**2).mean()):.2e}", flush=True)
