"""Times chain.hip at the Mistral-7B block shapes next to the four single launches (tools/debug/chain/test_chain.py helpers; needs MLX_PARALLM_AMD_LIB=.../alt/libmi355_chain.so, tools/debug/build_chain_lib.sh).
MI_CHAIN_DEBUG=<bits> selects the timing-only ablations of chain.hip (results wrong on purpose)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import test_chain as T  # noqa: E402

name, H, I, NQ, NQKV = (sys.argv[1:] + ["mistral-7b"])[0], 4096, 14336, 4096, 6144
if name == "qwen3-14b":
    H, I, NQ, NQKV = 5120, 17408, 5120, 7168
lin, ws, bufs, keep = T._block(H, I, NQ, NQKV, 8, "bfloat16")
_, t1 = T._run(lin, bufs, 8, H, I, NQKV, "bfloat16", chained=False, iters=30)
_, t2 = T._run(lin, bufs, 8, H, I, NQKV, "bfloat16", chained=True, iters=30)
print(f"{name}: singles {t1 * 1e3:.1f} us  chain {t2 * 1e3:.1f} us")
