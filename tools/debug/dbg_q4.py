import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle.numerics import matmul_nt, round_to
from mlx_parallm_amd import _lib as L
from gpu_helpers import dev, gemm_skinny, host, q4_force
import test_gpu_skinny as T
act, kind = "bfloat16", "q4_bf16"
for plan in [(1,8,1,1,2),(1,1,8,1,4)]:
  for M, N, K in [(64, 64, 256)]:
    ol, wdense, keep = T._weight(kind, N, K)
    x = round_to(T.RNG.standard_normal((M, K)).astype(np.float32), act)
    xd = dev(x, act)
    out = torch.full((M + 2, N), 7.0, dtype=xd.dtype, device="cuda")
    try:
        gemm_skinny(ol, xd, M, act, epi=L.EPI_STORE, out=out, ldo=N, ksplit=q4_force(*plan))
    except Exception as e:
        print(plan, (M,N,K), e); continue
    o = host(out)[:M]
    want = round_to(matmul_nt(x, wdense), act)
    err = np.abs(o - want)
    print(plan, (M,N,K), "nan", int(np.isnan(o).sum()), "bad", int((err > 0.05).sum() + np.isnan(err).sum()))
    np.set_printoptions(linewidth=200, precision=3, suppress=True)
    print(" got ", o[0, :20]); print(" want", want[0, :20]); print(" got r5", o[min(5,M-1), :20]); print(" want  ", want[min(5,M-1), :20])
