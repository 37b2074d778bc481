"""gemm_skinny.hip on quantised weights, four back-to-back launches per shape: how many elements are wrong per run (the first
run sees cold caches: a wait that is too loose shows there first) and whether the runs agree bit for bit."""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import ref_quant
from oracle.numerics import matmul_nt, round_to
from mlx_parallm_amd import _lib as L
from gpu_helpers import dev, dev_u32, gemm_skinny, host, op_linear, to_tiled
rng=np.random.default_rng(1)
def run(M,N,K,ks,kind="q4_bf16",act="bfloat16"):
    bits=4 if kind.startswith("q4") else 8
    w=rng.standard_normal((N,K)).astype(np.float32)*0.05
    packed,scales,biases=ref_quant.quantize(round_to(w,act),64,bits,act)
    pd,sd,bd=dev_u32(packed),dev(scales,act),dev(biases,act)
    ol=op_linear(kind,N,K,pd,sd,bd); keep=[pd,sd,bd]; assert to_tiled(ol,keep)
    wd=ref_quant.dequantize(packed,scales,biases,64,bits)
    x=round_to(rng.standard_normal((M,K)).astype(np.float32),act); xd=dev(x,act)
    want=round_to(matmul_nt(x,wd),act)
    outs=[]
    for r in range(4):
        out=torch.full((M+2,N),7.0,dtype=xd.dtype,device="cuda")
        used,_=gemm_skinny(ol,xd,M,act,epi=L.EPI_STORE,out=out,ldo=N,ksplit=ks)
        torch.cuda.synchronize(); outs.append(host(out)[:M])
    bad=[np.abs(o-want)>0.05*np.abs(want).max() for o in outs]
    print(f"M{M} N{N} K{K} ks{ks} {kind}: used {used}; wrong elems per run {[int(b.sum()) for b in bad]}; runs equal {[bool(np.array_equal(outs[0],o)) for o in outs[1:]]}")
    b=bad[0]
    if b.any():
        rows=np.where(b.any(axis=1))[0]; cols=np.where(b.any(axis=0))[0]
        print("   rows",rows[:20],"... cols tiles",sorted(set((cols//16).tolist())))
for args in [(40,144,4608,5),(40,144,4608,1),(40,144,1024,1),(40,144,512,1),(16,144,4608,5),(32,144,4608,5),(40,144,4608,5,"q8_bf16"),(64,144,4608,5)]:
    run(*args)
