"""Yardstick only (not part of the product): the rate torch.matmul (hipBLASLt) reaches on the prefill GEMM shapes,
to put the hand-written tile GEMM's TFLOP/s in proportion.  Run on the GPU box."""
import torch, time
torch.manual_seed(0)
for (M,K,N) in [(8192,4096,28672),(8192,4096,6144),(8192,14336,4096),(8192,4096,4096)]:
    a=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); b=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)
    for _ in range(3): c=a@b.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c=a@b.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(M,K,N, f"{ms:.3f} ms  {2*M*K*N/ms/1e9:.0f} TFLOP/s", flush=True)
