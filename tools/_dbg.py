import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops
BF16=torch.bfloat16; dev=torch.device('cuda:0')
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters
M,N,K=30976,16384,2048
a=torch.randn(M,K,device=dev).to(BF16); b=torch.randn(N,K,device=dev).to(BF16); out=torch.empty(M,N,dtype=BF16,device=dev)
ms=timeit(lambda: ops.gemm(a,b,out,M=M,N=N,K=K,lda=K,ldb=K,ldc=N))
print(f"cfg={os.environ.get('KAI0_GEMM_CFG')} ablate={os.environ.get('KAI0_GEMM_ABLATE')} NT {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s-equivalent")
z=torch.zeros(M,K,device=dev).to(BF16); zb=torch.zeros(N,K,device=dev).to(BF16)
ms=timeit(lambda: ops.gemm(z,zb,out,M=M,N=N,K=K,lda=K,ldb=K,ldc=N))
print(f"   zero data: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f}")
