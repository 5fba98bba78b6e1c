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
def run(lay,M,N,K,lda=None,ldb=None):
    out=torch.empty(M,N,dtype=BF16,device=dev)
    if lay=='NT':
        lda=lda or K; ldb=ldb or K
        a=torch.randn(M,lda,device=dev).to(BF16); b=torch.randn(N,ldb,device=dev).to(BF16); kw=dict(a_kc=True,b_kc=True)
    elif lay=='NN':
        lda=lda or K; ldb=ldb or N
        a=torch.randn(M,lda,device=dev).to(BF16); b=torch.randn(K,ldb,device=dev).to(BF16); kw=dict(a_kc=True,b_kc=False)
    else:
        lda=lda or M; ldb=ldb or N
        a=torch.randn(K,lda,device=dev).to(BF16); b=torch.randn(K,ldb,device=dev).to(BF16); kw=dict(a_kc=False,b_kc=False)
    ms=timeit(lambda: ops.gemm(a,b,out,M=M,N=N,K=K,lda=lda,ldb=ldb,ldc=N,**kw))
    print(f'{lay} M={M} N={N} K={K} lda={lda} ldb={ldb}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s',flush=True)
run('TN',16384,2048,30976)
run('TN',16384,2048,30976,lda=16384+64,ldb=2048+64)
run('TN',16384,2048,30976,lda=16384+8,ldb=2048+8)
run('TN',16384,2048,30976,lda=16384+128,ldb=2048+128)
run('NN',30976,2048,16384)
run('NN',30976,2048,16384,ldb=2048+64)
run('NN',30976,2048,16384,lda=16384+64,ldb=2048+64)
run('NT',30976,16384,2048)
run('NT',30976,16384,2048,lda=2048+64,ldb=2048+64)
run('NT',30976,2048,16384)
run('NT',30976,2048,16384,lda=16384+64,ldb=16384+64)
