import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from kai0_amd import ops
from test_kernels_gpu import rnd, dev, rel_err
for (M,N,K) in [(144,16,64),(20,16,64),(144,128,64),(20,128,64),(144,64,128),(20,64,128),(96,64,64),(96,136,64),(96,64,136),(144,64,64),(20,32,64),(8,16,8),(16,16,16),(24,8,8)]:
    x=rnd(M,K,seed=1).requires_grad_(True); w=rnd(N,K,seed=2,scale=0.1).requires_grad_(True); dy=rnd(M,N,seed=3)
    out=ops.linear(x,w); out.backward(dy)
    xr,wr=(t.detach().float().requires_grad_(True) for t in (x,w)); ref=xr@wr.t(); ref.backward(dy.float())
    print((M,N,K),f'out={rel_err(out,ref):.2e} dx={rel_err(x.grad,xr.grad):.2e} dw={rel_err(w.grad,wr.grad):.2e}')
