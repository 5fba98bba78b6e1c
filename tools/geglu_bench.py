"""The three N=16384, K=2048 GEMM variants of the Gemma MLP (plain gate, GeGLU forward, GeGLU backward): ms and TF/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
M, F, D = 30976, 16384, 2048
x = torch.randn(M, D, device=dev).to(BF16)
w = (torch.randn(F, D, device=dev) * 0.02).to(BF16)
g = torch.randn(M, F, device=dev).to(BF16)
u = torch.randn(M, F, device=dev).to(BF16)
o1 = torch.empty(M, F, dtype=BF16, device=dev)
o2 = torch.empty(M, F, dtype=BF16, device=dev)


def t(name, f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        f()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name:28s} {ms:7.3f} ms  {2 * M * F * D / ms / 1e9:7.1f} TF/s", flush=True)


t("plain", lambda: ops.gemm(x, w, o1, M=M, N=F, K=D, lda=D, ldb=D, ldc=F))
t("act2 (reads g; writes u,h)", lambda: ops.gemm(x, w, o1, M=M, N=F, K=D, lda=D, ldb=D, ldc=F, act=2, pre_out=o2, aux1=g, split_k=1))
t("act2 no pre_out", lambda: ops.gemm(x, w, o1, M=M, N=F, K=D, lda=D, ldb=D, ldc=F, act=2, aux1=g, split_k=1))
t("act3 (reads g,u; writes dg,du)", lambda: ops.gemm(x, w, o1, M=M, N=F, K=D, lda=D, ldb=D, ldc=F, act=3, pre_out=o2, aux1=g, aux2=u))
