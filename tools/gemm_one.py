"""Run one GEMM shape a few times (for rocprofv3 --pmc passes). usage: gemm_one.py NT|NN|TN M N K [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

lay, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
out = torch.empty(M, N, dtype=BF16, device=dev)
if lay == "NT":
    a, b = torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16)
    kw = dict(a_kc=True, b_kc=True, lda=K, ldb=K)
elif lay == "NN":
    a, b = torch.randn(M, K, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
    kw = dict(a_kc=True, b_kc=False, lda=K, ldb=N)
else:
    a, b = torch.randn(K, M, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
    kw = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
for _ in range(iters):
    ops.gemm(a, b, out, M=M, N=N, K=K, ldc=N, **kw)
torch.cuda.synchronize()
