"""Does the 8608-byte row stride of SigLIP's 4304-wide activations (not a multiple of the 128-B line) cost GEMM throughput?
Same problems with the activation leading dimension 4304 and padded to 4352."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
M, D, F = 24576, 1152, 4304


def t(name, f, flops):
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
    print(f"{name:44s} {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF/s", flush=True)


x = torch.randn(M, D, device=dev).to(BF16)
w1 = (torch.randn(F, D, device=dev) * 0.05).to(BF16)
w2 = (torch.randn(D, F, device=dev) * 0.05).to(BF16)
dy = torch.randn(M, D, device=dev).to(BF16)
fl = 2 * M * D * F
for ld in (4304, 4352, 4608):
    h = torch.randn(M, ld, device=dev).to(BF16)
    out = torch.empty(M, D, dtype=BF16, device=dev)
    dw1 = torch.empty(F, D, dtype=BF16, device=dev)
    dw2 = torch.empty(D, F, dtype=BF16, device=dev)
    t(f"fc1 fwd  NT [M,{D}]x[{F},{D}] -> ldc={ld}", lambda: ops.gemm(x, w1, h, M=M, N=F, K=D, lda=D, ldb=D, ldc=ld), fl)
    t(f"fc2 fwd  NT lda={ld}", lambda: ops.gemm(h, w2, out, M=M, N=D, K=F, lda=ld, ldb=F, ldc=D), fl)
    t(f"fc1 wgrad TN A=dpre lda={ld}", lambda: ops.gemm(h, x, dw1, M=F, N=D, K=M, a_kc=False, b_kc=False, lda=ld, ldb=D, ldc=D, split_k=ops.pick_split_k(F, D, M)), fl)
    t(f"fc2 wgrad TN B=h ldb={ld}", lambda: ops.gemm(dy, h, dw2, M=D, N=F, K=M, a_kc=False, b_kc=False, lda=D, ldb=ld, ldc=F, split_k=ops.pick_split_k(D, F, M)), fl)
