"""Does the power-of-two row stride of the contraction-strided operands cost the TN (weight-gradient) GEMM its DMA rate?
Same problem with lda / ldb padded by 0, 64, 128, 192 elements (a [K][M+pad] buffer of which M columns are used)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def t(f, flops, iters=8):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, flops / ms / 1e9


for (M, N, K) in ((16384, 2048, 30976), (2048, 16384, 30976), (2560, 2048, 30976), (4304, 1152, 24576)):
    for pa, pb in ((0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (192, 64), (32, 32)):
        a = torch.randn(K, M + pa, device=dev).to(BF16)
        b = torch.randn(K, N + pb, device=dev).to(BF16)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        sk = ops.pick_split_k_wgrad(M, N, K)
        ms, tf = t(lambda: ops.gemm(a, b, out, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M + pa, ldb=N + pb, ldc=N, split_k=sk), 2.0 * M * N * K)
        print(f"TN {M}x{N}x{K} split {sk} lda=M+{pa:3d} ldb=N+{pb:3d}: {ms:7.3f} ms {tf:7.1f} TF/s", flush=True)
        del a, b, out
