"""f32 GEMM shapes of the path: patch embed (fwd, wgrad), adaRMS dense (fwd, dgrad, wgrad): us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
F32 = torch.float32


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
    us = s.elapsed_time(e) / 10 * 1e3
    print(f"{name:40s} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s", flush=True)


for M, N, K, name in ((24576, 1152, 588, "patch embed B=32"), (768, 1152, 588, "patch embed B=1"), (32, 3072, 1024, "adaRMS dense")):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    dw = torch.empty(N, K, device=dev)
    t(f"{name} fwd  x W^T", lambda: ops.gemm_f32(x, K, 1, w, 1, K, out, M, N, K, bias=b), 2 * M * N * K)
    t(f"{name} dgrad dy W", lambda: ops.gemm_f32(dy, N, 1, w, K, 1, dx, M, K, N), 2 * M * N * K)
    t(f"{name} wgrad dy^T x", lambda: ops.gemm_f32(dy, 1, N, x, K, 1, dw, N, K, M), 2 * M * N * K)
