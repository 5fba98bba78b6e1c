"""NT GEMM probe for schedule experiments (env switches are read once per process): TF/s + a checksum per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [("gate", 30976, 16384, 2048), ("down", 30976, 2048, 16384), ("q/o", 30976, 2048, 2048), ("fc1", 24576, 4304, 1152),
          ("fc2", 24576, 1152, 4304), ("sq8k", 8192, 8192, 8192)]  # fmt: skip
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("KAI0_GEMM"))
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    f = lambda: ops.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)  # noqa: E731
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
    print(f"[{tag or 'default'}] {name:5s} {M}x{N}x{K}: {ms:7.3f} ms {2 * M * N * K / ms / 1e9:7.1f} TF/s  checksum {float(out.float().abs().sum()):.6e}", flush=True)
    del x, w, out
