"""The same chain measurement with this library's own small kernels (launched through ctypes -> hipLaunchKernelGGL under stream
capture): is their 4.5-5 us in the action-chunk graph a property of the launch path or of the kernels?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = 1000
BF16, F32 = torch.bfloat16, torch.float32


def chain(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    print(f"{name:44s} {min(ts) / N * 1e3:6.2f} us per launch")


x = torch.zeros(50 * 32, device=dev)
v = torch.ones(50 * 32, device=dev)
chain("euler_step_ (1600 f32)", lambda: ops.euler_step_(x, v, -0.1))
a = torch.randn(50, 1024, device=dev)
chain("cast f32 -> bf16 (50 x 1024)", lambda: ops.cast(a, BF16))
t = torch.zeros(50 * 32, device=dev)
chain("torch add_ (1600 f32)", lambda: t.add_(1.0))
h = torch.randn(50, 1024, device=dev).to(BF16)
w = torch.zeros(1024, device=dev)
chain("rmsnorm (50 x 1024 bf16)", lambda: ops.rmsnorm(h, w, 1e-6))
A = torch.randn(50, 32, device=dev)
W = torch.randn(1024, 32, device=dev)
b = torch.zeros(1024, device=dev)
chain("linear_f32 50x1024x32 (action_in_proj)", lambda: ops.linear_f32(A, W, b))
X = torch.randn(50, 1024, device=dev).to(BF16)
Wq = (torch.randn(1024, 1024, device=dev) * 0.05).to(BF16)
o = torch.empty(50, 1024, device=dev, dtype=BF16)
chain("skinny in-block 50x1024x1024 (plain)", lambda: ops.skinny_gemm(X, Wq, M=50, N=1024, K=1024, lda=1024, ldw=1024, split_k=-1, segs=[(o, 1024, 0, 1024, 0)]))
Wd = (torch.randn(1024, 4096, device=dev) * 0.05).to(BF16)
X4 = torch.randn(50, 4096, device=dev).to(BF16)
chain("skinny in-block 50x1024x4096 (down_proj)", lambda: ops.skinny_gemm(X4, Wd, M=50, N=1024, K=4096, lda=4096, ldw=4096, split_k=-1, segs=[(o, 1024, 0, 1024, 0)]))
