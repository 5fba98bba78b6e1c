"""Per-kernel cost of a chain of dependent trivial kernels replayed from a hipGraph, by grid size (x.add_(1) on n f32 elements:
torch launches ceil(n / 1024)-ish blocks of 256 threads).  usage: python tools/probes/launch_floor.py"""
import torch

dev = torch.device("cuda:0")
N = 2000
for n in (256, 16 * 1024, 64 * 1024, 256 * 1024, 1024 * 1024, 4 * 1024 * 1024):
    x = torch.zeros(n, device=dev)
    for _ in range(3):
        x.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            x.add_(1.0)
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
    # eager, same chain
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(N):
        x.add_(1.0)
    e.record()
    e.synchronize()
    print(f"n = {n:8d} f32 ({n * 4 / 1024:8.0f} KiB): graph {min(ts) / N * 1e3:6.2f} us per kernel, eager {s.elapsed_time(e) / N * 1e3:6.2f} us")

# the same chain with K different kernels alternating (is it the kernel SWITCH that costs? instruction cache, descriptors)
ops = [lambda x: x.add_(1.0), lambda x: x.mul_(1.0001), lambda x: x.neg_(), lambda x: x.abs_(), lambda x: x.sin_(), lambda x: x.cos_(),
       lambda x: x.exp_(), lambda x: x.sqrt_(), lambda x: x.tanh_(), lambda x: x.sigmoid_(), lambda x: x.relu_(), lambda x: x.floor_(),
       lambda x: x.clamp_(-1, 1), lambda x: x.erf_(), lambda x: x.atan_(), lambda x: x.round_()]
for n in (256, 64 * 1024):
    for K in (1, 2, 4, 8, 16):
        x = torch.zeros(n, device=dev)
        for o in ops[:K]:
            o(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(N):
                ops[i % K](x)
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
        print(f"n = {n:6d}, {K:2d} different kernels alternating: graph {min(ts) / N * 1e3:6.2f} us per kernel")
