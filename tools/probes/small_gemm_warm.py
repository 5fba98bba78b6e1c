"""Does the Infinity Cache keep the weights of the B = 1 GEMMs?  One GEMM shape in a graph-replayed chain cycling over L different
weight matrices: us per launch by L (L = 1: always the same, warm; large L: cold)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def chain(M, N, K, L, reps=32):
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF16) for _ in range(L)]
    x = torch.randn(M, K, device=dev).to(BF16)
    out = torch.empty(M, N, device=dev, dtype=BF16)

    def run():
        for i in range(reps):
            ops.gemm(x, ws[i % L], out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)

    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts) / reps * 1e3


for name, M, N, K in [("siglip out", 768, 1152, 1152), ("siglip fc1", 768, 4304, 1152), ("prefix o", 968, 2048, 2048), ("prefix gate", 968, 16384, 2048)]:
    line = f"{name:12s} {M}x{N}x{K} (W {N * K * 2 / 1e6:5.1f} MB): "
    for L in (1, 2, 4, 8, 16, 32):
        line += f" L={L}: {chain(M, N, K, L):6.1f}"
    print(line + "  us per launch", flush=True)
