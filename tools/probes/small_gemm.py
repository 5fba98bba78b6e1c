"""B = 1 inference GEMMs (M = 768 / 968 rows) in a graph-replayed chain over L different weight matrices (cold weights, as in the
SigLIP tower / prefix pass): us per launch by split-K.  usage: python tools/probes/small_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
L = 27


def chain(M, N, K, split, bias=True):
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF16) for _ in range(L)]
    b = torch.randn(N, device=dev).to(BF16) if bias else None
    x = torch.randn(M, K, device=dev).to(BF16)
    out = torch.empty(M, N, device=dev, dtype=BF16)

    def run():
        for w in ws:
            ops.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, split_k=split)

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
    return min(ts) / L * 1e3


for name, M, N, K in [("siglip qkv", 768, 3456, 1152), ("siglip out", 768, 1152, 1152), ("siglip fc1", 768, 4304, 1152), ("siglip fc2", 768, 1152, 4304),
                      ("prefix qkv", 968, 2560, 2048), ("prefix o", 968, 2048, 2048), ("prefix gate", 968, 16384, 2048), ("prefix down", 968, 2048, 16384)]:
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    line = f"{name:12s} {M}x{N}x{K} ({tiles:4d} tiles of 128^2; {2.0 * M * N * K / 1e9:5.1f} GF, W {N * K * 2 / 1e6:5.1f} MB): "
    for split in (1, 2, 3, 4, 6, 8):
        if K // split < 256:
            continue
        us = chain(M, N, K, split)
        line += f" s{split} {us:6.1f}"
    print(line + "  us (incl. reduce launch)", flush=True)
