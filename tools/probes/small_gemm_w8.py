"""The B = 1 passes' 128 x 128 GEMMs on four waves against eight (kai0_gemm_desc.small_w8; round 6): us per launch in a graph-replayed
chain that cycles over 16 different weight matrices (cold weights, as inside the action chunk), both arms in ONE process, alternating.
usage: python tools/probes/small_gemm_w8.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
L, REPS = 16, 32


def chain(M, N, K, mode, split, bias, res, act):
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF16) for _ in range(L)]
    x = torch.randn(M, K, device=dev).to(BF16)
    out = torch.empty(M, N, device=dev, dtype=BF16)
    kw = {}
    if bias:
        kw["bias"] = torch.randn(N, device=dev).to(BF16)
    if res:
        kw.update(residual=torch.randn(M, N, device=dev).to(BF16), ldr=N)
    if split > 1:
        kw["split_k"] = split

    def run():
        with ops.gemm_tuning(small_w8=mode):
            for i in range(REPS):
                ops.gemm(x, ws[i % L], out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=act, **kw)

    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    return g, (ws, x, out, kw)  # (the graph holds raw pointers: its operands must outlive it — torch.cuda.graph() empties the allocator's cache)


def timeit(g):
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts) / REPS * 1e3


CASES = [("siglip q|k|v", 768, 3456, 1152, 1, True, False, 0), ("siglip out_proj s3", 768, 1152, 1152, 3, True, True, 0),
         ("siglip out_proj s1", 768, 1152, 1152, 1, True, True, 0), ("siglip fc1 gelu", 768, 4304, 1152, 1, True, False, 1),
         ("siglip fc2 s4", 768, 1152, 4304, 4, True, True, 0), ("siglip fc2 s2", 768, 1152, 4304, 2, True, True, 0),
         ("prefix q|k|v", 968, 2560, 2048, 1, False, False, 0), ("prefix o_proj s2", 968, 2048, 2048, 2, False, True, 0),
         ("prefix o_proj s1", 968, 2048, 2048, 1, False, True, 0), ("projector", 768, 2048, 1152, 1, True, False, 0)]
for name, M, N, K, split, bias, res, act in CASES:
    (g4, keep4), (g8, keep8) = chain(M, N, K, 1, split, bias, res, act), chain(M, N, K, 2, split, bias, res, act)
    t4, t8 = [], []
    for _ in range(3):  # alternating passes
        t4.append(timeit(g4))
        t8.append(timeit(g8))
    print(f"{name:20s} {M}x{N}x{K} split {split}: four waves {min(t4):6.1f} us   eight waves {min(t8):6.1f} us   ({min(t8) / min(t4):.2f}x)"
          f"   [incl. the reduction launch when split]", flush=True)
