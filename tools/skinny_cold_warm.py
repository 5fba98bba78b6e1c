"""How much of a B = 1 weight-streaming GEMM's time is the cold-weight latency?  The gate|up GeGLU GEMM of one expert layer
(M = 50, K = 1024, N = 2 x 4096) timed in a hipGraph of 36 launches, (a) cycling through 36 different weight buffers (605 MB:
every launch streams from HBM, as in the denoise loop) and (b) re-using one buffer (16.8 MB: warm in L2 / Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
M, K, F, NW = 50, 1024, 4096, 36
x = torch.randn(M, K, device=dev).to(BF16)
ws = [(torch.randn(2 * F, K, device=dev) * 0.05).to(BF16) for _ in range(NW)]
h = torch.zeros(M, F, dtype=BF16, device=dev)


PF = False


def chain(weights):
    for i, w in enumerate(weights):
        nxt = weights[(i + 1) % len(weights)]
        kw = dict(prefetch=dict(W=nxt, N=2 * F, K=K, ldw=K, pair_stride=F)) if PF else {}  # PF needs the experimental build
        ops.skinny_gemm(x, w, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, segs=[(h, F, 0, F, 0)], **kw)


CASES = (("cold (36 buffers)", ws), ("warm (1 buffer)", [ws[0]] * NW), ("4 buffers (67 MB: Infinity Cache only)", (ws[:4] * 9)),
         ("9 buffers (151 MB)", ws[:9] * 4))
if os.environ.get("KAI0_PF_PROBE") is not None:  # PMC run: one variant only, no graph
    CASES = (CASES[int(os.environ["KAI0_PF_PROBE"])],)  # 0 cold, 1 warm, 2 four buffers
for name, weights in CASES:
    if isinstance(weights, str):
        PF, weights = True, ws
    else:
        PF = False
    chain(weights)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            chain(weights)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:40s}: {a.elapsed_time(b) / 20 / NW * 1e3:6.2f} us per launch")
