"""TN wgrad shapes of SigLIP (few output tiles, long contraction): time vs split-K factor."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
K = 24576
for Mo, No in ((1152, 4304), (4304, 1152), (3456, 1152), (1152, 1152), (2560, 2048), (2048, 2048)):
    Kc = 30976 if Mo in (2560, 2048) else K
    a = torch.randn(Kc, Mo, device=dev).to(BF16)
    b = torch.randn(Kc, No, device=dev).to(BF16)
    out = torch.empty(Mo, No, dtype=BF16, device=dev)
    line = f"TN {Mo}x{No}x{Kc} (auto split {ops.pick_split_k(Mo, No, Kc)}):"
    for sp in (1, 2, 3, 4, 6, 8, 12):
        f = lambda: ops.gemm(a, b, out, M=Mo, N=No, K=Kc, a_kc=False, b_kc=False, lda=Mo, ldb=No, ldc=No, split_k=sp)  # noqa: E731
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8):
            f()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 8
        line += f"  s{sp}: {ms * 1e3:5.0f}us {2 * Mo * No * Kc / ms / 1e9:5.0f}TF"
    print(line, flush=True)
