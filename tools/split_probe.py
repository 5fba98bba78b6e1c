"""Sweep the split-K factor of kai0_gemm_bf16 over a list of (layout, M, N, K) problems: us per launch for each split and the
split ops.pick_split_k would choose.  usage: split_probe.py train_expert | infer_prefix | infer_siglip"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
SETS = {
    "train_expert": [("NT", 1600, 1024, 4096), ("NT", 1600, 4096, 1024), ("NT", 1600, 2560, 1024), ("NT", 1600, 1024, 2048),
                     ("NN", 1600, 1024, 2560), ("NN", 1600, 2048, 1024), ("NN", 1600, 1024, 4096), ("NN", 1600, 4096, 1024),
                     ("TN", 4096, 1024, 1600), ("TN", 1024, 4096, 1600), ("TN", 2560, 1024, 1600), ("TN", 1024, 2048, 1600)],
    "infer_prefix": [("NT", 968, 2560, 2048), ("NT", 968, 2048, 2048), ("NT", 968, 16384, 2048), ("NT", 968, 2048, 16384)],
    "infer_siglip": [("NT", 768, 3456, 1152), ("NT", 768, 1152, 1152), ("NT", 768, 4304, 1152), ("NT", 768, 1152, 4304),
                     ("NT", 768, 2048, 1152)],
}  # fmt: skip
for lay, M, N, K in SETS[sys.argv[1]]:
    out = torch.empty(M, N, dtype=BF16, device=dev)
    if lay == "NT":
        a, b = torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16)
        kw = dict(a_kc=True, b_kc=True, lda=K, ldb=K)
    elif lay == "NN":
        a, b = torch.randn(M, K, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
        kw = dict(a_kc=True, b_kc=False, lda=K, ldb=N)
    else:
        a, b = torch.randn(K, M, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
        kw = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
    line = f"{lay} {M}x{N}x{K} (auto {ops.pick_split_k(M, N, K)}):"
    best = None
    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
        if K // sp < 128:
            continue
        f = lambda: ops.gemm(a, b, out, M=M, N=N, K=K, ldc=N, split_k=sp, **kw)  # noqa: E731
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        best = min(best or (us, sp), (us, sp))
        line += f" s{sp}:{us:5.1f}"
    print(line + f"  best s{best[1]}", flush=True)
