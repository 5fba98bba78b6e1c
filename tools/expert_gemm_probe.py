"""Split-K / tile choice for the action-expert GEMMs of the training step (M = 1600 rows): us per call by split."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import _lib, ops
BF16 = torch.bfloat16
dev = torch.device("cuda:0")

def timed(fn, iters=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3

cases = [("NT", 1600, 1024, 4096), ("NT", 1600, 4096, 1024), ("NT", 1600, 2560, 1024), ("NT", 1600, 1024, 2048),
         ("TN", 4096, 1024, 1600), ("TN", 1024, 4096, 1600), ("TN", 2560, 1024, 1600), ("TN", 1024, 2048, 1600),
         ("NN", 1600, 1024, 2560), ("NN", 1600, 2048, 1024)]
for lay, M, N, K in cases:
    if lay == "NT":
        a, b = torch.randn(M, K, device=dev).to(BF16), torch.randn(N, K, device=dev).to(BF16); kw = dict(lda=K, ldb=K)
    elif lay == "TN":
        a, b = torch.randn(K, M, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16); kw = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
    else:
        a, b = torch.randn(M, K, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16); kw = dict(a_kc=True, b_kc=False, lda=K, ldb=N)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    line = f"{lay} {M}x{N}x{K}: default split {ops.pick_split_k(M, N, K) if lay != 'TN' else ops.pick_split_k_wgrad(M, N, K)} |"
    for cfg in (0, 9):
        _lib.load().kai0_gemm_set_cfg(cfg)
        for sp in (1, 2, 3, 4, 6, 8):
            if K // sp < 128: continue
            try:
                us = timed(lambda: ops.gemm(a, b, out, M=M, N=N, K=K, ldc=N, split_k=sp, **kw))
                line += f" c{cfg}s{sp}:{us:5.1f}"
            except Exception as ex:
                line += f" c{cfg}s{sp}:err"
    _lib.load().kai0_gemm_set_cfg(0)
    print(line, flush=True)
