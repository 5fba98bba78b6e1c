"""Reference ceiling per shape (CDNA guide rule 10: "never infer a platform ceiling from your own failed attempts"): the vendor library
(torch.matmul -> hipBLASLt / rocBLAS, whatever torch picks) against kai0_gemm_bf16 on the training step's top shapes, plain epilogues,
random data, interleaved rounds in one process.  The vendor library is NOT on the product path; this is a measurement only.
usage: python tools/probes/vendor_gemm_reference.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=6, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


CASES = [  # (name, layout, M, N, K): NT = x[M,K] w[N,K]^T ; TN = a[K,M]^T b[K,N]  (weight gradients)
    ("gemma down fwd / gate dgrad", "NT", 30976, 2048, 16384), ("gemma gate fwd (plain)", "NT", 30976, 16384, 2048),
    ("gemma qkv fwd", "NT", 30976, 2560, 2048), ("gemma o_proj", "NT", 30976, 2048, 2048),
    ("gemma gate/up wgrad", "TN", 16384, 2048, 30976), ("gemma down wgrad", "TN", 2048, 16384, 30976), ("gemma qkv wgrad", "TN", 2560, 2048, 30976),
    ("siglip fc1 fwd (plain)", "NT", 24576, 4304, 1152), ("siglip fc2 fwd", "NT", 24576, 1152, 4304), ("siglip out_proj", "NT", 24576, 1152, 1152),
    ("siglip qkv", "NT", 24576, 3456, 1152), ("siglip fc1 wgrad", "TN", 4304, 1152, 24576), ("siglip fc2 wgrad", "TN", 1152, 4304, 24576),
    ("siglip out wgrad", "TN", 1152, 1152, 24576), ("square 8192", "NT", 8192, 8192, 8192),
]
print(f"{'shape':30s} {'layout':6s} {'M':>6s} {'N':>6s} {'K':>6s}   kai0 TF/s  vendor TF/s  kai0/vendor")
for name, lay, M, N, K in CASES:
    out = torch.empty(M, N, dtype=BF16, device=dev)
    if lay == "NT":
        a, b = torch.randn(M, K, device=dev).to(BF16), (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        ours = lambda: ops.gemm(a, b, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)  # noqa: E731
        vend = lambda: torch.matmul(a, b.t(), out=out)  # noqa: E731
    else:
        a, b = torch.randn(K, M, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
        sk = ops.pick_split_k_wgrad(M, N, K)
        ours = lambda: ops.gemm(a, b, out, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N, split_k=sk)  # noqa: E731
        vend = lambda: torch.matmul(a.t(), b, out=out)  # noqa: E731
    to, tv = [], []
    for _ in range(3):  # interleaved rounds
        to.append(timeit(ours))
        tv.append(timeit(vend))
    fl = 2.0 * M * N * K / 1e9
    print(f"{name:30s} {lay:6s} {M:6d} {N:6d} {K:6d}   {fl / min(to):9.0f}  {fl / min(tv):10.0f}  {min(tv) / min(to):8.2f}", flush=True)
