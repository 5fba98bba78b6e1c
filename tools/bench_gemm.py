"""GEMM micro-benchmark over the pi0.5 shapes (run on the GPU box): TFLOP/s per layout, random data."""

import json
import sys

import torch

sys.path.insert(0, ".")
from kai0_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    Mp = B * 968
    shapes = [
        ("gemma.q/o  NT", Mp, 2048, 2048), ("gemma.gate NT", Mp, 16384, 2048), ("gemma.down NT", Mp, 2048, 16384),
        ("siglip.qkv NT", 3 * B * 256, 1152, 1152), ("siglip.fc1 NT", 3 * B * 256, 4304, 1152),
        ("siglip.fc2 NT", 3 * B * 256, 1152, 4304), ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192),
    ]  # fmt: skip
    res = []
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ms = timeit(lambda: ops.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N))
        tf = 2 * M * N * K / ms / 1e9
        res.append((name, M, N, K, ms, tf))
        print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}  {ms:8.3f} ms  {tf:7.1f} TF/s", flush=True)
        # dgrad (NN) and wgrad (TN) of the same Linear
        dy = torch.randn(M, N, device=dev).to(BF16)
        dx = torch.empty(M, K, dtype=BF16, device=dev)
        ms = timeit(lambda: ops.gemm(dy, w, dx, M=M, N=K, K=N, a_kc=True, b_kc=False, lda=N, ldb=K, ldc=K))
        print(f"{'  dgrad NN':16s} {'':27s}  {ms:8.3f} ms  {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
        dw = torch.empty(N, K, dtype=BF16, device=dev)
        ms = timeit(lambda: ops.gemm(dy, x, dw, M=N, N=K, K=M, a_kc=False, b_kc=False, lda=N, ldb=K, ldc=K))
        print(f"{'  wgrad TN':16s} {'':27s}  {ms:8.3f} ms  {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
        del x, w, out, dy, dx, dw
    # rocBLAS/hipBLASLt through torch for orientation only (not used by the product)
    x = torch.randn(8192, 8192, device=dev).to(BF16)
    w = torch.randn(8192, 8192, device=dev).to(BF16)
    ms = timeit(lambda: torch.matmul(x, w.t()))
    print(f"torch.matmul 8192^3 (vendor BLAS, orientation only): {ms:.3f} ms {2 * 8192**3 / ms / 1e9:.1f} TF/s")
    json.dump(res, open("gpurun_out/bench_gemm.json", "w"))


if __name__ == "__main__":
    main()
