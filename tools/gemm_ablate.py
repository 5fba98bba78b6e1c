"""GEMM limiter probes: KAI0_GEMM_ABLATE=1 (no DMA in the K loop) and leading-dimension padding (channel conflicts).
Needs a library built with the ablation hooks: KAI0_HIPCC_FLAGS=-DKAI0_ABLATE python -m kai0_amd.build --force (round 5: the shipped
kernels carry none)."""
import os, sys, subprocess
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ".")
    from kai0_amd import ops
    from tools.bench_gemm import timeit
    dev = torch.device("cuda:0"); BF16 = torch.bfloat16
    cases = [("NT", 8192, 8192, 8192, 0), ("NT", 30976, 16384, 2048, 0), ("NT", 30976, 2048, 16384, 0),
             ("NT", 30976, 2048, 2048, 0), ("NT", 24576, 4304, 1152, 0)]
    for (lay, M, N, K, pad) in cases:
        out = torch.empty(M, N, dtype=BF16, device=dev)
        if lay == "NT":
            a, b = torch.randn(M, K + pad, device=dev).to(BF16), torch.randn(N, K + pad, device=dev).to(BF16)
            kw = dict(a_kc=True, b_kc=True, lda=K + pad, ldb=K + pad)
        else:
            a, b = torch.randn(K, M + pad, device=dev).to(BF16), torch.randn(K, N + pad, device=dev).to(BF16)
            kw = dict(a_kc=False, b_kc=False, lda=M + pad, ldb=N + pad)
        ms = timeit(lambda: ops.gemm(a, b, out, M=M, N=N, K=K, ldc=N, **kw), iters=8, warm=4)
        print(f"{sys.argv[1]:8s} {lay} {M}x{N}x{K} pad {pad:3d}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
else:
    for name, extra in (("default", {}), ("cfg8-384", {"KAI0_GEMM_CFG": "8"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **extra))
