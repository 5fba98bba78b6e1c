"""Time the SigLIP attention backward at the training shape (96 images): fused kernel vs GEMM formulation."""
import sys, torch
sys.path.insert(0, ".")
from kai0_amd import ops
from tools.bench_gemm import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
n, S, NH, HD = 96, 256, 16, 72
E = NH * HD
q, k, v, do = (torch.randn(n * S, E, device=dev).to(BF16) for _ in range(4))
for mode in (True, False):
    ops._SIGLIP_BWD_FUSED = mode
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops.siglip_attention(qq, kk, vv, n, S, NH, HD)
    ms = timeit(lambda: torch.autograd.grad(o, (qq, kk, vv), do, retain_graph=True), iters=10, warm=3)
    print(f"siglip attention backward, fused={mode}: {ms:.3f} ms")
