"""Masked softmax at the training attention shape: achieved bandwidth (logits read + probs written, bf16)."""
import sys, torch
sys.path.insert(0, ".")
from kai0_amd import ops, _lib
from kai0_amd.model import build_mask_codes
from tools.bench_gemm import timeit
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
B, S, H, S_ld = 32, 1018, 8, 1024
M = S * H
x = torch.randn(B, M, S_ld, device=dev).to(BF16)
pad = torch.ones(B, S, dtype=torch.bool, device=dev); pad[:, 900:968] = False
att = torch.zeros(B, S, dtype=torch.bool, device=dev); att[:, 968] = True
qc, kc, _ = build_mask_codes(pad, att)
y = torch.empty_like(x)
f = lambda: _lib.call("kai0_softmax_mask_fwd", x.data_ptr(), y.data_ptr(), qc.data_ptr(), kc.data_ptr(), B, S, H, S, S_ld, M * S_ld, 0, qc.stride(0), kc.stride(0), ops._stream())
ms = timeit(f, iters=10, warm=3)
print(f"softmax_mask_fwd [{B}x{M}x{S_ld}]: {ms:.3f} ms  {2 * x.numel() * 2 / ms / 1e9:.2f} TB/s")
