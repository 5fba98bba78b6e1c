"""Joint (Gemma) attention forward / backward at the training shape (B = 32, H = 8, HD = 256, P = 968, Hs = 50): ms per call for
the variants of the backward.  usage: python tools/attn_bwd_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402
from kai0_amd.model import build_mask_codes  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, HD, P, Hs = 8, 256, 968, 50
S = P + Hs
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
pad = torch.ones((B, S), dtype=torch.bool, device=dev)
pad[:, 768 + 100 : P] = False
att = torch.zeros((B, S), dtype=torch.bool, device=dev)
att[:, P] = True
qcode, kcode, pos = build_mask_codes(pad, att)
inv = (1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.int64).float() / HD))).to(dev)
flat = []
for L in (P, Hs):
    flat += [torch.randn(B * L, H * HD, device=dev).to(BF16).requires_grad_(True), torch.randn(B * L, HD, device=dev).to(BF16).requires_grad_(True),
             torch.randn(B * L, HD, device=dev).to(BF16).requires_grad_(True)]  # fmt: skip
douts = [torch.randn(B * P, H * HD, device=dev).to(BF16), torch.randn(B * Hs, H * HD, device=dev).to(BF16)]


def run(n=6):
    tf, tb = [], []
    for _ in range(n):
        for t in flat:
            t.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        outs = ops.joint_attention(pos, qcode, kcode, inv, H, HD, (P, Hs), flat)
        e[1].record()
        torch.autograd.backward(list(outs), douts)
        e[2].record()
        torch.cuda.synchronize()
        tf.append(e[0].elapsed_time(e[1]))
        tb.append(e[1].elapsed_time(e[2]))
    return sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2], [t.grad.clone() for t in flat]


variants = [("old (NN dq, split 1)", False, "1"), ("NT dq, split 1", True, "1"), ("NT dq, split auto", True, "auto"), ("NT dq, split 3", True, "3")]
if hasattr(ops, "_ATTN_BWD_FUSED"):
    variants.append(("fused kernels", True, "auto"))
base = None
for name, nt, sp in variants:
    ops._ATTN_BWD_NT, ops._ATTN_BWD_SPLIT = nt, sp
    if hasattr(ops, "_ATTN_BWD_FUSED"):
        ops._ATTN_BWD_FUSED = name.startswith("fused")
    f, b, g = run()
    if base is None:
        base = g
    err = max(float((a.float() - c.float()).norm() / (c.float().norm() + 1e-9)) for a, c in zip(g, base))
    print(f"{name:24s} fwd {f:7.3f} ms  bwd {b:7.3f} ms  (x18 layers: {18 * (f + b):6.1f} ms/step)  max rel diff vs first {err:.2e}", flush=True)
