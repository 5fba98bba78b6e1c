"""Round-4 attention A/B at the training shapes (B = 32): stored-P (round 3) against recompute (one-pass forward + lse, backward
recomputes P), Gemma joint attention (H = 8, HD = 256, P = 968, Hs = 50) and SigLIP (96 images, 16 heads x 72, 256 tokens).
Prints ms per layer for forward / backward and the relative difference of outputs / gradients between the two.
usage: python tools/attn_r4_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402
from kai0_amd.model import build_mask_codes  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def timed(fwd, bwd, n=8):
    tf, tb = [], []
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        o = fwd()
        e[1].record()
        bwd(o)
        e[2].record()
        torch.cuda.synchronize()
        tf.append(e[0].elapsed_time(e[1]))
        tb.append(e[1].elapsed_time(e[2]))
    return sorted(tf)[n // 2], sorted(tb)[n // 2]


def gemma():
    H, HD, P, Hs = 8, 256, 968, 50
    S = P + Hs
    pad = torch.ones((B, S), dtype=torch.bool, device=dev)
    pad[:, 768 + 100 : P] = False
    att = torch.zeros((B, S), dtype=torch.bool, device=dev)
    att[:, P] = True
    qcode, kcode, pos = build_mask_codes(pad, att)
    inv = (1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.int64).float() / HD))).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    flat = []
    for L in (P, Hs):
        flat += [torch.randn(B * L, w, device=dev, generator=g).to(BF16).requires_grad_(True) for w in (H * HD, HD, HD)]
    douts = [torch.randn(B * L, H * HD, device=dev, generator=g).to(BF16) for L in (P, Hs)]
    res = {}
    for name, store in (("stored P (r3)", True), ("recompute (r4)", False)):
        ops._ATTN_STORE_P = store

        def fwd():
            for t in flat:
                t.grad = None
            return ops.joint_attention(pos, qcode, kcode, inv, H, HD, (P, Hs), flat)

        f, b = timed(fwd, lambda o: torch.autograd.backward(list(o), douts))
        outs = fwd()
        torch.autograd.backward(list(outs), douts)
        res[name] = ([o.detach().clone() for o in outs], [t.grad.clone() for t in flat])
        print(f"gemma  {name:16s} fwd {f:7.3f} ms  bwd {b:7.3f} ms  x18 = {18 * (f + b):6.1f} ms/step", flush=True)
    a, r = res["recompute (r4)"], res["stored P (r3)"]
    print("gemma  recompute vs stored: out", " ".join(f"{rel(x, y):.2e}" for x, y in zip(a[0], r[0])), "grads",
          " ".join(f"{rel(x, y):.2e}" for x, y in zip(a[1], r[1])), flush=True)


def siglip():
    n, S, NH, HD = 3 * B, 256, 16, 72
    E = NH * HD
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v = (torch.randn(n * S, E, device=dev, generator=g).to(BF16).requires_grad_(True) for _ in range(3))
    do = torch.randn(n * S, E, device=dev, generator=g).to(BF16)
    res = {}
    for name, store in (("stored P (r3)", True), ("recompute (r4)", False)):
        ops._ATTN_STORE_P = store

        def fwd():
            for t in (q, k, v):
                t.grad = None
            return ops.siglip_attention(q, k, v, n, S, NH, HD)

        f, b = timed(fwd, lambda o: o.backward(do))
        o = fwd()
        o.backward(do)
        res[name] = (o.detach().clone(), [t.grad.clone() for t in (q, k, v)])
        print(f"siglip {name:16s} fwd {f:7.3f} ms  bwd {b:7.3f} ms  x27 = {27 * (f + b):6.1f} ms/step", flush=True)
    a, r = res["recompute (r4)"], res["stored P (r3)"]
    print("siglip recompute vs stored: out", f"{rel(a[0], r[0]):.2e}", "grads", " ".join(f"{rel(x, y):.2e}" for x, y in zip(a[1], r[1])), flush=True)


which = os.environ.get("ATTN_BENCH", "gemma,siglip").split(",")
if "gemma" in which:
    gemma()
if "siglip" in which:
    siglip()
ops._ATTN_STORE_P = False
