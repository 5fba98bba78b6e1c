"""Within-process A/B of the NT GEMM schedules (kai0_gemm_set_cfg): 0 = automatic, 5 = two-buffer ping-pong, 9 / 10 = the
quadrant schedule with the DMA pieces after the fragment reads / between the MFMAs.  Interleaved rounds, median TF/s, random
data; every variant's output must equal the default's bit for bit (same accumulation order).
usage: python tools/gemm_sched_ab.py [rounds]   -> gpurun_out/gemm_sched_ab.json"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import _lib, ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CFGS = [int(c) for c in os.environ.get("AB_CFGS", "5,9,10").split(",")]


def set_cfg(c):
    _lib.load().kai0_gemm_set_cfg(c)


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


cases = [
    ("gate  NT 30976x16384x2048", 30976, 16384, 2048, 0),
    ("up+geglu (act 2)         ", 30976, 16384, 2048, 2),
    ("dh+geglu bwd (act 3)     ", 30976, 16384, 2048, 3),
    ("down  NT 30976x2048x16384", 30976, 2048, 16384, 0),
    ("o/q   NT 30976x2048x2048 ", 30976, 2048, 2048, 0),
    ("fc1   NT 24576x4304x1152 ", 24576, 4304, 1152, 0),
    ("fc1 + bias + GELU (act 1) ", 24576, 4304, 1152, 1),
    ("fc2   NT 24576x1152x4304 ", 24576, 1152, 4304, 0),
    ("B=1 gate NT 968x16384x2048", 968, 16384, 2048, 0),
    ("square 8192              ", 8192, 8192, 8192, 0),
]
if os.environ.get("AB_LAYOUT", "NT") != "NT":
    lay = os.environ["AB_LAYOUT"]
    cases = [(f"{lay} wgrad gate 16384x2048x30976", 16384, 2048, 30976, 0), (f"{lay} wgrad down 2048x16384x30976", 2048, 16384, 30976, 0),
             (f"{lay} wgrad qkv 2560x2048x30976   ", 2560, 2048, 30976, 0), (f"{lay} siglip fc1 4304x1152x24576 ", 4304, 1152, 24576, 0),
             (f"{lay} siglip qkv 3456x1152x24576 ", 3456, 1152, 24576, 0), (f"{lay} ragged 1000x520x4104       ", 1000, 520, 4104, 0),
             (f"{lay} square 8192               ", 8192, 8192, 8192, 0)]  # fmt: skip
out = []
for name, M, N, K, act in cases:
    lay = os.environ.get("AB_LAYOUT", "NT")
    if lay == "NT":
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        lkw = dict(lda=K, ldb=K)
    elif lay == "TN":
        x = torch.randn(K, M, device=dev).to(BF16)
        w = (torch.randn(K, N, device=dev) * 0.05).to(BF16)
        lkw = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
    elif lay == "AT":  # A contraction-strided [K][M], B K-contiguous [N][K] (a weight gradient with its small operand transposed)
        x = torch.randn(K, M, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        lkw = dict(a_kc=False, b_kc=True, lda=M, ldb=K)
    else:  # NN
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(K, N, device=dev) * 0.05).to(BF16)
        lkw = dict(a_kc=True, b_kc=False, lda=K, ldb=N)
    ldc = (N + 63) // 64 * 64
    res = {c: torch.empty(M, ldc, dtype=BF16, device=dev) for c in CFGS}
    kw = dict(M=M, N=N, K=K, ldc=ldc, **lkw)
    if lay in ("TN", "AT", "NN") and K >= 8192:
        kw["split_k"] = ops.pick_split_k_wgrad(M, N, K)
    extra = {}
    if act:
        g = torch.randn(M, ldc, device=dev).to(BF16)
        u = torch.randn(M, ldc, device=dev).to(BF16)
        pre = torch.empty(M, ldc, dtype=BF16, device=dev)
        extra = dict(act=act, aux1=g, pre_out=pre, split_k=1)
        if act == 3:
            extra["aux2"] = u
        if act == 1:  # SigLIP fc1 forward: bias, pre-activation kept for the backward, tanh-GELU
            extra = dict(act=1, pre_out=pre, bias=torch.randn(N, device=dev).to(BF16), split_k=1)

    def run(c):
        set_cfg(c)
        ops.gemm(x, w, res[c], **kw, **extra)

    for c in CFGS:
        run(c)
    torch.cuda.synchronize()
    same = {c: bool(torch.equal(res[c][:, :N], res[CFGS[0]][:, :N])) for c in CFGS}
    iters = max(3, int(2e13 / (2.0 * M * N * K)))
    ts = {c: [] for c in CFGS}
    for _ in range(rounds):
        for c in CFGS:
            ts[c].append(timed(lambda: run(c), iters))
    row = {"case": name.strip(), "M": M, "N": N, "K": K, "act": act}
    line = f"{name} "
    for c in CFGS:
        ms = statistics.median(ts[c])
        tf = 2.0 * M * N * K / ms / 1e9
        row[f"cfg{c}"] = {"ms": ms, "tflops": tf, "min_ms": min(ts[c]), "equal_to_default": same[c]}
        line += f"| cfg{c}: {ms:7.3f} ms {tf:7.1f} TF/s {'==' if same[c] else '!='} "
    print(line, flush=True)
    out.append(row)
    del x, w, res
set_cfg(0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/gemm_sched_ab_{os.environ.get('AB_LAYOUT', 'NT')}.json", "w"), indent=1)
