"""Is the bf16 GEMM plateau a power / clock cap?  Polls rocm-smi (shader clock, socket power) from a thread while this process keeps one
kind of launch running for a few seconds: idle, the persistent NT GEMM on a long contraction (92 % of its time inside the K loop), the
fused GeGLU-backward GEMM (46 % of its time in VALU epilogues), the vendor library on the first shape, and an HBM-bound copy.
usage: python tools/probes/clock_under_gemm.py"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import _lib  # noqa: E402
from kai0_amd.ops import gemm  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
lib = _lib.load()
samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), "")
            mclk = next((v for k, v in card.items() if "mclk" in k.lower()), "")
            pw = next((v for k, v in card.items() if "power" in k.lower() and "cap" not in k.lower()), "")
            samples.append((time.time(), sclk, mclk, pw))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), f"error {e}", "", ""))
        time.sleep(0.05)


def mhz(s):
    m = re.search(r"(\d+)\s*Mhz", str(s), re.I)
    return float(m.group(1)) if m else float("nan")


def watts(s):
    try:
        return float(s)
    except Exception:  # noqa: BLE001
        return float("nan")


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).to(BF16)


def phase(name, fn, seconds=4.0, flops=0.0):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    t1 = time.time()
    mine = [s for s in samples if t0 + 1.0 <= s[0] <= t1]  # skip the first second (ramp)
    clk = [mhz(s[1]) for s in mine if mhz(s[1]) == mhz(s[1])]
    pw = [watts(s[3]) for s in mine if watts(s[3]) == watts(s[3])]
    rate = flops * n / (t1 - t0) / 1e12 if flops else 0.0
    print(f"{name:46s} {n:6d} launches  {rate:7.1f} TFLOP/s  sclk MHz min {min(clk, default=float('nan')):6.0f} mean {sum(clk) / max(len(clk), 1):6.0f} max {max(clk, default=float('nan')):6.0f}"
          f"   power W mean {sum(pw) / max(len(pw), 1):6.0f} max {max(pw, default=float('nan')):6.0f}   ({len(mine)} samples)", flush=True)


th = threading.Thread(target=poll, daemon=True)
th.start()
time.sleep(1.0)
print("first sample:", samples[0] if samples else None, flush=True)
phase("idle", lambda: None, 2.0)
M = 30976
A, W = rnd(M, 16384), rnd(2048, 16384, sc=0.03)
out = torch.empty(M, 2048, dtype=BF16, device=dev)
phase("persistent NT 30976x2048x16384 (K loop 92 %)", lambda: gemm(A, W, out, M=M, N=2048, K=16384, lda=16384, ldb=16384, ldc=2048), flops=2.0 * M * 2048 * 16384)
phase("vendor library, same shape", lambda: torch.matmul(A, W.t(), out=out), flops=2.0 * M * 2048 * 16384)
A2, W2 = rnd(M, 2048), rnd(16384, 2048, sc=0.03)
o2, pre = torch.empty(M, 16384, dtype=BF16, device=dev), torch.empty(M, 16384, dtype=BF16, device=dev)
g, u = rnd(M, 16384), rnd(M, 16384)
phase("GeGLU-backward GEMM 30976x16384x2048 (act 3)", lambda: gemm(A2, W2, o2, M=M, N=16384, K=2048, lda=2048, ldb=2048, ldc=16384, act=3, aux1=g, aux2=u, pre_out=pre),
      flops=2.0 * M * 16384 * 2048)
phase("plain NT 30976x16384x2048", lambda: gemm(A2, W2, o2, M=M, N=16384, K=2048, lda=2048, ldb=2048, ldc=16384), flops=2.0 * M * 16384 * 2048)
phase("vendor library, same shape", lambda: torch.matmul(A2, W2.t(), out=o2), flops=2.0 * M * 16384 * 2048)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
big2 = torch.empty_like(big)
phase("HBM copy 1 GiB", lambda: big2.copy_(big))
stop = True
