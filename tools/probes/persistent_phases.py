"""Where a tile's time goes in the persistent NT GEMM kernel (library built with KAI0_HIPCC_FLAGS=-DKAI0_PS_TRACE): per-block sums of
100 MHz ticks in  tile start | K loop | hand-over issue + epilogue passes | ticket, drain, barriers,  averaged per tile over the blocks of the
last launch, on the MLP shapes of the pi0.5 training step (B = 32: 30976 rows).  usage: python tools/probes/persistent_phases.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import _lib  # noqa: E402
from kai0_amd import ops  # noqa: E402
from kai0_amd.ops import gemm  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
lib = _lib.load()
if not hasattr(lib, "kai0_debug_ps_trace"):
    raise SystemExit("build the library with KAI0_HIPCC_FLAGS=-DKAI0_PS_TRACE first")
ops.GEMM_TUNING["persist"] = 2  # every eligible NT launch on the persistent kernel


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).to(BF16)


def run(name, M, N, K, **kw):
    A, W = rnd(M, K), rnd(N, K, sc=0.03)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    k2 = dict(kw)
    act = kw.get("act", 0)
    if act == 6:
        k2.update(B2=rnd(N, K, sc=0.03), pre_out=torch.empty(M, N, dtype=BF16, device=dev), pre_out2=torch.empty(M, N, dtype=BF16, device=dev))
    if act in (2, 3, 5):
        k2["aux1"] = rnd(M, N)
    if act == 3:
        k2.update(aux2=rnd(M, N), pre_out=torch.empty(M, N, dtype=BF16, device=dev))
    for _ in range(3):
        gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **k2)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (256 * 8))()
    rc = lib.kai0_debug_ps_trace(buf)
    assert rc == 0, rc
    t = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)
    tiles = t[:, 4].clamp(min=1)
    per = (t[:, :4] / tiles[:, None]).mean(0) * 0.01  # us per tile
    tot = float(per.sum())
    print(f"{name:34s} {M}x{N}x{K}: tiles/block {t[:, 4].mean():5.1f} | per tile: start {per[0]:5.2f}  K loop {per[1]:7.2f} us ({per[1] / ((K + 63) // 64):.3f} per K-tile, "
          f"{100 * per[1] / tot:4.1f} %)  epilogue {per[2]:6.2f} ({100 * per[2] / tot:4.1f} %)  ticket+drain {per[3]:5.2f} ({100 * per[3] / tot:4.1f} %)", flush=True)


M = 30976
_a, _w, _o = rnd(M, 2048), rnd(16384, 2048, sc=0.03), torch.empty(M, 16384, dtype=BF16, device=dev)
for _ in range(30):  # clock / power state of a running step before the first measured case (the first case read 10 % slow without it)
    gemm(_a, _w, _o, M=M, N=16384, K=2048, lda=2048, ldb=2048, ldc=16384)
torch.cuda.synchronize()
del _a, _w, _o
run("gate|up pair fwd (act 6)", M, 16384, 2048, act=6)
run("dh + GeGLU bwd (act 3)", M, 16384, 2048, act=3)
run("up + GeGLU fwd (act 2)", M, 16384, 2048, act=2)
run("plain N=16384", M, 16384, 2048)
run("down fwd K=16384 (gate+residual)", M, 2048, 16384, gate=rnd(32, 2048), gate_rpb=968, gate_ld=2048, residual=rnd(M, 2048), ldr=2048)
run("dgrad K=16384 plain", M, 2048, 16384)
run("dgrad K=16384 accumulate", M, 2048, 16384, accumulate=True)
run("siglip fc1 (bias + gelu, act 1)", 24576, 4304, 1152, act=1, bias=rnd(4304), pre_out=torch.empty(24576, 4304, dtype=BF16, device=dev))
run("siglip dgrad fc2 (act 5)", 24576, 4304, 1152, act=5)
run("siglip qkv (bias)", 24576, 3456, 1152, bias=rnd(3456))
