"""Persistent NT kernel (dynamic tile queue) against one block per tile on the pi0.5 training shapes: bit-identity and TFLOP/s.
usage: python tools/gemm_persist_ab.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import _lib, ops
from kai0_amd.ops import gemm
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
lib = _lib.load()

def timeit(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(BF16)

cases = []
def case(name, M, N, K, act=0, bias=False, residual=False):
    A = rnd(M, K); W = rnd(N, K, scale=0.03)
    kw = {}
    outs = {}
    if act == 6:
        W2 = rnd(N, K, scale=0.03); kw = dict(act=6, B2=W2)
    if act in (2, 3, 5):
        kw = dict(act=act, aux1=rnd(M, N))
        if act == 3: kw["aux2"] = rnd(M, N)
    if act == 1: kw = dict(act=1)
    if bias: kw["bias"] = rnd(N)
    if residual: kw.update(residual=rnd(M, N), ldr=N)
    res = {}
    for mode in (0, 2):
        ops.GEMM_TUNING["persist"] = 2 if mode == 2 else 1
        out = torch.empty(M, N, dtype=BF16, device=dev)
        pre = torch.empty(M, N, dtype=BF16, device=dev) if act in (2, 3, 6, 1) else None
        pre2 = torch.empty(M, N, dtype=BF16, device=dev) if act == 6 else None
        k2 = dict(kw)
        if pre is not None and act != 5: k2["pre_out"] = pre
        if pre2 is not None: k2["pre_out2"] = pre2
        fn = lambda: gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **k2)
        fn(); torch.cuda.synchronize()
        ms = timeit(fn)
        flops = 2.0 * M * N * K * (2 if act == 6 else 1)
        res[mode] = (ms, flops / ms / 1e9, out.clone(), None if pre is None else pre.clone(), None if pre2 is None else pre2.clone())
    ops.GEMM_TUNING["persist"] = 0
    same = torch.equal(res[0][2], res[2][2]) and all((a is None) or torch.equal(a, b) for a, b in zip(res[0][3:], res[2][3:]))
    print(f"{name:34s} {M}x{N}x{K} act {act}: plain {res[0][1]:7.1f} TF/s ({res[0][0]:.3f} ms)  persistent {res[2][1]:7.1f} TF/s ({res[2][0]:.3f} ms)  "
          f"{'bit-identical' if same else 'DIFFERENT'}  {100 * (res[2][1] / res[0][1] - 1):+.1f} %", flush=True)
    cases.append(dict(case=name, M=M, N=N, K=K, act=act, tflops_plain=res[0][1], tflops_persistent=res[2][1], identical=same))

case("gate|up pair (act 6)", 30976, 16384, 2048, act=6)
case("dh + GeGLU bwd (act 3)", 30976, 16384, 2048, act=3)
case("up + GeGLU (act 2)", 30976, 16384, 2048, act=2)
case("gate plain", 30976, 16384, 2048)
case("down NT K=16384", 30976, 2048, 16384)
case("o/q NT 2048x2048", 30976, 2048, 2048)
case("qkv NT", 30976, 2560, 2048)
case("siglip fc1 (bias+gelu, act 1)", 24576, 4304, 1152, act=1, bias=True)
case("siglip fc2 (bias+res)", 24576, 1152, 4304, bias=True, residual=True)
case("siglip dgrad fc2 (act 5)", 24576, 4304, 1152, act=5)
case("siglip out_proj (bias+res)", 24576, 1152, 1152, bias=True, residual=True)
case("siglip qkv (bias)", 24576, 3456, 1152, bias=True)
case("ragged M N", 3000, 4104, 1152, bias=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(cases, open("gpurun_out/gemm_persist_ab_r4.json", "w"), indent=1)
