"""Tile configuration A/B on the SigLIP-shaped GEMMs (short contraction, fused epilogues, ragged N = 1152 / 4304):
auto rule against the forced 128x128 (2 blocks per CU) and 256x256 configurations.  usage: python tools/gemm_cfg_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import _lib, ops
from kai0_amd.ops import gemm
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
lib = _lib.load()

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(BF16)

def case(name, M, N, K, act=0, bias=False, residual=False, tn=False, split=1, cfgs=(0, 1, 2, 4, 5)):
    kw = {}
    if tn:   # weight gradient: C[M,N] = A[K,M]^T B[K,N]
        A = rnd(K, M); W = rnd(K, N)
        lay = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
    else:
        A = rnd(M, K); W = rnd(N, K, scale=0.03)
        lay = dict(lda=K, ldb=K)
    if act == 5: kw = dict(act=5, aux1=rnd(M, N))
    if act == 1: kw = dict(act=1, pre_out=torch.empty(M, N, dtype=BF16, device=dev))
    if bias: kw["bias"] = rnd(N)
    if residual: kw.update(residual=rnd(M, N), ldr=N)
    out = torch.zeros(M, N, dtype=BF16, device=dev)
    line = f"{name:30s} {M}x{N}x{K}"
    ref = None
    for cfg in cfgs:
        ops.GEMM_TUNING["tile_cfg"] = cfg
        fn = lambda: gemm(A, W, out, M=M, N=N, K=K, ldc=N, split_k=split, **lay, **kw)
        ms = timeit(fn)
        o = out.clone()
        if ref is None: ref = o
        line += f"  cfg{cfg} {2.0 * M * N * K / ms / 1e9:6.0f}{'' if torch.equal(o, ref) else '!'}"
    ops.GEMM_TUNING["tile_cfg"] = 0
    print(line, flush=True)

case("siglip fc1 (act 1)", 24576, 4304, 1152, act=1, bias=True)
case("siglip dgrad fc2 (act 5)", 24576, 4304, 1152, act=5)
case("siglip fc2 (bias+res)", 24576, 1152, 4304, bias=True, residual=True)
case("siglip dgrad fc1", 24576, 1152, 4304)
case("siglip out_proj (bias+res)", 24576, 1152, 1152, bias=True, residual=True)
case("siglip qkv (bias)", 24576, 3456, 1152, bias=True)
case("siglip dgrad qkv", 24576, 1152, 3456)
case("gemma o/q 2048", 30976, 2048, 2048)
for sp in (1, 2, 3, 4, 6):
    case(f"wgrad fc1 split {sp}", 1152, 4304, 24576, tn=True, split=sp, cfgs=(0, 1, 2, 4))
for sp in (1, 2, 3, 4, 6):
    case(f"wgrad fc2 split {sp}", 4304, 1152, 24576, tn=True, split=sp, cfgs=(0, 1, 2, 4))
for sp in (1, 2, 3, 4, 6):
    case(f"wgrad qkv split {sp}", 3456, 1152, 24576, tn=True, split=sp, cfgs=(0, 1, 2, 4))
for sp in (1, 2, 4, 6, 8):
    case(f"wgrad out split {sp}", 1152, 1152, 24576, tn=True, split=sp, cfgs=(0, 1, 2, 4))
