"""Tile configuration A/B (kai0_gemm_desc.tile_cfg) on the training step's GEMM shapes with their fused epilogues: the automatic rule (0)
against 1 = 128x128 four waves two stages (2 blocks per CU), 6 = 128x128 EIGHT waves two stages (2 blocks per CU, 16 waves), 7 = 256x128x32
on four waves, three stages (2 blocks per CU) — round 6, VERDICT r5 #2 "two tiles in flight per CU".  `!` = bits differ from cfg 0.
usage: python tools/gemm_cfg_ab.py"""
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

def case(name, M, N, K, act=0, bias=False, residual=False, tn=False, nn=False, split=1, cfgs=(0, 1, 6, 7)):
    kw = {}
    if tn:   # weight gradient: C[M,N] = A[K,M]^T B[K,N]
        A = rnd(K, M); W = rnd(K, N)
        lay = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
    elif nn:  # dgrad: C[M,N] = A[M,K] B[K,N]
        A = rnd(M, K); W = rnd(K, N, scale=0.03)
        lay = dict(a_kc=True, b_kc=False, lda=K, ldb=N)
    else:
        A = rnd(M, K); W = rnd(N, K, scale=0.03)
        lay = dict(lda=K, ldb=K)
    if act == 5: kw = dict(act=5, aux1=rnd(M, N))
    if act == 3: kw = dict(act=3, aux1=rnd(M, N), aux2=rnd(M, N), pre_out=torch.empty(M, N, dtype=BF16, device=dev))
    if act == 2: kw = dict(act=2, aux1=rnd(M, N), pre_out=torch.empty(M, N, dtype=BF16, device=dev))
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
case("siglip dgrad fc2 (act 5)", 24576, 4304, 1152, act=5, nn=True)
case("siglip fc2 (bias+res)", 24576, 1152, 4304, bias=True, residual=True)
case("siglip dgrad fc1", 24576, 1152, 4304, nn=True)
case("siglip out_proj (bias+res)", 24576, 1152, 1152, bias=True, residual=True)
case("siglip qkv (bias)", 24576, 3456, 1152, bias=True)
case("siglip dgrad qkv", 24576, 1152, 3456, nn=True)
case("siglip dgrad out", 24576, 1152, 1152, nn=True)
case("gemma o / q 2048 (res)", 30976, 2048, 2048, residual=True)
case("gemma qkv", 30976, 2560, 2048)
case("gemma dgrad down + geglu bwd (act 3)", 30976, 16384, 2048, act=3, nn=True)
case("gemma up + geglu (act 2)", 30976, 16384, 2048, act=2)
case("gemma down (res)", 30976, 2048, 16384, residual=True)
case("gemma dgrad gate/up", 30976, 2048, 16384, nn=True)
for sp in (1, 2, 4):
    case(f"wgrad siglip fc1 split {sp}", 1152, 4304, 24576, tn=True, split=sp, cfgs=(0, 6, 7))
    case(f"wgrad siglip qkv split {sp}", 3456, 1152, 24576, tn=True, split=sp, cfgs=(0, 6, 7))
    case(f"wgrad siglip out split {sp}", 1152, 1152, 24576, tn=True, split=sp, cfgs=(0, 6, 7))
