"""Phase timeline of the in-block weight-streaming kernels of the denoise loop (kai0_gemm_skinny_bf16, split_k = -1, trace buffer):
shader-clock stamps of wave 0 of every block at  0 entry | 1 all loads issued | 2 A tile in LDS | 3 row statistics exchanged |
4 A fragments ready | 5 MFMAs issued | 6 wave partials in LDS | 7 epilogue stores issued.  18 different weight sets are cycled so that the
weights arrive cold, as in the chunk.  usage: python tools/probes/sk2_phases.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
M, De, F, NQ, HD, L = 50, 1024, 4096, 2048, 256, 18
S_ld, P = 1024, 968
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s, dt=BF16, sc=0.05):
    return (torch.randn(*s, generator=g, device=dev) * sc).to(dt)


w_qkv = [ops.pack_skinny_weight(rnd(NQ + 2 * HD, De)) for _ in range(L)]
w_o = [ops.pack_skinny_weight(rnd(De, NQ)) for _ in range(L)]
w_gu = [ops.pack_skinny_weight(rnd(2 * F, De)) for _ in range(L)]
w_d = [ops.pack_skinny_weight(rnd(De, F)) for _ in range(L)]
xs, x1, h = rnd(M, De, sc=1.0), rnd(M, De, sc=1.0), rnd(M, F, sc=1.0)
att = rnd(1, S_ld, NQ, sc=1.0)
q_buf, k_c, vt = torch.zeros(1, S_ld, NQ, dtype=BF16, device=dev), torch.zeros(1, S_ld, HD, dtype=BF16, device=dev), torch.zeros(1, HD, S_ld, dtype=BF16, device=dev)
mod = rnd(1, 3 * De, dt=F32, sc=0.3)
gate = rnd(1, De, sc=1.0)
cos, sin = torch.rand(M, HD // 2, device=dev), torch.rand(M, HD // 2, device=dev)
out_d, out_h = torch.empty(M, De, dtype=BF16, device=dev), torch.empty(M, F, dtype=BF16, device=dev)
trace = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)


def call(kind, l, tr):
    if kind == "qkv":
        ops.skinny_gemm(xs, w_qkv[l], M=M, N=NQ + 2 * HD, K=De, lda=De, ldw=De, mode=1, pair_stride=HD // 2, split_k=-1,
                        segs=[(q_buf, NQ, 0, NQ, 1), (k_c, HD, NQ, NQ + HD, 1), (vt, S_ld, NQ + HD, NQ + 2 * HD, 2)], c_map=(M, S_ld, P),
                        rope_cos=cos, rope_sin=sin, rope_half=HD // 2, mod=mod, mod_ld=3 * De, mod_rpb=M, eps=1e-6, w_packed=True, workspace=tr)
    elif kind == "o_proj":
        ops.skinny_gemm(att, w_o[l], M=M, N=De, K=NQ, lda=NQ, ldw=NQ, split_k=-1, a_map=(M, S_ld, P), segs=[(out_d, De, 0, De, 0)],
                        gate=gate, gate_rpb=M, gate_ld=De, residual=xs, ldr=De, w_packed=True, workspace=tr)
    elif kind == "gate_up":
        ops.skinny_gemm(x1, w_gu[l], M=M, N=2 * F, K=De, lda=De, ldw=De, mode=2, pair_stride=F, split_k=-1, segs=[(out_h, F, 0, F, 0)],
                        mod=mod, mod_ld=3 * De, mod_rpb=M, eps=1e-6, w_packed=True, workspace=tr)
    else:
        ops.skinny_gemm(h, w_d[l], M=M, N=De, K=F, lda=F, ldw=F, split_k=-1, segs=[(out_d, De, 0, De, 0)], gate=gate, gate_rpb=M,
                        gate_ld=De, residual=x1, ldr=De, w_packed=True, workspace=tr)


nblk = {"qkv": 80 * 4, "o_proj": 64 * 4, "gate_up": 256, "down": 64 * 4}
for kind in ("qkv", "o_proj", "gate_up", "down"):
    for l in range(L):
        call(kind, l, None)
    torch.cuda.synchronize()
    acc = torch.zeros(8, dtype=torch.float64)
    tot = 0.0
    n = 0
    for rep in range(2):
        for l in range(L):
            trace.zero_()
            call(kind, l, trace)
            torch.cuda.synchronize()
            t = trace[: nblk[kind] * 8].view(-1, 8).cpu().double()
            t0 = t[:, 0].min()
            acc += (t - t[:, :1]).mean(0)
            tot += float(t[:, 7].max() - t0)
            n += 1
    rel = (acc / n).tolist()
    # timed without the trace, for the scale (events around 18 cold launches)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for l in range(L):
        call(kind, l, None)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / L * 1e3
    print(f"{kind:8s} eager {us:6.2f} us/launch | block span (first entry -> last exit) {tot / n:8.0f} clk | mean clk since block entry: "
          + "  ".join(f"{i}:{v:7.0f}" for i, v in enumerate(rel)))
