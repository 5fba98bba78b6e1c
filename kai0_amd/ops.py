"""Host-side operator layer over the C-ABI (include/kai0hip.h).

Two levels:
  * `raw.*`-style functions (`gemm`, `rmsnorm_fwd`, ...) — 1:1 over the C entry points: they take torch CUDA
    tensors, pass `data_ptr()` + the current HIP stream, allocate nothing but their outputs.
  * `torch.autograd.Function` shims (`LinearFn`, `RMSNormFn`, ...) that pair each forward kernel with its
    backward kernels so the model stays an ordinary nn.Module with `.grad` semantics — which is what
    kai0's model_arithmetic (arithmetic_torch.py:197-218) and train_pytorch.py:547-567 rely on.

PyTorch is used here for device memory, streams and autograd bookkeeping only; all arithmetic on the path
is done by libkai0hip.so.  There is deliberately no CPU / eager fallback.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os
import math

import torch

from . import _lib
from ._lib import AttnBwdDesc, AttnDesc, GemmDesc

BF16 = torch.bfloat16
F32 = torch.float32

NORM_PARTIAL_BLOCKS = 512  # blocks (x4 waves) producing dw partials in norm backward
COLSUM_BLOCKS = 256


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise _lib.Kai0HipError(f"{name}: expected a CUDA (HIP) tensor; the product path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


# ------------------------------------------------------------------------------------------------ GEMM
# The A/B and test hooks of kai0_gemm_desc (kai0hip.h: tile_cfg / persist / general_epilogue), all 0 in production.  They travel with
# every call's descriptor — the library itself has no process-wide switch; tests and tools set them through `gemm_tuning(...)`.
# (KAI0_GEMM_PERSIST=0 / 1 / 2 — never / the library's rule / every eligible NT launch — is read HERE, on the host side of the boundary.)
GEMM_TUNING = {"tile_cfg": 0, "persist": {"0": 1, "2": 2}.get(os.environ.get("KAI0_GEMM_PERSIST", "1"), 0), "general_epilogue": 0,
               "small_w8": {"0": 1, "2": 2}.get(os.environ.get("KAI0_GEMM_W8", "1"), 0)}  # env 0 / 1 / 2 = never / the rule / always, like KAI0_GEMM_PERSIST


@contextlib.contextmanager
def gemm_tuning(**kw):
    """with ops.gemm_tuning(persist=2): ...  — descriptor hooks for the GEMM launches made inside the block."""
    unknown = set(kw) - set(GEMM_TUNING)
    if unknown:
        raise KeyError(f"gemm_tuning: unknown hook(s) {sorted(unknown)}")
    old = dict(GEMM_TUNING)
    GEMM_TUNING.update(kw)
    try:
        yield
    finally:
        GEMM_TUNING.update(old)


def gemm(
    A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int, a_kc: bool = True,
    b_kc: bool = True, lda: int, ldb: int, ldc: int, batch: int = 1, batch_inner: int = 1,
    sA=(0, 0), sB=(0, 0), sC=(0, 0), a_map=None, b_map=None, c_map=None, bias: torch.Tensor | None = None,
    scale: float = 1.0, act: int = 0, pre_out: torch.Tensor | None = None, gate: torch.Tensor | None = None,
    gate_rpb: int = 0, gate_ld: int = 0, residual: torch.Tensor | None = None, ldr: int = 0, sR=(0, 0),
    accumulate: bool = False, a_off_elems: int = 0, b_off_elems: int = 0, c_off_elems: int = 0, split_k: int = 1,
    aux1: torch.Tensor | None = None, aux2: torch.Tensor | None = None, segs=None, rowvec=None, rv=(0, 0, 1),
    B2: torch.Tensor | None = None, pre_out2: torch.Tensor | None = None, norm=None, nt_out: bool = False, rope=None,
) -> torch.Tensor:  # fmt: skip
    """kai0_gemm_bf16. `*_map` = (rows_per_batch, batch_stride_rows, row_offset). `*_off_elems` shift the base
    pointer (for column slices such as a head inside a fused projection)."""
    for t in (A, B, out):
        if not t.is_cuda:
            raise _lib.Kai0HipError("gemm: expected CUDA (HIP) tensors; the product path has no CPU fallback")
    if A.dtype != BF16 or B.dtype != BF16:
        raise TypeError(f"gemm: bf16 operands expected, got {A.dtype} / {B.dtype}")
    d = GemmDesc()
    d.A = A.data_ptr() + 2 * a_off_elems
    d.B = B.data_ptr() + 2 * b_off_elems
    out_f32 = out.dtype == F32
    d.C = out.data_ptr() + (4 if out_f32 else 2) * c_off_elems
    d.M, d.N, d.K = M, N, K
    d.a_kc, d.b_kc = int(a_kc), int(b_kc)
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.batch, d.batch_inner = batch, batch_inner
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    if a_map:
        d.a_rpb, d.a_bs, d.a_off = a_map
    if b_map:
        d.b_rpb, d.b_bs, d.b_off = b_map
    if c_map:
        d.c_rpb, d.c_bs, d.c_off = c_map
    if bias is not None:
        d.bias = bias.data_ptr()
        d.bias_f32 = int(bias.dtype == F32)
    d.scale = scale
    d.act = act
    d.out_f32 = int(out_f32)
    d.pre_out = _p(pre_out)
    d.aux1, d.aux2 = _p(aux1), _p(aux2)
    if rope is not None:  # act 7: (cos bf16 [M, half], sin bf16 [M, half], half, n_end) — B's rows permuted (rope_permutation)
        rc, rs, half, n_end = rope
        if rc.dtype != BF16 or rs.dtype != BF16 or not rc.is_contiguous() or not rs.is_contiguous() or rc.shape != (M, half) or rs.shape != (M, half):
            raise TypeError("gemm: rope tables must be contiguous bf16 [M, half]")
        d.rope_cos, d.rope_sin, d.rope_half, d.rope_n_end = rc.data_ptr(), rs.data_ptr(), half, n_end
    if B2 is not None:  # act 6: gate (B) | up (B2) weights of the GeGLU pair GEMM
        if B2.dtype != BF16 or not B2.is_cuda:
            raise TypeError("gemm: B2 must be a bf16 HIP tensor")
        d.B2 = B2.data_ptr()
    d.pre_out2 = _p(pre_out2)
    if gate is not None:
        d.gate = gate.data_ptr()
        d.gate_rpb = gate_rpb
        d.gate_ld = gate_ld
    if residual is not None:
        d.residual = residual.data_ptr()
        d.ldr = ldr
        d.sR1, d.sR2 = sR
    d.accumulate = int(accumulate)
    d.c_nontemporal = int(nt_out)  # weight gradients: stored past the caches (kai0hip.h)
    if rowvec is not None:  # act 4: D index = z1*rv[0] + z2*rv[1] + row*rv[2]
        if rowvec.dtype != F32:
            raise TypeError("gemm: rowvec must be f32")
        d.rowvec = rowvec.data_ptr()
        d.rv_s1, d.rv_s2, d.rv_ld = rv
    if segs:  # [(dst, ld, n_begin)]: output columns routed to several destinations
        d.nseg = len(segs)
        for i, (dst, ld, nb) in enumerate(segs):
            d.seg[i].dst, d.seg[i].ld, d.seg[i].n_begin = dst.data_ptr(), ld, nb
    if split_k > 1:
        ws = _workspace(batch * split_k * M * N * 4, A.device)
        d.split_k, d.workspace, d.workspace_bytes = split_k, ws.data_ptr(), ws.numel()
    if norm is not None:  # (kind, norm_out, weight, bias | None, eps): the consumer's norm, fused into the split-K reduction
        kind, nout, nw, nb, neps = norm
        _chk(nout, BF16, "gemm: norm_out")
        _chk(nw, F32 if kind == 1 else BF16, "gemm: norm weight")
        d.norm_kind, d.norm_out, d.norm_w, d.norm_eps = int(kind), nout.data_ptr(), nw.data_ptr(), float(neps)
        if nb is not None:
            _chk(nb, BF16, "gemm: norm bias")
            d.norm_b = nb.data_ptr()
    d.tile_cfg, d.persist, d.general_epilogue, d.small_w8 = (GEMM_TUNING["tile_cfg"], GEMM_TUNING["persist"], GEMM_TUNING["general_epilogue"],
                                                               GEMM_TUNING["small_w8"])
    _lib.call("kai0_gemm_bf16", C.byref(d), _stream())
    return out


def skinny_split_k(N: int, K: int) -> int:
    """Split of the contraction for a mode-0 kai0_gemm_skinny_bf16 whose consumer is kai0_adarms_combine: enough blocks
    to occupy the chip (N/32 tiles x split >= ~190), K / split in {1024, 512}."""
    if K % 512:
        raise _lib.Kai0HipError(f"skinny gemm: K={K} must be a multiple of 512")
    tiles = N // 32
    if K % 1024 == 0 and tiles * (K // 1024) >= 190:
        return K // 1024
    return K // 512


def skinny_workspace(M: int, N: int, split_k: int, device) -> torch.Tensor:
    """f32 [split_k][M][N] partial-product buffer of a split-K skinny GEMM."""
    return torch.empty((max(split_k, 1), M, N), dtype=F32, device=device)


def skinny_gemm(A, W, *, M: int, N: int, K: int, lda: int, ldw: int, mode: int = 0, pair_stride: int = 16, segs=None,
                split_k: int = 1, workspace=None, a_map=None, c_map=None, gate=None, gate_rpb: int = 0, gate_ld: int = 0,
                residual=None, ldr: int = 0, rope_cos=None, rope_sin=None, rope_half: int = 0, mod=None, mod_ld: int = 0,
                mod_rpb: int = 0, eps: float = 1e-6, w_packed: bool = False, rowsq_in=None, rowsq_parts: int = 0, cvec=None,
                rowsq_out=None):  # fmt: skip
    """kai0_gemm_skinny_bf16.  segs = [(dst, ld, n_begin, n_end, rope)].  split_k = -1: the whole contraction inside one block
    (no partial products); with `mod` (f32 view [b][>= 2K], row stride mod_ld) the A operand is adaRMS-normalised on the fly;
    `w_packed`: W is the output of `pack_skinny_weight` (fragment-major 1-KiB blocks).  Folded adaRMS (kai0hip.h): W already
    carries (1 + scale), `cvec` [N] f32 the shift term, `rowsq_in` [parts, >= M] f32 the producer's partial sums of squares of the
    rows of A; `rowsq_out` [N / 16, >= M] f32 (mode 0) receives this launch's own partials."""
    for t in (A, W):
        if not t.is_cuda or t.dtype != BF16:
            raise _lib.Kai0HipError("skinny_gemm: expected bf16 CUDA (HIP) tensors; the product path has no CPU fallback")
    d = _lib.SkinnyDesc()
    d.A, d.W, d.lda, d.ldw = A.data_ptr(), W.data_ptr(), lda, ldw
    d.M, d.N, d.K = M, N, K
    d.pair_stride, d.mode, d.split_k = pair_stride, mode, split_k
    d.w_packed = int(w_packed)
    if a_map:
        d.a_rpb, d.a_bs, d.a_off = a_map
    if c_map:
        d.c_rpb, d.c_bs, d.c_off = c_map
    d.nseg = len(segs or ())
    for i, (dst, ld, nb, ne, rope) in enumerate(segs or ()):
        if not dst.is_cuda or dst.dtype != BF16:
            raise _lib.Kai0HipError("skinny_gemm: destinations must be bf16 CUDA (HIP) tensors")
        d.seg[i].dst, d.seg[i].ld, d.seg[i].n_begin, d.seg[i].n_end, d.seg[i].rope = dst.data_ptr(), ld, nb, ne, int(rope)
    if gate is not None:
        d.gate, d.gate_rpb, d.gate_ld = gate.data_ptr(), gate_rpb, gate_ld
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), ldr
    if rope_cos is not None:
        d.rope_cos, d.rope_sin, d.rope_half = rope_cos.data_ptr(), rope_sin.data_ptr(), rope_half
    if split_k > 1:
        if workspace is None or workspace.dtype != F32:
            raise _lib.Kai0HipError("skinny_gemm: split_k > 1 writes f32 partial products into `workspace` (skinny_workspace)")
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * 4
    elif split_k == -1 and workspace is not None:  # diagnostics: phase trace of the in-block kernels (tools/probes/sk2_phases.py)
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if mod is not None:
        if mod.dtype != F32 or not mod.is_cuda or mod.stride(-1) != 1:
            raise _lib.Kai0HipError("skinny_gemm: mod must be an f32 CUDA (HIP) tensor with unit inner stride")
        d.mod, d.mod_ld, d.mod_rpb, d.eps = mod.data_ptr(), mod_ld, mod_rpb, eps
    if rowsq_in is not None:
        _chk(rowsq_in, F32, "rowsq_in")
        _chk(cvec, F32, "cvec")
        d.rowsq_in, d.cvec, d.rowsq_parts, d.rowsq_ld, d.eps = rowsq_in.data_ptr(), cvec.data_ptr(), rowsq_parts, rowsq_in.stride(0), eps
    if rowsq_out is not None:
        _chk(rowsq_out, F32, "rowsq_out")
        d.rowsq_out, d.rowsq_out_ld = rowsq_out.data_ptr(), rowsq_out.stride(0)
    _lib.call("kai0_gemm_skinny_bf16", C.byref(d), _stream())


def pack_skinny_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> the fragment-major layout kai0_gemm_skinny_bf16 streams with `w_packed` (kai0hip.h): for every 16-row
    tile t and 32-wide contraction step s one contiguous 1-KiB block holding W[16 t + i][32 s + 8 g + e] at (i + 16 g) * 8 + e.
    A copy made once per inference engine: the weights stay the checkpoint-visible parameters."""
    N, K = w.shape
    if N % 16 or K % 32:
        raise ValueError(f"pack_skinny_weight: N % 16 == 0 and K % 32 == 0 required, got {tuple(w.shape)}")
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def adarms_combine(partials, gate_prev, residual, mod, rows_per_batch: int, eps: float = 1e-6):
    """kai0_adarms_combine: partials f32 [S][rows][D] -> (x, y, gate) with x = gated residual of the summed product,
    (y, gate) = adaRMS(x, mod)."""
    S, rows, D = partials.shape
    Bn = rows // rows_per_batch
    dev = partials.device
    x = torch.empty((rows, D), dtype=BF16, device=dev)
    y = torch.empty((rows, D), dtype=BF16, device=dev)
    gate = torch.empty((Bn, D), dtype=BF16, device=dev)
    _lib.call("kai0_adarms_combine", partials.data_ptr(), S, rows * D, _p(gate_prev), _p(residual), x.data_ptr(),
              mod.data_ptr(), y.data_ptr(), gate.data_ptr(), rows, rows_per_batch, D, eps, _stream())  # fmt: skip
    return x, y, gate


def attn_decode(Q, K, Vt, O, qcode, kcode, *, batch, rows, H, HD, Sk, q0, q_bs, k_bs, k_ld, k_rows, vt_bs, vt_ld, scale):
    """kai0_attn_decode (masked MQA of the denoise loop in two launches; Vt is the transposed value cache)."""
    for t in (Q, K, Vt, O):
        if not t.is_cuda or t.dtype != BF16:
            raise _lib.Kai0HipError("attn_decode: expected bf16 CUDA (HIP) tensors; the product path has no CPU fallback")
    ws = _workspace(int(_lib.load().kai0_attn_decode_workspace_bytes(batch, rows)), Q.device)
    _lib.call("kai0_attn_decode", Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), O.data_ptr(), _p(qcode), _p(kcode), batch, rows,
              H, HD, Sk, q0, q_bs, k_bs, k_ld, k_rows, vt_bs, vt_ld, qcode.stride(0) if qcode is not None else 0,
              kcode.stride(0) if kcode is not None else 0, scale, ws.data_ptr(), ws.numel(), _stream())  # fmt: skip


def transpose_strided(src, dst, *, R, C, src_ld, dst_ld, batch=1, src_bs=0, dst_bs=0):
    """dst[z][c][r] = src[z][r][c] (kai0_transpose_strided_bf16)."""
    _lib.call("kai0_transpose_strided_bf16", src.data_ptr(), dst.data_ptr(), R, C, src_ld, dst_ld, batch, src_bs, dst_bs,
              _stream())  # fmt: skip


def rowdot(a, b, D: int):
    """out[r] = sum_d a[r, d] * b[r, d] in f32 over rows of D contiguous bf16 elements (a, b contiguous, same shape)."""
    if a.dtype != BF16 or b.dtype != BF16 or not a.is_contiguous() or not b.is_contiguous() or a.numel() != b.numel():
        raise _lib.Kai0HipError("rowdot: two contiguous bf16 tensors of the same size expected")
    rows = a.numel() // D
    out = torch.empty((rows,), dtype=F32, device=a.device)
    _lib.call("kai0_rowdot_bf16", a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, D, _stream())
    return out


def rope_permutation(n_total: int, n_rope_end: int, half: int = 128) -> torch.Tensor:
    """Row order of a stacked q | k | v weight for the RoPE epilogue of kai0_gemm_bf16 (act 7, kai0hip.h): permuted row pc holds real row
    perm[pc]; inside [0, n_rope_end) every 128-row tile = 64 first-half rows of a head followed by their 64 partners."""
    if half != 128 or n_rope_end % 256 != 0 or n_rope_end > n_total:
        raise ValueError("rope_permutation: head_dim 256 (half 128), rotated range a whole number of heads")
    pc = torch.arange(n_total)
    real = (pc // 256) * 256 + ((pc % 128) // 64) * 128 + 64 * ((pc % 256) // 128) + pc % 64
    return torch.where(pc < n_rope_end, real, pc)


def rope_table(pos, inv_freq, bf16: bool = False):
    """-> (cos, sin) [rows, HD/2], bf16-rounded values, rows = pos.numel() (pos int32); stored as f32, or as bf16 with `bf16`."""
    pos = pos.contiguous()
    if pos.dtype != torch.int32:
        raise TypeError("rope_table: pos must be int32")
    rows, half = pos.numel(), inv_freq.numel()
    cos = torch.empty((rows, half), dtype=BF16 if bf16 else F32, device=pos.device)
    sin = torch.empty((rows, half), dtype=BF16 if bf16 else F32, device=pos.device)
    _lib.call("kai0_rope_table", pos.data_ptr(), inv_freq.data_ptr(), cos.data_ptr(), sin.data_ptr(), rows, half, int(bf16), _stream())
    return cos, sin


def prefix_codes(img_masks, lang_mask, n_img: int, Hs: int):
    """kai0_prefix_codes: (qcode, kcode, pos) int32 [B, ncam n_img + T + Hs] of one pi0.5 request from its camera masks (bool [B] each) and
    prompt mask (bool [B, T]) in one launch — bit for bit `model.build_mask_codes` on embed_prefix's / embed_suffix's pad and att masks."""
    B, T = lang_mask.shape
    ms = [m.contiguous() for m in img_masks]
    lm = lang_mask.contiguous()
    for m in (*ms, lm):
        if m.dtype != torch.bool or not m.is_cuda:
            raise TypeError("prefix_codes: bool CUDA (HIP) masks expected")
    if any(m.shape != (B,) for m in ms):
        raise ValueError("prefix_codes: camera masks must be [B]")
    S = len(ms) * n_img + T + Hs
    q, k, p = (torch.empty((B, S), dtype=torch.int32, device=lm.device) for _ in range(3))
    ptrs = (C.c_void_p * len(ms))(*[m.data_ptr() for m in ms])
    _lib.call("kai0_prefix_codes", C.addressof(ptrs), len(ms), lm.data_ptr(), B, n_img, T, Hs, q.data_ptr(), k.data_ptr(), p.data_ptr(), _stream())
    return q, k, p


_WS: dict = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): split-K partial tiles.  Stream-ordered reuse is safe because
    every user (GEMM + its reduce) is enqueued on the same stream before the next user."""
    key = (device.index, _stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def pick_split_k(M: int, N: int, K: int, batch: int = 1) -> int:
    """Split-K factor for a GEMM with few 128x128 output tiles: spread the contraction over the chip's ~512 block slots
    (256 CUs x 2 blocks) without making a chunk shorter than 4 K-tiles.  Covers both the long-contraction wgrads
    (K = B*S) and the skinny weight-streaming GEMMs of B=1 inference (M = 50)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 384 or K < 512:
        return 1
    if M <= 128:  # one row of tiles: pure weight streaming, latency-bound -> chunks as short as 2 K-tiles
        return max(1, min(32, 768 // tiles, K // 128))
    # a few hundred to a few thousand rows (the action expert at B = 32, the B = 1 prefix pass): measured on MI355X
    # (a sweep of us per call over split 1..8, round 2): ~400 blocks of 128x128 in total with chunks of >= 768 is the
    # optimum; the former rule (up to 768 blocks, chunks down to 256) cost 20-60 % on these shapes
    return max(1, min(round(400 / tiles), K // 768))


# Split-K of the weight-gradient GEMMs (TN, contraction = B*S rows) whose output has few 256x256 tiles, measured on MI355X
# with a round-3 probe (tools/tn_split_probe.py in the git history; us per launch at split 1/2/3/4/6/8): the best split fills ONE round of the 256 CUs
# (tiles x split <= 256), and fewer, longer chunks win when that leaves CUs idle anyway — a CU that shares the chip with
# fewer running blocks stages its tiles faster.  Exact pi0.5 shapes at B = 32 first, then the rule they follow.
_WGRAD_SPLIT = {(1152, 4304, 24576): 3, (4304, 1152, 24576): 3, (3456, 1152, 24576): 2, (1152, 1152, 24576): 6,
                (2560, 2048, 30976): 3, (2048, 2048, 30976): 4}  # fmt: skip


def pick_split_k_wgrad(M: int, N: int, K: int) -> int:
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    if t256 >= 200 or K < 4096:
        return pick_split_k(M, N, K)
    return _WGRAD_SPLIT.get((M, N, K)) or max(1, min(6, 256 // t256, K // 2048))


def linear_fwd(x, w, bias=None, residual=None, act=0, want_pre=False, gate=None, gate_rpb=0, out=None):
    """y = epilogue(x @ w.T) for flat bf16 x [M,K], w [N,K]."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    pre = torch.empty((M, N), dtype=BF16, device=x.device) if want_pre else None
    gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act=act, pre_out=pre, residual=residual, ldr=N,
         gate=gate, gate_rpb=gate_rpb, gate_ld=N, split_k=pick_split_k(M, N, K))  # fmt: skip
    return (out, pre) if want_pre else out


def gemm_f32(A, sam, sak, Bm, sbk, sbn, out, M, N, K, bias=None, accumulate=False, split_k=None):
    for t in (A, Bm, out):
        if not t.is_cuda or t.dtype != F32:
            raise _lib.Kai0HipError("gemm_f32: expected f32 CUDA (HIP) tensors; the product path has no CPU fallback")
    if split_k is None:  # long contraction, few 64x64 output tiles: spread K over the chip
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        split_k = 1 if (tiles >= 512 or K < 512) else max(1, min(64, 1024 // tiles, K // 128))
    ws = _workspace(split_k * M * N * 4, A.device) if split_k > 1 else None
    _lib.call("kai0_gemm_f32", A.data_ptr(), sam, sak, Bm.data_ptr(), sbk, sbn, out.data_ptr(), out.stride(0), M, N, K,
              _p(bias), int(accumulate), split_k, _p(ws), ws.numel() if ws is not None else 0, _stream())  # fmt: skip
    return out


def _grad_dst(param, dtype):
    """Where a parameter's gradient should be produced: the slice of the trainer's flat gradient buffer published by
    sharded.py (`_kai0_grad_out`, so no copy is needed afterwards) or a fresh tensor."""
    dst = getattr(param, "_kai0_grad_out", None)
    if dst is not None and dst.shape == param.shape and dst.dtype == dtype:
        ensure = getattr(param, "_kai0_grad_ensure", None)
        if ensure is not None:
            ensure()  # fsdp staging buffer: make sure its storage exists (sharded.py _ensure_grad; a no-op when resident)
        return dst
    return torch.empty(param.shape, dtype=dtype, device=param.device)


def _grad_ret(param, g):
    """What a backward returns for `param`: if `g` was produced in the trainer's flat buffer, tell the trainer (its
    bucket bookkeeping runs now) and return None, so autograd neither clones nor accumulates it; otherwise g."""
    if g is None or param is None:
        return g
    dst = getattr(param, "_kai0_grad_out", None)
    if dst is not None and g.data_ptr() == dst.data_ptr():
        if _DUAL:
            _note_side_gradient(param)
        param._kai0_grad_done()
        return None
    return g


# ------------------------------------------------------------------------------------------ second HIP stream
# model.forward_joint runs the action expert's chain on a second stream (and autograd its backward): (main, side) per device.
_DUAL: dict = {}
_JOIN_PENDING: dict = {}


def side_stream(device):
    """The second stream of `device`, paired with the CURRENT stream as its main one."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(idx)
    pair = _DUAL.get(idx)
    if pair is None or pair[0] != main:
        pair = _DUAL[idx] = (main, pair[1] if pair is not None else torch.cuda.Stream(device=idx))
    return pair[1]


def stream_pair(device):
    """(main, side) if the second stream has been used on `device`, else None."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _DUAL.get(idx)


def join_streams(device) -> None:
    """The current stream waits for both streams of the pair (end of a backward pass, before the optimizer / a collective)."""
    pair = stream_pair(device)
    if pair is not None:
        cur = torch.cuda.current_stream(pair[0].device_index)
        for st in pair:
            if st != cur:
                cur.wait_stream(st)
    flush_deferred(device.index if device.index is not None else torch.cuda.current_device())


def reset_backward_state() -> None:
    """Start of a training step: forget a main<-side join that a previous backward pass queued but never ran (the pass raised
    before its final callbacks) — otherwise no later backward would queue the join again."""
    _DEFERRED.clear()  # reductions an aborted backward queued: their destinations belong to that pass
    for idx in list(_JOIN_PENDING):
        if _JOIN_PENDING[idx]:
            _JOIN_PENDING[idx] = False
            pair = _DUAL.get(idx)
            if pair is not None:
                pair[0].wait_stream(pair[1])  # whatever that pass left on the second stream is ordered before this step


def _note_side_gradient(param) -> None:
    """A gradient was just written in place (into the trainer's flat buffer) by a backward node running on the second stream:
    have the main stream pick it up when this backward pass ends (autograd itself only synchronises gradients it accumulates)."""
    pair = stream_pair(param.device)
    if pair is None or torch.cuda.current_stream(pair[0].device_index) != pair[1]:
        return
    idx = pair[0].device_index
    if _JOIN_PENDING.get(idx):
        return

    def join():
        _JOIN_PENDING[idx] = False
        pair[0].wait_stream(pair[1])

    try:
        torch.autograd.Variable._execution_engine.queue_callback(join)
        _JOIN_PENDING[idx] = True
    except RuntimeError:  # not inside a backward pass
        pair[0].wait_stream(pair[1])


# ---------------------------------------------------------------------------------- deferred final reductions
# The last step of every norm-weight / bias gradient is a column sum over per-block partial rows: ~9 us launches, 254 of them
# per pi0.5 step.  When the destination is the trainer's flat gradient buffer nothing reads it before the bucket's collective or
# the optimizer, so those sums are queued and run 32 per launch (kai0_reduce_partials_batch): flushed by join_streams (which
# precedes every collective and the optimizer) and by a callback at the end of the backward pass.  Gradients handed back to
# autograd as tensors are NOT deferred: AccumulateGrad may read them as soon as the node returns.
_DEFER_REDUCE = True  # (round 5: the environment switch that launched every final sum on its own is gone)
_DEFERRED: dict = {}  # device index -> [(ReduceItem fields, tensors kept alive, producing stream)]


def _reduce_partials(param, part, nb: int, ncols: int, ld: int, out, col0: int = 0):
    """out[c] = sum_b part[b][col0 + c]; queued when `out` is `param`'s slice of the flat gradient buffer."""
    src = part.data_ptr() + 4 * col0
    f32 = int(out.dtype == F32)
    dst = getattr(param, "_kai0_grad_out", None)
    if not (_DEFER_REDUCE and dst is not None and out.data_ptr() == dst.data_ptr() and out.is_cuda):
        _lib.call("kai0_reduce_partials", src, nb, ncols, ld, out.data_ptr(), f32, _stream())
        return
    idx = out.device.index
    q = _DEFERRED.setdefault(idx, [])
    q.append(((src, out.data_ptr(), ld, nb, ncols, f32, 0), (part, out), torch.cuda.current_stream(idx)))
    if len(q) == 1:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_deferred(idx))
        except RuntimeError:  # not inside a backward pass
            flush_deferred(idx)


def flush_deferred(idx=None) -> None:
    """Run the queued reductions of device `idx` (all devices if None) on the current stream, behind their producers."""
    for i in list(_DEFERRED) if idx is None else [idx]:
        q = _DEFERRED.get(i)
        if not q:
            continue
        _DEFERRED[i] = []
        cur = torch.cuda.current_stream(i)
        for st in {e[2] for e in q}:
            if st != cur:
                cur.wait_stream(st)
        items = (_lib.ReduceItem * len(q))(*[_lib.ReduceItem(*e[0]) for e in q])
        with torch.cuda.device(i):
            _lib.call("kai0_reduce_partials_batch", C.addressof(items), len(q), cur.cuda_stream)
        for e in q:  # partial buffers allocated under another stream: not to be reused before this launch has read them
            if e[2] != cur:
                e[1][0].record_stream(cur)


def _bias_grads(dy, M: int, N: int, ld: int, biases):
    """Column sums of dy [M, N] (row stride ld) into the gradients of `biases` = [(bias, first column, width)]: one pass over dy
    into per-block partials, then one (queued) final sum per bias.  Returns what the backward hands to autograd per bias."""
    scratch = torch.empty((COLSUM_BLOCKS, N), dtype=F32, device=dy.device)
    used = C.c_int(0)
    _lib.call("kai0_colsum_partials_bf16", dy.data_ptr(), M, N, ld, scratch.data_ptr(), COLSUM_BLOCKS, C.addressof(used), _stream())
    out = []
    for b, c0, width in biases:
        db = _grad_dst(b, b.dtype)
        _reduce_partials(b, scratch, used.value, width, N, db, col0=c0)
        out.append(_grad_ret(b, db))
    return out


# ----------------------------------------------------------------------------------------- autograd shims
class LinearFn(torch.autograd.Function):
    """bf16 Linear with fused bias / GELU-tanh / residual epilogue (nn.Linear + F.gelu + `x + y`)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, act: int):
        _chk(x, BF16, "linear.x")
        _chk(w, BF16, "linear.w")
        need_pre = act == 1 and (x.requires_grad or w.requires_grad)
        r = linear_fwd(x, w, bias, residual, act, want_pre=need_pre)
        out, pre = r if need_pre else (r, None)
        ctx.save_for_backward(x, w, pre)
        ctx.bias = bias  # only its grad destination / dtype are used
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.has_res = residual is not None
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, pre = ctx.saved_tensors
        dout = dout.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dres = dout if ctx.has_res else None
        dy = dout
        if ctx.act == 1:
            dy = torch.empty_like(dout)
            _lib.call("kai0_gelu_bwd", dout.data_ptr(), pre.data_ptr(), dy.data_ptr(), dout.numel(), _stream())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=BF16, device=x.device)
            if M >= 4096 and N % 8 == 0 and K % 8 == 0:
                # many rows: transpose W once (2 x |W| bytes, ~0.2 % of what the GEMM streams) and run dgrad in the
                # K-contiguous (NT) form, which is ~30 % faster than reading W through ds_read_b64_tr_b16
                wt = transpose(w)
                gemm(dy, wt, dx, M=M, N=K, K=N, a_kc=True, b_kc=True, lda=N, ldb=N, ldc=K)
                del wt
            else:
                # dx[M,K] = dy[M,N] @ w[N,K]  (A K-contig over N; B stored [N][K] = [contraction][cols])
                gemm(dy, w, dx, M=M, N=K, K=N, a_kc=True, b_kc=False, lda=N, ldb=K, ldc=K, split_k=pick_split_k(M, K, N))
        if ctx.needs_input_grad[1]:
            # the sharded trainer publishes each parameter's slice of its flat gradient buffer: write dW there
            # directly, so no gradient copy is needed afterwards (sharded.py)
            dw = _grad_dst(w, BF16)
            # dw[N,K] = dy[M,N]^T @ x[M,K]  (both stored [contraction][cols])
            gemm(dy, x, dw, M=N, N=K, K=M, a_kc=False, b_kc=False, lda=N, ldb=K, ldc=K, split_k=pick_split_k_wgrad(N, K, M), nt_out=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            (db,) = _bias_grads(dy, M, N, N, [(ctx.bias, 0, N)])
        return dx, _grad_ret(w, dw), db, dres, None


def transpose(x: torch.Tensor) -> torch.Tensor:
    """bf16 [R, C] -> [C, R] (kai0_transpose_bf16)."""
    _chk(x, BF16, "transpose.x")
    R, Cc = x.shape
    y = torch.empty((Cc, R), dtype=BF16, device=x.device)
    _lib.call("kai0_transpose_bf16", x.data_ptr(), y.data_ptr(), R, Cc, _stream())
    return y


def linear(x, w, bias=None, residual=None, act=0):
    return LinearFn.apply(x, w, bias, residual, act)


def fused_columns(rows: int, widths, device):
    """Column slices [rows, w_i] of one fresh bf16 [rows, sum(w)] buffer.  A backward that hands these to autograd lets
    LinearMultiFn recognise them (`_as_fused`) and run its dgrad / wgrad on the whole buffer without a concatenation."""
    buf = torch.empty((rows, sum(widths)), dtype=BF16, device=device)
    out, c = [], 0
    for w in widths:
        out.append(buf[:, c : c + w])
        c += w
    return out


def _as_fused(ts, widths):
    """The [rows, sum(widths)] tensor whose column slices `ts` are (in order), or None."""
    t0 = ts[0]
    if t0 is None or t0.dim() != 2:
        return None
    total = sum(widths)
    off = t0.storage_offset()
    for t, w in zip(ts, widths):
        if (t is None or t.dtype != t0.dtype or t.dim() != 2 or t.shape != (t0.shape[0], w) or t.stride() != (total, 1)
                or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr() or t.storage_offset() != off):  # fmt: skip
            return None
        off += w
    return t0.as_strided((t0.shape[0], total), (total, 1), t0.storage_offset())


_FUSE_MULTI = True  # q|k|v as one stacked GEMM whenever the widths allow it (round 5: no environment switch)


class LinearMultiFn(torch.autograd.Function):
    """Several Linears of the same input (q/k/v projections): y_i = x W_i^T (+ b_i), as ONE GEMM over the stacked weight
    [sum N_i, K] whose epilogue routes the column ranges to the separate outputs; the backward is one dgrad (K = sum N_i)
    and one wgrad over the stacked dY.  The weights stay separate parameters (state-dict contract): stacking them is a
    copy of the weights only (MBs per layer against GEMMs of GFLOPs), and so is splitting the stacked gradient."""

    @staticmethod
    def forward(ctx, x, n: int, *wb):
        _chk(x, BF16, "linear_multi.x")
        ws, bs = wb[:n], wb[n:]
        widths = tuple(int(w.shape[0]) for w in ws)
        has_b = [b is not None for b in bs]
        fuse = (_FUSE_MULTI and 2 <= n <= 3 and all(w % 8 == 0 for w in widths) and (all(has_b) or not any(has_b))
                and (not any(has_b) or len({b.dtype for b in bs}) == 1))  # fmt: skip
        ctx.n, ctx.biases, ctx.widths, ctx.fuse = n, bs, widths, fuse
        if not fuse:
            outs = tuple(linear_fwd(x, w, b) for w, b in zip(ws, bs))
            ctx.save_for_backward(x, *ws)
            return outs
        M, K = x.shape
        Nt = sum(widths)
        # (a trainer's flat parameter buffer holds q / k / v back to back: the stacked weight is then a view, not a copy)
        wcat = _as_rows([w.detach() for w in ws])
        if wcat is None:
            wcat = torch.cat(ws, dim=0)
        bcat = None
        if all(has_b):
            bcat = _as_rows([b.detach() for b in bs])
            if bcat is None:
                bcat = torch.cat(bs, dim=0)
        outs = tuple(torch.empty((M, w), dtype=BF16, device=x.device) for w in widths)
        segs, c = [], 0
        for o, w in zip(outs, widths):
            segs.append((o, w, c))
            c += w
        gemm(x, wcat, outs[0], M=M, N=Nt, K=K, lda=K, ldb=K, ldc=widths[0], bias=bcat, segs=segs, split_k=pick_split_k(M, Nt, K))
        ctx.save_for_backward(x, wcat, *ws)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        n, bs, widths = ctx.n, ctx.biases, ctx.widths
        if not ctx.fuse or any(d is None for d in douts):
            x, *rest = ctx.saved_tensors
            ws = rest[1:] if ctx.fuse else rest
            return LinearMultiFn._backward_separate(ctx, x, ws, douts)
        x, wcat, *ws = ctx.saved_tensors
        M, K = x.shape
        Nt = sum(widths)
        dev = x.device
        dy = _as_fused(douts, widths)
        if dy is None:
            dy = torch.cat([d if d.dtype == BF16 else d.to(BF16) for d in douts], dim=1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=BF16, device=dev)
            if M >= 4096:
                wt = transpose(wcat)  # [K, Nt]: K-contiguous operand for the fast NT schedule
                gemm(dy, wt, dx, M=M, N=K, K=Nt, lda=Nt, ldb=Nt, ldc=K)
                del wt
            else:
                gemm(dy, wcat, dx, M=M, N=K, K=Nt, a_kc=True, b_kc=False, lda=Nt, ldb=K, ldc=K)
        dws, dbs = [None] * n, [None] * n
        if any(ctx.needs_input_grad[2 + i] for i in range(n)):
            dsts = [_grad_dst(w, BF16) for w in ws]
            dwcat = _as_rows(dsts)
            tmp = dwcat is None
            if tmp:
                dwcat = torch.empty((Nt, K), dtype=BF16, device=dev)
            gemm(dy, x, dwcat, M=Nt, N=K, K=M, a_kc=False, b_kc=False, lda=Nt, ldb=K, ldc=K, split_k=pick_split_k_wgrad(Nt, K, M))
            r = 0
            for i, (w, dst) in enumerate(zip(ws, dsts)):
                if tmp:
                    dst.copy_(dwcat[r : r + widths[i]])
                r += widths[i]
                dws[i] = _grad_ret(w, dst)
        if bs[0] is not None and any(ctx.needs_input_grad[2 + n + i] for i in range(n)):
            offs = [sum(widths[:i]) for i in range(n)]
            dbs = _bias_grads(dy, M, Nt, Nt, [(b, offs[i], widths[i]) for i, b in enumerate(bs)])
        return (dx, None, *dws, *dbs)

    @staticmethod
    def _backward_separate(ctx, x, ws, douts):
        n, bs = ctx.n, ctx.biases
        M, K = x.shape
        dev = x.device
        dws, dbs = [None] * n, [None] * n
        dx = torch.empty((M, K), dtype=BF16, device=dev) if ctx.needs_input_grad[0] else None
        first = True
        for i, (w, b, dy) in enumerate(zip(ws, bs, douts)):
            if dy is None:
                continue
            dy = dy.contiguous()
            N = w.shape[0]
            if dx is not None:
                if M >= 4096 and N % 8 == 0:
                    wt = transpose(w)
                    gemm(dy, wt, dx, M=M, N=K, K=N, lda=N, ldb=N, ldc=K, accumulate=not first)
                    del wt
                else:
                    gemm(dy, w, dx, M=M, N=K, K=N, a_kc=True, b_kc=False, lda=N, ldb=K, ldc=K, accumulate=not first)
                first = False
            if ctx.needs_input_grad[2 + i]:
                dw = _grad_dst(w, BF16)
                gemm(dy, x, dw, M=N, N=K, K=M, a_kc=False, b_kc=False, lda=N, ldb=K, ldc=K, split_k=pick_split_k_wgrad(N, K, M), nt_out=True)
                dws[i] = _grad_ret(w, dw)
            if b is not None and ctx.needs_input_grad[2 + n + i]:
                (dbs[i],) = _bias_grads(dy, M, N, N, [(b, 0, N)])
        if dx is not None and first:
            dx.zero_()
        return (dx, None, *dws, *dbs)


def _as_rows(ts):
    """The [sum rows, K] tensor made of the contiguous 2-D tensors `ts` if they lie back to back in one storage, or None."""
    t0 = ts[0]
    ptr = t0.data_ptr()
    for t in ts:
        if (not t.is_contiguous() or t.dtype != t0.dtype or t.shape[1:] != t0.shape[1:] or t.data_ptr() != ptr
                or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr()):  # fmt: skip
            return None
        ptr += t.numel() * t.element_size()
    return t0.as_strided((sum(t.shape[0] for t in ts), *t0.shape[1:]), t0.stride(), t0.storage_offset())


def linear_multi(x, weights, biases=None):
    n = len(weights)
    biases = list(biases) if biases is not None else [None] * n
    return LinearMultiFn.apply(x, n, *weights, *biases)


class LinearF32Fn(torch.autograd.Function):
    """f32 Linear (exact-f32 MFMA): adaRMS dense, time MLP, action in/out projections."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _chk(x, F32, "linear_f32.x")
        _chk(w, F32, "linear_f32.w")
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty((M, N), dtype=F32, device=x.device)
        if M <= 16 and K % 4 == 0:
            _lib.call("kai0_linear_rows_f32", x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr(), N, M, N, K, _stream())
        else:
            gemm_f32(x, K, 1, w, 1, K, out, M, N, K, bias=bias)
        ctx.save_for_backward(x, w)
        ctx.bias = bias
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        dout = dout.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=F32, device=x.device)
            gemm_f32(dout, N, 1, w, K, 1, dx, M, K, N)  # dx[m,k] = sum_n dout[m,n] w[n,k]
        if ctx.needs_input_grad[1]:
            dw = _grad_dst(w, F32)
            gemm_f32(dout, 1, N, x, K, 1, dw, N, K, M)  # dw[n,k] = sum_m dout[m,n] x[m,k]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _grad_dst(ctx.bias, F32)
            ones = torch.ones((1, M), dtype=F32, device=x.device)
            gemm_f32(ones, M, 1, dout, N, 1, db.view(1, N), 1, N, M)  # db[n] = sum_m dout[m,n]
        return dx, _grad_ret(w, dw), _grad_ret(ctx.bias, db)


def linear_f32(x, w, bias=None):
    return LinearF32Fn.apply(x, w, bias)


class RMSNormFn(torch.autograd.Function):
    """(x, y = rmsnorm(x)).  x is handed back as the first output so that the residual branch hangs off THIS node: the
    backward then receives both gradients and adds them inside the norm-backward kernel (no separate add pass)."""

    @staticmethod
    def forward(ctx, x, w, eps: float):
        _chk(x, BF16, "rmsnorm.x")
        _chk(w, F32, "rmsnorm.w")
        rows, D = x.shape
        y = torch.empty_like(x)
        rstd = torch.empty((rows,), dtype=F32, device=x.device)
        _lib.call("kai0_rmsnorm_fwd", x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows, D, eps, _stream())
        ctx.save_for_backward(x, w, rstd)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, w, rstd = ctx.saved_tensors
        rows, D = x.shape
        if dy is None:
            return dres, None, None
        dy = dy.contiguous()
        dres = dres.contiguous() if dres is not None else None
        dx = torch.empty_like(x)
        nb = NORM_PARTIAL_BLOCKS
        part = torch.empty((nb, D), dtype=F32, device=x.device)  # one partial row per block
        _lib.call("kai0_rmsnorm_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                  part.data_ptr(), nb, _p(dres), rows, D, _stream())  # fmt: skip
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _grad_dst(w, F32)
            _reduce_partials(w, part, nb, D, D, dw)
        return dx, _grad_ret(w, dw), None


def rmsnorm_res(x, w, eps=1e-6):
    """-> (x for the residual branch, rmsnorm(x))."""
    return RMSNormFn.apply(x, w, eps)


def rmsnorm(x, w, eps=1e-6):
    return RMSNormFn.apply(x, w, eps)[1]


class AdaRMSFn(torch.autograd.Function):
    """adaRMSNorm given the precomputed modulation `mod` = dense(cond) [B, 3D] f32.  Returns (x, y, gate); x is passed
    through for the residual branch (see RMSNormFn)."""

    @staticmethod
    def forward(ctx, x, mod, rows_per_batch: int, eps: float):
        _chk(x, BF16, "adarms.x")
        _chk(mod, F32, "adarms.mod")
        rows, D = x.shape
        Bn = rows // rows_per_batch
        y = torch.empty_like(x)
        gate = torch.empty((Bn, D), dtype=BF16, device=x.device)
        rstd = torch.empty((rows,), dtype=F32, device=x.device)
        _lib.call("kai0_adarms_fwd", x.data_ptr(), mod.data_ptr(), y.data_ptr(), gate.data_ptr(), rstd.data_ptr(), rows,
                  rows_per_batch, D, eps, _stream())  # fmt: skip
        ctx.save_for_backward(x, mod, rstd)
        ctx.rpb = rows_per_batch
        return x.view_as(x), y, gate

    @staticmethod
    def backward(ctx, dres, dy, dgate):
        x, mod, rstd = ctx.saved_tensors
        rows, D = x.shape
        if dy is None:
            dy = torch.zeros_like(x)
        dy = dy.contiguous()
        dgate = dgate.contiguous() if dgate is not None else None
        dres = dres.contiguous() if dres is not None else None
        dx = torch.empty_like(x)
        dmod = torch.empty_like(mod)
        _lib.call("kai0_adarms_bwd", dy.data_ptr(), _p(dgate), x.data_ptr(), mod.data_ptr(), rstd.data_ptr(),
                  dx.data_ptr(), dmod.data_ptr(), _p(dres), rows, ctx.rpb, D, _stream())  # fmt: skip
        return dx, dmod, None, None


def adarms_res(x, mod, rows_per_batch, eps=1e-6):
    """-> (x for the residual branch, y, gate)."""
    return AdaRMSFn.apply(x, mod, rows_per_batch, eps)


def adarms(x, mod, rows_per_batch, eps=1e-6):
    _, y, gate = AdaRMSFn.apply(x, mod, rows_per_batch, eps)
    return y, gate


class LayerNormFn(torch.autograd.Function):
    """(x, y = layer_norm(x)); x passed through for the residual branch (see RMSNormFn)."""

    @staticmethod
    def forward(ctx, x, w, b, eps: float):
        _chk(x, BF16, "layernorm.x")
        rows, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((rows,), dtype=F32, device=x.device)
        rstd = torch.empty((rows,), dtype=F32, device=x.device)
        _lib.call("kai0_layernorm_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), rows, D, eps, _stream())  # fmt: skip
        ctx.save_for_backward(x, w, b, mean, rstd)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, w, b, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        dy = dy.contiguous()
        dres = dres.contiguous() if dres is not None else None
        rows, D = x.shape
        dx = torch.empty_like(x)
        nb = NORM_PARTIAL_BLOCKS
        part = torch.empty((nb, 2 * D), dtype=F32, device=x.device)  # one partial row per block
        _lib.call("kai0_layernorm_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  dx.data_ptr(), part.data_ptr(), nb, _p(dres), rows, D, _stream())  # fmt: skip
        dw, db = _grad_dst(w, w.dtype), _grad_dst(b, b.dtype)
        _reduce_partials(w, part, nb, D, 2 * D, dw)
        _reduce_partials(b, part, nb, D, 2 * D, db, col0=D)
        return dx, _grad_ret(w, dw), _grad_ret(b, db), None


def layernorm_res(x, w, b, eps=1e-6):
    return LayerNormFn.apply(x, w, b, eps)


def layernorm(x, w, b, eps=1e-6):
    return LayerNormFn.apply(x, w, b, eps)[1]


class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, u):
        _chk(g, BF16, "geglu.g")
        _chk(u, BF16, "geglu.u")
        h = torch.empty_like(g)
        _lib.call("kai0_geglu_fwd", g.data_ptr(), u.data_ptr(), h.data_ptr(), g.numel(), _stream())
        ctx.save_for_backward(g, u)
        return h

    @staticmethod
    def backward(ctx, dh):
        g, u = ctx.saved_tensors
        dh = dh.contiguous()
        dg = torch.empty_like(g)
        du = torch.empty_like(u)
        _lib.call("kai0_geglu_bwd", dh.data_ptr(), g.data_ptr(), u.data_ptr(), dg.data_ptr(), du.data_ptr(), g.numel(),
                  _stream())  # fmt: skip
        return dg, du


def geglu(g, u):
    return GegluFn.apply(g, u)


_GEGLU_PAIR = True  # set_geglu_pair(False): gate GEMM + up GEMM with the act-2 epilogue (tests; the fallback of widths % 32 != 0)


def set_geglu_pair(on: bool) -> bool:
    global _GEGLU_PAIR
    old, _GEGLU_PAIR = _GEGLU_PAIR, bool(on)
    return old


class GegluMlpFn(torch.autograd.Function):
    """Gemma MLP `down(gelu_tanh(gate(x)) * up(x)) (+ residual)` (modeling_gemma.py:113-126) as three GEMMs with the
    GeGLU fused into epilogues: forward in the up-projection GEMM (reads g, writes u and h), backward in the
    down-projection dgrad (dh never reaches HBM; the epilogue writes dg and du), and the two dgrads into x accumulate in
    the second GEMM's epilogue instead of a separate add."""

    @staticmethod
    def forward(ctx, x, wg, wu, wd, residual):
        _chk(x, BF16, "geglu_mlp.x")
        M, D = x.shape
        F = wg.shape[0]
        u = torch.empty((M, F), dtype=BF16, device=x.device)
        h = torch.empty((M, F), dtype=BF16, device=x.device)
        if _GEGLU_PAIR and F % 32 == 0:
            # gate | up as ONE GEMM over the two weights, GeGLU in registers (act 6): no read-back of g in an epilogue
            g = torch.empty((M, F), dtype=BF16, device=x.device)
            gemm(x, wg, h, M=M, N=F, K=D, lda=D, ldb=D, ldc=F, act=6, B2=wu, pre_out=g, pre_out2=u)
        else:
            g = linear_fwd(x, wg)
            gemm(x, wu, h, M=M, N=F, K=D, lda=D, ldb=D, ldc=F, act=2, pre_out=u, aux1=g, split_k=1)
        out = linear_fwd(h, wd, residual=residual)
        ctx.save_for_backward(x, wg, wu, wd, g, u, h)
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wg, wu, wd, g, u, h = ctx.saved_tensors
        dout = dout.contiguous()
        M, D = x.shape
        F = wg.shape[0]
        dev = x.device

        def wgrad(dy, inp, w, n, k):
            dw = _grad_dst(w, BF16)
            gemm(dy, inp, dw, M=n, N=k, K=M, a_kc=False, b_kc=False, lda=n, ldb=k, ldc=k, split_k=pick_split_k_wgrad(n, k, M), nt_out=True)
            return dw

        dwd = wgrad(dout, h, wd, D, F) if ctx.needs_input_grad[3] else None
        # dh = dout @ wd (through wd^T, K-contiguous) with the GeGLU backward in the epilogue: C = dg, pre_out = du
        dg = torch.empty((M, F), dtype=BF16, device=dev)
        du = torch.empty((M, F), dtype=BF16, device=dev)
        wdt = transpose(wd)  # [F, D]
        gemm(dout, wdt, dg, M=M, N=F, K=D, lda=D, ldb=D, ldc=F, act=3, pre_out=du, aux1=g, aux2=u)
        del wdt
        dwg = wgrad(dg, x, wg, F, D) if ctx.needs_input_grad[1] else None
        dwu = wgrad(du, x, wu, F, D) if ctx.needs_input_grad[2] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, D), dtype=BF16, device=dev)
            wgt = transpose(wg)  # [D, F]
            gemm(dg, wgt, dx, M=M, N=D, K=F, lda=F, ldb=F, ldc=D, split_k=1)
            wut = transpose(wu)
            gemm(du, wut, dx, M=M, N=D, K=F, lda=F, ldb=F, ldc=D, accumulate=True, split_k=1)
            del wgt, wut
        return dx, _grad_ret(wg, dwg), _grad_ret(wu, dwu), _grad_ret(wd, dwd), (dout if ctx.has_res else None)


_PAD_MLP_ROWS = True  # [M, F] intermediates with rows padded to 128 bytes (F = 4304: +15-18 % on the GEMMs that touch them)


class GeluMlpFn(torch.autograd.Function):
    """SigLIP's MLP `fc2(gelu_tanh(fc1(x))) + residual` (modeling_siglip.py:348-362 + the residual add of :476-478) as one
    autograd node, so that the [M, F] intermediates (fc1 pre-activation, its GELU, and their gradients) can live in
    buffers whose rows are padded to a multiple of 128 bytes: F = 4304 gives 8608-byte rows, and GEMM tiles that start
    mid-line cost the K = 4304 / N = 4304 GEMMs 15-18 % (round-1 probe tools/gemm_align_probe.py, git history)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual):
        _chk(x, BF16, "gelu_mlp.x")
        M, D = x.shape
        F = w1.shape[0]
        ldh = round_up(F, 64) if _PAD_MLP_ROWS else F
        dev = x.device
        pre = torch.empty((M, ldh), dtype=BF16, device=dev)
        h = torch.empty((M, ldh), dtype=BF16, device=dev)
        gemm(x, w1, h, M=M, N=F, K=D, lda=D, ldb=D, ldc=ldh, bias=b1, act=1, pre_out=pre, split_k=1)
        out = torch.empty((M, w2.shape[0]), dtype=BF16, device=dev)
        gemm(h, w2, out, M=M, N=w2.shape[0], K=F, lda=ldh, ldb=F, ldc=w2.shape[0], bias=b2, residual=residual,
             ldr=w2.shape[0], split_k=pick_split_k(M, w2.shape[0], F))  # fmt: skip
        ctx.save_for_backward(x, w1, w2, pre, h)
        ctx.biases = (b1, b2)
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2, pre, h = ctx.saved_tensors
        b1, b2 = ctx.biases
        dout = dout.contiguous()
        M, D = x.shape
        F, Do = w1.shape[0], w2.shape[0]
        ldh = pre.shape[1]
        dev = x.device
        big = M >= 4096

        def colsum(dy, N, ld, b):
            return _bias_grads(dy, M, N, ld, [(b, 0, N)])[0]

        dw2 = db2 = dw1 = db1 = dx = None
        if ctx.needs_input_grad[3]:
            dw2 = _grad_dst(w2, BF16)
            gemm(dout, h, dw2, M=Do, N=F, K=M, a_kc=False, b_kc=False, lda=Do, ldb=ldh, ldc=F, split_k=pick_split_k_wgrad(Do, F, M), nt_out=True)
            dw2 = _grad_ret(w2, dw2)
        if b2 is not None and ctx.needs_input_grad[4]:
            db2 = colsum(dout, Do, Do, b2)
        # dpre = (dout @ w2) * gelu'(pre): the GELU backward runs in the dgrad GEMM's epilogue (kai0hip.h act 5)
        dpre = torch.empty((M, ldh), dtype=BF16, device=dev)
        if big:
            w2t = transpose(w2)  # [F, Do]
            gemm(dout, w2t, dpre, M=M, N=F, K=Do, lda=Do, ldb=Do, ldc=ldh, act=5, aux1=pre)  # dh never reaches HBM
            del w2t
        else:
            gemm(dout, w2, dpre, M=M, N=F, K=Do, a_kc=True, b_kc=False, lda=Do, ldb=F, ldc=ldh, act=5, aux1=pre)
        if ctx.needs_input_grad[1]:
            dw1 = _grad_dst(w1, BF16)
            gemm(dpre, x, dw1, M=F, N=D, K=M, a_kc=False, b_kc=False, lda=ldh, ldb=D, ldc=D, split_k=pick_split_k_wgrad(F, D, M), nt_out=True)
            dw1 = _grad_ret(w1, dw1)
        if b1 is not None and ctx.needs_input_grad[2]:
            db1 = colsum(dpre, F, ldh, b1)
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, D), dtype=BF16, device=dev)
            if big:
                w1t = transpose(w1)  # [D, F]
                gemm(dpre, w1t, dx, M=M, N=D, K=F, lda=ldh, ldb=F, ldc=D)
                del w1t
            else:
                gemm(dpre, w1, dx, M=M, N=D, K=F, a_kc=True, b_kc=False, lda=ldh, ldb=D, ldc=D, split_k=pick_split_k(M, D, F))
        return dx, dw1, db1, dw2, db2, (dout if ctx.has_res else None)


def gelu_mlp(x, w1, b1, w2, b2, residual=None):
    return GeluMlpFn.apply(x, w1, b1, w2, b2, residual)


def geglu_mlp(x, wg, wu, wd, residual=None):
    return GegluMlpFn.apply(x, wg, wu, wd, residual)


class GatedResidualFn(torch.autograd.Function):
    """out = x + y * gate[b]  (modeling_gemma.py:209-227), gate [B, D] broadcast over the rows of batch b."""

    @staticmethod
    def forward(ctx, x, y, gate, rows_per_batch: int):
        rows, D = x.shape
        out = torch.empty_like(x)
        _lib.call("kai0_gated_fwd", x.data_ptr(), y.data_ptr(), gate.data_ptr(), out.data_ptr(), rows, rows_per_batch, D,
                  _stream())  # fmt: skip
        ctx.save_for_backward(y, gate)
        ctx.rpb = rows_per_batch
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gate = ctx.saved_tensors
        dout = dout.contiguous()
        rows, D = y.shape
        dy = torch.empty_like(y)
        dgate = torch.empty_like(gate)
        _lib.call("kai0_gated_bwd", dout.data_ptr(), y.data_ptr(), gate.data_ptr(), dy.data_ptr(), dgate.data_ptr(), rows,
                  ctx.rpb, D, _stream())  # fmt: skip
        return dout, dy, dgate, None


def gated_residual(x, y, gate, rows_per_batch):
    return GatedResidualFn.apply(x, y, gate, rows_per_batch)


class SiluF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x, F32, "silu.x")
        y = torch.empty_like(x)
        _lib.call("kai0_silu_fwd_f32", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.call("kai0_silu_bwd_f32", dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream())
        return dx


def silu_f32(x):
    return SiluF32Fn.apply(x)


class CastFn(torch.autograd.Function):
    """dtype cast f32<->bf16 whose backward casts the gradient back (autograd `.to()` semantics)."""

    @staticmethod
    def forward(ctx, x, to_bf16: bool):
        ctx.to_bf16 = to_bf16
        return cast(x, BF16 if to_bf16 else F32)

    @staticmethod
    def backward(ctx, dy):
        return cast(dy.contiguous(), F32 if ctx.to_bf16 else BF16), None


def cast(x: torch.Tensor, dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    if dtype == BF16:
        _chk(x, F32, "cast.x")
        _lib.call("kai0_cast_f32_to_bf16", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    else:
        _chk(x, BF16, "cast.x")
        _lib.call("kai0_cast_bf16_to_f32", x.data_ptr(), y.data_ptr(), x.numel(), _stream())
    return y


def cast_ag(x, dtype):
    if x.dtype == dtype:
        return x
    return CastFn.apply(x, dtype == BF16)


class EmbedFn(torch.autograd.Function):
    """embed_tokens(tokens) * sqrt(D)  (gemma_pytorch.py:88-89; pi0_pytorch.py:213-216) -> [B*T, D] bf16."""

    @staticmethod
    def forward(ctx, table, tokens, scale: float):
        _chk(table, BF16, "embed.table")
        if tokens.dtype != torch.int64:
            raise TypeError("embed.tokens must be int64")
        Bn, T = tokens.shape
        D = table.shape[1]
        out = torch.empty((Bn * T, D), dtype=BF16, device=table.device)
        _lib.call("kai0_embed_gather", table.data_ptr(), tokens.data_ptr(), out.data_ptr(), Bn, T, D, scale, T * D, 0, D,
                  _stream())  # fmt: skip
        ctx.save_for_backward(tokens)
        ctx.table = table
        ctx.shape = table.shape
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        dout = dout.contiguous()
        Bn, T = tokens.shape
        V, D = ctx.shape
        dst = getattr(ctx.table, "_kai0_grad_out", None)
        if dst is not None and dst.shape == (V, D) and dst.dtype == BF16:
            dtable = _grad_dst(ctx.table, BF16)  # pre-zeroed slice of the trainer's flat gradient buffer (fsdp: storage ensured first)
            ctx.table._kai0_grad_accumulates = True  # the scatter-add needs it zeroed again before the next backward
        else:
            dtable = torch.zeros((V, D), dtype=BF16, device=dout.device)
        _lib.call("kai0_embed_grad", dout.data_ptr(), tokens.data_ptr(), dtable.data_ptr(), Bn, T, D, ctx.scale, T * D, 0,
                  D, _stream())  # fmt: skip
        return _grad_ret(ctx.table, dtable), None, None


def embed(table, tokens, scale):
    if tokens.dtype in (torch.int32, torch.int16, torch.uint8):  # nn.Embedding takes int32 indices too (the data pipeline's dtype)
        tokens = tokens.to(torch.int64)
    return EmbedFn.apply(table, tokens.contiguous(), scale)


class PatchEmbedFn(torch.autograd.Function):
    """SigLIP patch embedding: Conv2d(3->D, k=P, s=P) as im2col + exact-f32 GEMM, + position embedding, cast to
    bf16 (modeling_siglip.py:220-226,271-281,777-778).  img f32 [N,3,HW,HW] -> bf16 [N*G*G, D]."""

    @staticmethod
    def forward(ctx, img, w, b, pos, patch: int):
        _chk(img, F32, "patch_embed.img")
        _chk(w, F32, "patch_embed.weight")
        n, Cc, HW, _ = img.shape
        D = w.shape[0]
        G = HW // patch
        Kd = Cc * patch * patch
        rows = n * G * G
        cols = torch.empty((rows, Kd), dtype=F32, device=img.device)
        _lib.call("kai0_patch_im2col", img.data_ptr(), cols.data_ptr(), n, Cc, HW, patch, _stream())
        pe = torch.empty((rows, D), dtype=F32, device=img.device)
        w2 = w.view(D, Kd)
        gemm_f32(cols, Kd, 1, w2, 1, Kd, pe, rows, D, Kd, bias=b)
        out = torch.empty((rows, D), dtype=BF16, device=img.device)
        _lib.call("kai0_add_pos_cast", pe.data_ptr(), pos.data_ptr(), out.data_ptr(), rows, G * G, D, _stream())
        ctx.save_for_backward(cols)
        ctx.params = (w, b, pos)
        ctx.dims = (n, G, D, Kd, tuple(w.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        (cols,) = ctx.saved_tensors
        n, G, D, Kd, wshape = ctx.dims
        rows = n * G * G
        d32 = cast(dout.contiguous(), F32)
        pw, pb, ppos = ctx.params
        dw = _grad_dst(pw, F32).view(D, Kd)
        gemm_f32(d32, 1, D, cols, Kd, 1, dw, D, Kd, rows)  # dw[d,k] = sum_r d32[r,d] cols[r,k]
        db = _grad_dst(pb, F32)
        ones = torch.ones((1, rows), dtype=F32, device=dout.device)
        gemm_f32(ones, rows, 1, d32, D, 1, db.view(1, D), 1, D, rows)
        # dpos[p,d] = sum_n d32[n*G*G + p, d]
        GG = G * G
        dpos = _grad_dst(ppos, F32)
        if n == 1:
            dpos.copy_(d32)
        else:
            ones_n = torch.ones((1, n), dtype=F32, device=dout.device)
            # view d32 as [n][GG*D]: dpos_flat[j] = sum_n d32[n, j]
            gemm_f32(ones_n, n, 1, d32.view(n, GG * D), GG * D, 1, dpos.view(1, GG * D), 1, GG * D, n)
        return None, _grad_ret(pw, dw.view(wshape)), _grad_ret(pb, db), _grad_ret(ppos, dpos), None


def patch_embed(img, w, b, pos, patch):
    return PatchEmbedFn.apply(img, w, b, pos, patch)


# --------------------------------------------------------------------------------------------- attention
def _copy_rows(src, dst, Bn, rows, D, sbs, sr0, sld, dbs, dr0, dld):
    _lib.call("kai0_copy_rows_bf16", src.data_ptr(), dst.data_ptr(), Bn, rows, D, sbs, sr0, sld, dbs, dr0, dld, _stream())


def rope_(x, pos, inv_freq, Bn, S, s_ld, row0, H, HD, inverse=False):
    _lib.call("kai0_rope_inplace", x.data_ptr(), pos.data_ptr(), inv_freq.data_ptr(), Bn, S, s_ld, row0, H, HD,
              int(inverse), _stream())  # fmt: skip


def rope2_(x, H, x2, H2, pos, inv_freq, Bn, S, s_ld, row0, HD):
    """kai0_rope_inplace2: rotate q (H heads) and k (H2 heads) of one layer in place with one launch."""
    _lib.call("kai0_rope_inplace2", x.data_ptr(), H, x2.data_ptr(), H2, pos.data_ptr(), inv_freq.data_ptr(), Bn, S, s_ld, row0, HD,
              _stream())


def rope_copy(src_t, dst_t, pos, inv_freq, Bn, S, H, HD, *, src, dst, pos_bs, pos_off=0, inverse=False):
    """dst[b][dst_row0 + s] = rope(src[b][src_row0 + s], pos[b][pos_off + s]) (kai0_rope_copy).  `src` / `dst` =
    (batch stride, row stride, first row) in elements / rows of the tensors' data pointers (which may be column slices)."""
    sbs, sld, sr0 = src
    dbs, dld, dr0 = dst
    _lib.call("kai0_rope_copy", src_t.data_ptr() + 2 * sr0 * sld, dst_t.data_ptr() + 2 * dr0 * dld, pos.data_ptr() + 4 * pos_off,
              inv_freq.data_ptr(), Bn, S, H, HD, sbs, sld, dbs, dld, pos_bs, int(inverse), _stream())  # fmt: skip


def pack_rows(parts, HD: int, pos=None, pos_bs: int = 0, inv_freq=None):
    """kai0_pack_rows: up to 12 strided row moves in one launch.  parts = [(src | None, src element offset, dst, dst element offset,
    (src batch stride, src row stride), (dst batch stride, dst row stride), B, rows, cols, mode, pos_off)] with mode 0 copy,
    1 RoPE, 2 inverse RoPE, 3 zero fill."""
    for c0 in range(0, len(parts), 12):
        chunk = parts[c0 : c0 + 12]
        arr = (_lib.PackPart * len(chunk))()
        for a, (src, soff, dst, doff, (sbs, sld), (dbs, dld), Bn, rows, cols, mode, pos_off) in zip(arr, chunk):
            a.src = None if src is None else src.data_ptr() + 2 * soff
            a.dst = dst.data_ptr() + 2 * doff
            a.src_bs, a.src_ld, a.dst_bs, a.dst_ld = sbs, sld, dbs, dld
            a.B, a.rows, a.cols, a.mode, a.pos_off = Bn, rows, cols, mode, pos_off
        _lib.call("kai0_pack_rows", C.addressof(arr), len(chunk), _p(pos), pos_bs, _p(inv_freq), HD, _stream())


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def attn_fwd_keysplit(Q, K, V, O, *, rows, Sk, HD, H, q0, ldk, ldv, qcode, kcode, scale, q_off, o_off, parts: int = 4):
    """One batch entry's masked attention with the keys cut into `parts` equal ranges (kai0_attn_fwd over the ranges as batch
    entries + kai0_attn_combine): for grids whose row blocks alone cannot fill the chip (the B = 1 prefix pass).  Q rows are the folded
    (position, head) rows at Q + q_off, contiguous [rows, HD]; O likewise at O + o_off; qcode / kcode 1-D views of this entry's codes."""
    if Sk % parts != 0:
        raise ValueError("attn_fwd_keysplit: Sk must be a multiple of the number of key ranges")
    rng = Sk // parts
    dev = Q.device
    o_parts = torch.empty((parts, rows, HD), dtype=BF16, device=dev)
    lse = torch.empty((parts, rows), dtype=torch.float32, device=dev)
    d = AttnDesc()
    d.Q, d.K, d.V, d.O = Q.data_ptr() + 2 * q_off, K.data_ptr(), V.data_ptr(), o_parts.data_ptr()
    d.qcode, d.kcode = _p(qcode), _p(kcode)
    d.rows, d.Sk, d.HD, d.H, d.q0, d.batch, d.batch_inner = rows, rng, HD, H, q0, parts, 1
    d.ldq, d.ldk, d.ldv, d.ldo = HD, ldk, ldv, HD
    d.sQ1, d.sK1, d.sV1, d.sO1 = 0, rng * ldk, rng * ldv, rows * HD
    d.qcode_ld, d.kcode_ld = 0, rng
    d.scale = scale
    d.online = 1
    d.lse, d.s_lse = lse.data_ptr(), rows
    _lib.call("kai0_attn_fwd", C.byref(d), _stream())
    _lib.call("kai0_attn_combine", o_parts.data_ptr(), lse.data_ptr(), O.data_ptr() + 2 * o_off, parts, rows, HD, HD, rows * HD, rows, _stream())


def attn_fwd(Q, K, V, O, P, *, rows, Sk, HD, H=1, q0=0, batch=1, batch_inner=1, ldq, ldk, ldv, ldo, ldp=0, sQ=(0, 0),
             sK=(0, 0), sV=(0, 0), sO=(0, 0), sP=0, qcode=None, kcode=None, scale, q_off=0, o_off=0, lse=None, online=0):
    """kai0_attn_fwd (fused logits + mask + softmax + P V).  P: optional probabilities output (the exact two-pass form);
    lse: optional f32 [batch, >= rows] log-sum-exp output — what the recompute backward needs instead of P."""
    d = AttnDesc()
    d.Q, d.K, d.V, d.O = Q.data_ptr() + 2 * q_off, K.data_ptr(), V.data_ptr(), O.data_ptr() + 2 * o_off
    d.P = _p(P)
    d.qcode, d.kcode = _p(qcode), _p(kcode)
    d.rows, d.Sk, d.HD, d.H, d.q0, d.batch, d.batch_inner = rows, Sk, HD, H, q0, batch, batch_inner
    d.ldq, d.ldk, d.ldv, d.ldo, d.ldp = ldq, ldk, ldv, ldo, ldp
    d.sQ1, d.sQ2 = sQ
    d.sK1, d.sK2 = sK
    d.sV1, d.sV2 = sV
    d.sO1, d.sO2 = sO
    d.sP = sP
    if qcode is not None:
        d.qcode_ld, d.kcode_ld = qcode.stride(0), kcode.stride(0)
    d.scale = scale
    d.online = online
    if lse is not None:
        _chk(lse, torch.float32, "lse")
        d.lse, d.s_lse = lse.data_ptr(), lse.stride(0)
    _lib.call("kai0_attn_fwd", C.byref(d), _stream())


# KAI0_ATTN_STORE_P=1: the round-3 training attention (exact two-pass forward that stores P, backward reads it) for A/B runs
_ATTN_STORE_P = os.environ.get("KAI0_ATTN_STORE_P", "0") == "1"


def mqa_attention_fwd(q_all, k_all, v_all, qcode, kcode, Bn, Sq, q0, Sk, S_ld, H, HD, scale, want_probs=True, want_lse=False):
    """Prefix-LM masked multi-query attention over padded buffers (modeling_gemma.py:230-253), one fused kernel.

    q_all [B, S_ld, H*HD] (query rows q0..q0+Sq used), k_all/v_all [B, S_ld, HD].  The H query heads of a position
    are folded into the row dimension: Q viewed as [Sq*H, HD] per batch entry.
    Returns (att [B, S_ld, H*HD] with rows q0..q0+Sq written, probs [B, Sq*H, S_ld] or None); with want_lse (one-pass forward,
    no probabilities) the second value is lse f32 [B, Sq*H] instead."""
    dev = q_all.device
    M = Sq * H
    if want_lse:
        lse = torch.empty((Bn, M), dtype=torch.float32, device=dev)
        att = torch.empty((Bn, S_ld, H * HD), dtype=BF16, device=dev)
        attn_fwd(q_all, k_all, v_all, att, None, rows=M, Sk=Sk, HD=HD, H=H, q0=q0, batch=Bn, ldq=HD, ldk=HD, ldv=HD, ldo=HD,
                 sQ=(S_ld * H * HD, 0), sK=(S_ld * HD, 0), sV=(S_ld * HD, 0), sO=(S_ld * H * HD, 0), qcode=qcode, kcode=kcode,
                 scale=scale, q_off=q0 * H * HD, o_off=q0 * H * HD, lse=lse)
        return att, lse
    probs = torch.empty((Bn, M, S_ld), dtype=BF16, device=dev) if want_probs else None
    att = torch.empty((Bn, S_ld, H * HD), dtype=BF16, device=dev)
    attn_fwd(q_all, k_all, v_all, att, probs, rows=M, Sk=Sk, HD=HD, H=H, q0=q0, batch=Bn, ldq=HD, ldk=HD, ldv=HD, ldo=HD,
             ldp=S_ld, sQ=(S_ld * H * HD, 0), sK=(S_ld * HD, 0), sV=(S_ld * HD, 0), sO=(S_ld * H * HD, 0), sP=M * S_ld,
             qcode=qcode, kcode=kcode, scale=scale, q_off=q0 * H * HD, o_off=q0 * H * HD)
    return att, probs


def _attn_bwd_split(Bn: int, S_ld: int, HD: int, M: int) -> int:
    """Split of the row contraction of dV = P^T dO / dK = dS^T Q so that (256x256 tiles) x batch x split fills the chip once."""
    t256 = ((S_ld + 255) // 256) * ((HD + 255) // 256) * Bn
    if t256 >= 160 or M < 4096:
        return 1
    return max(1, min(8, -(-256 // t256), M // 2048))


def _joint_attention_grads(cfg, pos, inv_freq, dq_all, dk_all, dv_all):
    """The per-segment gradients out of the joint buffers: inverse RoPE while the rows are gathered (q, k), row copy (v).
    (Takes what it needs of the saved tensors from the caller: under activation checkpointing they can be unpacked once only.)"""
    Bn, S, S_ld, H, HD, seg_lens, scale, recompute = cfg
    dev = dq_all.device
    grads, parts = [], []
    r0 = 0
    for Li in seg_lens:
        W3 = (H + 2) * HD  # dq | dk | dv as column slices of one [Bn*Li, W3] buffer (see fused_columns)
        dq, dk, dv = fused_columns(Bn * Li, (H * HD, HD, HD), dev)
        # the inverse rotation is applied while the segment's rows are gathered out of the joint gradient buffers
        parts.append((dq_all, r0 * H * HD, dq, 0, (S_ld * H * HD, H * HD), (Li * W3, W3), Bn, Li, H * HD, 2, r0))
        parts.append((dk_all, r0 * HD, dk, 0, (dk_all.stride(0), HD), (Li * W3, W3), Bn, Li, HD, 2, r0))
        parts.append((dv_all, r0 * HD, dv, 0, (dv_all.stride(0), HD), (Li * W3, W3), Bn, Li, HD, 0, 0))
        grads += [dq, dk, dv]
        r0 += Li
    pack_rows(parts, HD, pos, S, inv_freq)
    return (None, None, None, None, None, None, None, *grads)


class JointAttentionFn(torch.autograd.Function):
    """The shared attention of one joint layer (gemma_pytorch.py:165-219): concat the per-expert q/k/v over the
    sequence, RoPE, prefix-LM masked MQA attention, split back.

    Inputs are flat [B*S_i, ...] bf16 tensors per segment i (prefix, suffix); outputs are the flat attention
    outputs per segment ([B*S_i, H*HD])."""

    @staticmethod
    def forward(ctx, pos, qcode, kcode, inv_freq, H: int, HD: int, seg_lens: tuple, *qkv):
        nseg = len(seg_lens)
        qs, ks, vs = qkv[0::3], qkv[1::3], qkv[2::3]
        dev = qs[0].device
        S = sum(seg_lens)
        Bn = qs[0].shape[0] // seg_lens[0]
        S_ld = round_up(S, 8)
        q_all = torch.empty((Bn, S_ld, H * HD), dtype=BF16, device=dev)
        k_all = torch.empty((Bn, S_ld, HD), dtype=BF16, device=dev)
        v_all = torch.empty((Bn, S_ld, HD), dtype=BF16, device=dev)
        # one launch: q / k of every segment rotated on their way into the joint buffers, v copied, and the padding rows cleared
        # (only they need defined contents: they enter the dK / dV GEMMs multiplied by P = 0)
        parts = []
        r0 = 0
        for i in range(nseg):
            Li = seg_lens[i]
            parts.append((qs[i], 0, q_all, r0 * H * HD, (Li * qs[i].stride(0), qs[i].stride(0)), (S_ld * H * HD, H * HD), Bn, Li, H * HD, 1, r0))
            parts.append((ks[i], 0, k_all, r0 * HD, (Li * ks[i].stride(0), ks[i].stride(0)), (S_ld * HD, HD), Bn, Li, HD, 1, r0))
            parts.append((vs[i], 0, v_all, r0 * HD, (Li * vs[i].stride(0), vs[i].stride(0)), (S_ld * HD, HD), Bn, Li, HD, 0, 0))
            r0 += Li
        if S_ld > S:
            parts.append((None, 0, q_all, S * H * HD, (0, 0), (S_ld * H * HD, H * HD), Bn, S_ld - S, H * HD, 3, 0))
            parts.append((None, 0, k_all, S * HD, (0, 0), (S_ld * HD, HD), Bn, S_ld - S, HD, 3, 0))
            parts.append((None, 0, v_all, S * HD, (0, 0), (S_ld * HD, HD), Bn, S_ld - S, HD, 3, 0))
        pack_rows(parts, HD, pos, S, inv_freq)
        scale = HD**-0.5
        # one pass, no stored probabilities: the backward recomputes them from lse (kai0_attn_bwd_dq2)
        recompute = not _ATTN_STORE_P and S <= 2048
        att, probs = mqa_attention_fwd(q_all, k_all, v_all, qcode, kcode, Bn, S, 0, S, S_ld, H, HD, scale, want_lse=recompute)
        outs, parts = [], []
        r0 = 0
        for i in range(nseg):
            Li = seg_lens[i]
            o = torch.empty((Bn * Li, H * HD), dtype=BF16, device=dev)
            parts.append((att, r0 * H * HD, o, 0, (S_ld * H * HD, H * HD), (Li * H * HD, H * HD), Bn, Li, H * HD, 0, 0))
            outs.append(o)
            r0 += Li
        pack_rows(parts, HD)
        ctx.save_for_backward(q_all, k_all, v_all, probs, pos, inv_freq, att, qcode, kcode)
        ctx.cfg = (Bn, S, S_ld, H, HD, seg_lens, scale, recompute)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        q_all, k_all, v_all, probs, pos, inv_freq, att, qcode, kcode = ctx.saved_tensors
        Bn, S, S_ld, H, HD, seg_lens, scale, recompute = ctx.cfg
        dev = q_all.device
        M = S * H
        dq_all = torch.empty((Bn, S_ld, H * HD), dtype=BF16, device=dev)
        if recompute:
            # P and dS side by side, dV and dK likewise: the two TN GEMMs that consume them run as ONE batched launch below
            pd = torch.empty((2, Bn, M, S_ld), dtype=BF16, device=dev)
            lse, probs, dscores = probs, pd[0], pd[1]
        datt = torch.empty((Bn, S_ld, H * HD), dtype=BF16, device=dev)
        parts = []
        r0 = 0
        for i, Li in enumerate(seg_lens):
            parts.append((douts[i].contiguous(), 0, datt, r0 * H * HD, (Li * H * HD, H * HD), (S_ld * H * HD, H * HD), Bn, Li, H * HD, 0, 0))
            r0 += Li
        if S_ld > S:
            parts.append((None, 0, datt, S * H * HD, (0, 0), (S_ld * H * HD, H * HD), Bn, S_ld - S, H * HD, 3, 0))
        pack_rows(parts, HD)
        if recompute:
            # query side first: it (re)produces P for the dV GEMM below — S, P = exp(S - lse), dP, dS, dQ in one launch
            d = AttnBwdDesc()
            d.dO, d.O, d.Q, d.K, d.V = datt.data_ptr(), att.data_ptr(), q_all.data_ptr(), k_all.data_ptr(), v_all.data_ptr()
            d.lse, d.qcode, d.kcode = lse.data_ptr(), _p(qcode), _p(kcode)
            d.P, d.dS, d.dQ = probs.data_ptr(), dscores.data_ptr(), dq_all.data_ptr()
            d.batch, d.rows, d.Sk, d.HD, d.H = Bn, M, S, HD, H
            d.ldo, d.ldk, d.ldv, d.ldp = HD, HD, HD, S_ld
            d.sO, d.sK, d.sV, d.sP, d.s_lse = S_ld * H * HD, S_ld * HD, S_ld * HD, M * S_ld, lse.stride(0)
            if qcode is not None:
                d.qcode_ld, d.kcode_ld = qcode.stride(0), kcode.stride(0)
            d.scale = scale
            _lib.call("kai0_attn_bwd_dq2", C.byref(d), _stream())
        # dV[b] [S_ld, HD] = P[b]^T [S_ld, M] @ dO[b] [M, HD]
        # (few output tiles per sample, long contraction over the M = S*H folded rows: split it so that the 256x256 ring
        # schedule gets a block per CU instead of falling back to 128x128 tiles — 578 -> ~1000 TFLOP/s)
        if recompute and (q_all.data_ptr() - datt.data_ptr()) % 16 == 0:
            # dV[b] = P[b]^T dO[b] and dK[b] = dS[b]^T Q[b] as one launch of 2 Bn entries: entry (j, b) reads A = pd[j][b] and
            # B = (dO, Q)[j][b] (the two B operands are separate allocations: their distance is the outer batch stride).  256 tiles of
            # 256 x 256 instead of 2 x 128 with a two-way split-K each: no partial products, no reduction launches
            dkv = torch.empty((2, Bn, S_ld, HD), dtype=BF16, device=dev)
            gemm(pd, datt, dkv, M=S_ld, N=HD, K=M, a_kc=False, b_kc=False, lda=S_ld, ldb=HD, ldc=HD, batch=2 * Bn, batch_inner=Bn,
                 sA=(Bn * M * S_ld, M * S_ld), sB=((q_all.data_ptr() - datt.data_ptr()) // 2, S_ld * H * HD),
                 sC=(Bn * S_ld * HD, S_ld * HD), split_k=_attn_bwd_split(2 * Bn, S_ld, HD, M))  # fmt: skip
            dv_all, dk_all = dkv[0], dkv[1]
            return _joint_attention_grads(ctx.cfg, pos, inv_freq, dq_all, dk_all, dv_all)
        kv_split = _attn_bwd_split(Bn, S_ld, HD, M)
        dv_all = torch.empty((Bn, S_ld, HD), dtype=BF16, device=dev)
        gemm(probs, datt, dv_all, M=S_ld, N=HD, K=M, a_kc=False, b_kc=False, lda=S_ld, ldb=HD, ldc=HD, batch=Bn,
             sA=(M * S_ld, 0), sB=(S_ld * H * HD, 0), sC=(S_ld * HD, 0), split_k=kv_split)  # fmt: skip
        if not recompute:
            dscores = torch.empty_like(probs)
            # stored-P form (KAI0_ATTN_STORE_P=1): one launch — D = rowsum(dO * O), dP = dO V^T in f32, dS written once, dQ = dS K on chip
            _lib.call("kai0_attn_bwd_dq", datt.data_ptr(), att.data_ptr(), probs.data_ptr(), k_all.data_ptr(), v_all.data_ptr(),
                      dscores.data_ptr(), dq_all.data_ptr(), Bn, M, S, HD, HD, HD, HD, S_ld, S_ld * H * HD, S_ld * HD, S_ld * HD,
                      M * S_ld, scale, _stream())  # fmt: skip
        # dK[b] [S_ld, HD] = dS[b]^T [S_ld, M] @ Q[b] [M, HD]
        dk_all = torch.empty((Bn, S_ld, HD), dtype=BF16, device=dev)
        gemm(dscores, q_all, dk_all, M=S_ld, N=HD, K=M, a_kc=False, b_kc=False, lda=S_ld, ldb=HD, ldc=HD, batch=Bn,
             sA=(M * S_ld, 0), sB=(S_ld * H * HD, 0), sC=(S_ld * HD, 0), split_k=kv_split)  # fmt: skip
        return _joint_attention_grads(ctx.cfg, pos, inv_freq, dq_all, dk_all, dv_all)


def joint_attention(pos, qcode, kcode, inv_freq, H, HD, seg_lens, qkv):
    return JointAttentionFn.apply(pos, qcode, kcode, inv_freq, H, HD, tuple(seg_lens), *qkv)


_SIGLIP_BWD_FUSED = True  # (other shapes than 256 x 72 take the GEMM-based backward / the general forward kernel by shape)
_SIGLIP_FWD_DEDICATED = True


def siglip_attn_fwd(q, k, v, out, *, n_img, S, NH, HD, ld_qkv, ld_out, lse=None):
    """kai0_siglip_attn_fwd: the real tower's attention (256 tokens, head_dim 72), one block per (image, head), exact softmax in one
    pass.  q / k / v: [n_img * S, >= NH * HD] bf16 views with row stride ld_qkv (column slices of a stacked buffer allowed)."""
    _lib.call("kai0_siglip_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _p(lse), n_img, S, NH, HD, ld_qkv,
              ld_qkv, ld_qkv, ld_out, S * ld_qkv, S * ld_qkv, S * ld_qkv, S * ld_out, HD**-0.5, _stream())


class SiglipAttentionFn(torch.autograd.Function):
    """Unmasked multi-head attention of SigLIP (modeling_siglip.py:325-345): q,k,v flat [N*S, NH*HD] bf16."""

    @staticmethod
    def forward(ctx, q, k, v, n_img: int, S: int, NH: int, HD: int):
        dev = q.device
        E = NH * HD
        S_ld = round_up(S, 8)
        scale = HD**-0.5
        out = torch.empty((n_img * S, E), dtype=BF16, device=dev)
        # the real tower: no stored probabilities, the fused backward recomputes them from lse (kai0_siglip_attn_bwd2)
        recompute = S == 256 and HD == 72 and S_ld == 256 and _SIGLIP_BWD_FUSED and not _ATTN_STORE_P
        if recompute and _SIGLIP_FWD_DEDICATED and q.stride(0) == k.stride(0) == v.stride(0):
            probs = torch.empty((n_img * NH, S), dtype=torch.float32, device=dev)  # lse
            siglip_attn_fwd(q, k, v, out, n_img=n_img, S=S, NH=NH, HD=HD, ld_qkv=q.stride(0), ld_out=E, lse=probs)
        elif recompute:
            probs = torch.empty((n_img * NH, S), dtype=torch.float32, device=dev)  # lse
            attn_fwd(q, k, v, out, None, rows=S, Sk=S, HD=HD, H=1, batch=n_img * NH, batch_inner=NH, ldq=E, ldk=E, ldv=E, ldo=E,
                     sQ=(S * E, HD), sK=(S * E, HD), sV=(S * E, HD), sO=(S * E, HD), scale=scale, lse=probs)
        else:
            probs = torch.empty((n_img * NH, S, S_ld), dtype=BF16, device=dev)
            attn_fwd(q, k, v, out, probs, rows=S, Sk=S, HD=HD, H=1, batch=n_img * NH, batch_inner=NH, ldq=E, ldk=E, ldv=E,
                     ldo=E, ldp=S_ld, sQ=(S * E, HD), sK=(S * E, HD), sV=(S * E, HD), sO=(S * E, HD), sP=S * S_ld, scale=scale)
        ctx.save_for_backward(q, k, v, probs, out)
        ctx.cfg = (n_img, S, S_ld, NH, HD, scale, recompute)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, probs, out = ctx.saved_tensors
        n_img, S, S_ld, NH, HD, scale, recompute = ctx.cfg
        dev = q.device
        E = NH * HD
        dout = dout.contiguous()
        if recompute:
            dq, dk, dv = fused_columns(q.shape[0], (E, E, E), dev)
            _lib.call("kai0_siglip_attn_bwd2", q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr(), out.data_ptr(),
                      probs.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), n_img, S, NH, HD, 3 * E, scale, _stream())  # fmt: skip
            return dq, dk, dv, None, None, None, None
        if S == 256 and HD == 72 and S_ld == 256 and _SIGLIP_BWD_FUSED:
            # the real tower (so400m/14 @ 224): one block per (image, head) runs the whole backward out of LDS
            # dq | dk | dv are the column slices of one buffer: the fused q|k|v projection backward reads it whole
            dq, dk, dv = fused_columns(q.shape[0], (E, E, E), dev)
            _lib.call("kai0_siglip_attn_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr(), out.data_ptr(),
                      probs.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), n_img, S, NH, HD, S_ld, 3 * E, scale,
                      _stream())  # fmt: skip
            return dq, dk, dv, None, None, None, None
        nb = n_img * NH
        sP = (NH * S * S_ld, S * S_ld)
        sE = (S * E, HD)
        dv = torch.empty_like(v)
        gemm(probs, dout, dv, M=S, N=HD, K=S, a_kc=False, b_kc=False, lda=S_ld, ldb=E, ldc=E, batch=nb, batch_inner=NH,
             sA=sP, sB=sE, sC=sE)  # fmt: skip
        # dS = softmax'(dO V^T) in the GEMM epilogue (act 4); row term = rowsum(dO * O), rows enumerated (image, token, head)
        dsum = rowdot(dout, out, HD)  # [n_img * S * NH]
        dprobs = torch.zeros_like(probs) if S_ld != S else torch.empty_like(probs)
        gemm(dout, v, dprobs, M=S, N=S, K=HD, lda=E, ldb=E, ldc=S_ld, batch=nb, batch_inner=NH, sA=sE, sB=sE, sC=sP, act=4,
             aux1=probs, rowvec=dsum, rv=(S * NH, 1, NH), scale=scale)  # fmt: skip
        dq = torch.empty_like(q)
        gemm(dprobs, k, dq, M=S, N=HD, K=S, a_kc=True, b_kc=False, lda=S_ld, ldb=E, ldc=E, batch=nb, batch_inner=NH,
             sA=sP, sB=sE, sC=sE)  # fmt: skip
        dk = torch.empty_like(k)
        gemm(dprobs, q, dk, M=S, N=HD, K=S, a_kc=False, b_kc=False, lda=S_ld, ldb=E, ldc=E, batch=nb, batch_inner=NH,
             sA=sP, sB=sE, sC=sE)  # fmt: skip
        return dq, dk, dv, None, None, None, None


def siglip_attention(q, k, v, n_img, S, NH, HD):
    return SiglipAttentionFn.apply(q, k, v, n_img, S, NH, HD)


# ---------------------------------------------------------------------------------------- flow matching
def flow_mix(noise, actions, time):
    """x_t = t*noise + (1-t)*actions, u_t = noise - actions (pi0_pytorch.py:326-328)."""
    Bn = actions.shape[0]
    HA = actions.numel() // Bn
    x_t = torch.empty_like(actions)
    u_t = torch.empty_like(actions)
    _lib.call("kai0_flow_mix", noise.data_ptr(), actions.data_ptr(), time.data_ptr(), x_t.data_ptr(), u_t.data_ptr(), Bn,
              HA, _stream())  # fmt: skip
    return x_t, u_t


class MseFn(torch.autograd.Function):
    """F.mse_loss(u, v, reduction="none") with gradient only into v (pi0_pytorch.py:373)."""

    @staticmethod
    def forward(ctx, u, v):
        _chk(u, F32, "mse.u")
        _chk(v, F32, "mse.v")
        loss = torch.empty_like(v)
        _lib.call("kai0_mse_fwd", u.data_ptr(), v.data_ptr(), loss.data_ptr(), v.numel(), _stream())
        ctx.save_for_backward(u, v)
        return loss

    @staticmethod
    def backward(ctx, dl):
        u, v = ctx.saved_tensors
        dl = dl.contiguous()
        dv = torch.empty_like(v)
        _lib.call("kai0_mse_bwd", u.data_ptr(), v.data_ptr(), dl.data_ptr(), dv.data_ptr(), v.numel(), _stream())
        return None, dv


def mse_loss(u, v):
    return MseFn.apply(u, v)


def denoise_glue(x_t, *, xs=None, mod=None, mod_ld=0, rows_per_batch=1, eps=1e-6, w_out=None, b_out=None, dt=0.0, w_in=None,
                 b_in=None, xs_next=None, rowsq_next=None):
    """kai0_denoise_glue: close a denoise step (final adaRMS -> action_out_proj -> Euler update of x_t, in place) and / or open
    the next one (action_in_proj -> bf16 suffix embedding) in one launch.  x_t: f32 [rows, A] contiguous."""
    rows, A = x_t.shape[0], x_t.shape[1]
    D = (xs if xs is not None else xs_next).shape[1]
    for t in (x_t, w_out, b_out, w_in, b_in, mod):
        if t is not None and t.dtype != F32:
            raise TypeError("denoise_glue: x_t, weights, biases and modulations are f32")
    _lib.call("kai0_denoise_glue", _p(xs), _p(mod), mod_ld, rows_per_batch, eps, _p(w_out), _p(b_out), x_t.data_ptr(), dt, _p(w_in),
              _p(b_in), _p(xs_next), rows, D, A, _p(rowsq_next), _stream())


def euler_step_(x, v, dt: float):
    _lib.call("kai0_euler_step", x.data_ptr(), v.data_ptr(), dt, x.numel(), _stream())
    return x


def sqrt_scale(dim: int) -> float:
    """`math.sqrt(dim)` as the f32 scalar torch multiplies a bf16 tensor by."""
    return float(torch.tensor(math.sqrt(dim), dtype=torch.float32))
