"""Parity of every HIP kernel against a plain torch fp32 reference of the same op (run on the GPU box).
Calls go through the C-ABI (kai0_amd.ops -> ctypes -> libkai0hip.so)."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    from kai0_amd import ops as _ops

    return _ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, dtype=BF16, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def assert_close_bf16(out, ref, tol=6e-3, what=""):
    """`ref` is an fp32 result; `out` is bf16: one bf16 rounding of the exact value is <= 2^-8 relative."""
    out, ref = out.float(), ref.float()
    err = (out - ref).abs()
    bound = tol * ref.abs() + tol * ref.abs().mean() + 1e-6
    bad = (err > bound).float().mean().item()
    assert bad < 1e-4, f"{what}: {bad:.2e} of elements out of tolerance; rel-L2 {rel_err(out, ref):.3e}, max abs {err.max().item():.3e}"
    assert rel_err(out, ref) < 4e-3, f"{what}: rel-L2 {rel_err(out, ref):.3e}"


# --------------------------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [
    (128, 128, 64),
    (256, 384, 512),
    (200, 136, 72),      # ragged M, N; K tail (72 = 64 + 8)
    (968, 2048, 2048),   # one pi0.5 prefix sample through q_proj
    (50, 1024, 1024),    # one action-expert chunk
    (1018 * 2, 256, 2048),
    (264, 4304, 1152),   # SigLIP fc1 (N = 4304 = 33.6 tiles)
    (264, 1152, 4304),   # SigLIP fc2 (K tail: 4304 = 67*64 + 16)
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt(ops, M, N, K):
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    out = ops.linear_fwd(x, w)
    ref = x.float() @ w.float().t()
    assert_close_bf16(out, ref, what=f"NT {M}x{N}x{K}")


def test_gemm_nt_is_not_transposed(ops):
    """A = I check with an asymmetric B (guide rule: symmetric inputs hide a row/col swap)."""
    n = 128
    eye = torch.eye(n, dtype=BF16, device=dev())
    w = (torch.arange(n * n, device=dev()).reshape(n, n) % 251).to(BF16) * 0.01
    out = ops.linear_fwd(eye, w)  # I @ w^T = w^T
    assert torch.equal(out, w.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 512, 384), (200, 136, 72), (968, 2048, 2048), (96, 72, 256)])
def test_gemm_nn(ops, M, N, K):
    """dgrad shape: C[M,N] = A[M,K] @ B[K,N], B stored [K][N] (contraction-strided -> ds_read_b64_tr_b16)."""
    a, b = rnd(M, K, seed=3), rnd(K, N, seed=4, scale=0.05)
    out = torch.empty((M, N), dtype=BF16, device=dev())
    ops.gemm(a, b, out, M=M, N=N, K=K, a_kc=True, b_kc=False, lda=K, ldb=N, ldc=N)
    assert_close_bf16(out, a.float() @ b.float(), what=f"NN {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 500), (136, 72, 1000), (2048, 256, 3872), (72, 256, 264)])
def test_gemm_tn(ops, M, N, K):
    """wgrad shape: C[M,N] = A[K,M]^T @ B[K,N], both stored contraction-major."""
    a, b = rnd(K, M, seed=5), rnd(K, N, seed=6, scale=0.05)
    out = torch.empty((M, N), dtype=BF16, device=dev())
    ops.gemm(a, b, out, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N)
    assert_close_bf16(out, a.float().t() @ b.float(), what=f"TN {M}x{N}x{K}")


@pytest.mark.parametrize("lay,M,N,K,S", [("TN", 256, 2048, 30976, 8), ("TN", 1152, 1152, 24576, 6), ("NT", 136, 264, 4104, 3),
                                         ("NN", 200, 136, 1000, 4), ("TN", 72, 256, 1000, 16)])
def test_gemm_split_k(ops, lay, M, N, K, S):
    a_kc, b_kc = lay[0] == "N", lay[1] == "T"
    a = rnd(M, K, seed=1) if a_kc else rnd(K, M, seed=1)
    b = rnd(N, K, seed=2, scale=0.05) if b_kc else rnd(K, N, seed=2, scale=0.05)
    out = torch.empty((M, N), dtype=BF16, device=dev())
    ops.gemm(a, b, out, M=M, N=N, K=K, a_kc=a_kc, b_kc=b_kc, lda=a.shape[1], ldb=b.shape[1], ldc=N, split_k=S)
    A = a.float() if a_kc else a.float().t()
    Bm = b.float().t() if b_kc else b.float()
    assert_close_bf16(out, A @ Bm, what=f"split-K {lay} {M}x{N}x{K}/{S}")
    assert ops.pick_split_k(256, 2048, 30976) > 1 and ops.pick_split_k(16384, 2048, 30976) == 1


def test_gemm_split_k_full_epilogue(ops):
    """Skinny inference shape: M = 50 rows, gate + residual + bias through the split-K reduction, row-remapped output."""
    B, rpb, N, K, S_ld, row0 = 1, 50, 264, 4096, 64, 8
    M = B * rpb
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    gate, res, bias = rnd(B, N, seed=5), rnd(M, N, seed=6), rnd(N, seed=7)
    ref = (x.float() @ w.float().t() + bias.float()).to(BF16).float()
    ref = (ref * gate.float().repeat_interleave(rpb, 0)).to(BF16).float() + res.float()
    for S in (1, 7):
        out = torch.empty((M, N), dtype=BF16, device=dev())
        ops.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, residual=res, ldr=N, gate=gate, gate_rpb=rpb,
                 gate_ld=N, split_k=S)
        assert_close_bf16(out, ref, what=f"split {S} epilogue")
    buf = torch.zeros((B, S_ld, N), dtype=BF16, device=dev())
    ops.gemm(x, w, buf, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, c_map=(rpb, S_ld, row0), split_k=5)
    assert_close_bf16(buf[:, row0 : row0 + rpb].reshape(M, N), x.float() @ w.float().t(), what="split + c_map")


def test_gemm_tn_a_only(ops):
    """C = A[K,M]^T @ B[N,K]^T (A contraction-strided, B K-contiguous)."""
    M, N, K = 136, 200, 328
    a, b = rnd(K, M, seed=7), rnd(N, K, seed=8, scale=0.05)
    out = torch.empty((M, N), dtype=BF16, device=dev())
    ops.gemm(a, b, out, M=M, N=N, K=K, a_kc=False, b_kc=True, lda=M, ldb=K, ldc=N)
    assert_close_bf16(out, a.float().t() @ b.float().t(), what="TN(a only)")


def test_gemm_epilogue_bias_gelu_residual(ops):
    M, N, K = 264, 520, 328
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.08)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    out, pre = ops.linear_fwd(x, w, bias=bias, residual=res, act=1, want_pre=True)
    y = (x.float() @ w.float().t() + bias.float()).to(BF16)
    assert_close_bf16(pre, y.float(), what="pre-activation")
    ref = (torch.nn.functional.gelu(pre.float(), approximate="tanh").to(BF16).float() + res.float())
    assert_close_bf16(out, ref, what="bias+gelu+residual")
    # f32 bias variant
    out2 = ops.linear_fwd(x, w, bias=bias.float())
    assert_close_bf16(out2, x.float() @ w.float().t() + bias.float(), what="f32 bias")


def test_gemm_epilogue_gate_scale(ops):
    B, rpb, N, K = 3, 50, 136, 264
    M = B * rpb
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.08)
    gate, res = rnd(B, N, seed=5), rnd(M, N, seed=6)
    out = ops.linear_fwd(x, w, residual=res, gate=gate, gate_rpb=rpb)
    y = (x.float() @ w.float().t()).to(BF16).float()
    ref = (y * gate.float().repeat_interleave(rpb, 0)).to(BF16).float() + res.float()
    assert_close_bf16(out, ref, what="gate+residual")
    out = torch.empty((M, N), dtype=BF16, device=dev())
    ops.gemm(x, w, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, scale=0.0625)
    assert_close_bf16(out, y * 0.0625, what="scale")


def test_gemm_row_remaps_and_batch(ops):
    """c_map writes flat rows into a padded [B, S_ld, N] buffer; a_map reads them back; batched two-level strides."""
    B, rows, S_ld, row0, N, K = 3, 50, 64, 8, 136, 72
    x, w = rnd(B * rows, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    buf = torch.zeros((B, S_ld, N), dtype=BF16, device=dev())
    ops.gemm(x, w, buf, M=B * rows, N=N, K=K, lda=K, ldb=K, ldc=N, c_map=(rows, S_ld, row0))
    ref = (x.float() @ w.float().t()).view(B, rows, N)
    assert_close_bf16(buf[:, row0 : row0 + rows], ref, what="c_map")
    assert float(buf[:, :row0].abs().max()) == 0.0 and float(buf[:, row0 + rows :].abs().max()) == 0.0
    w2 = rnd(K, N, seed=3, scale=0.1)  # [N2=K, K2=N]
    out = torch.empty((B * rows, K), dtype=BF16, device=dev())
    ops.gemm(buf, w2, out, M=B * rows, N=K, K=N, lda=N, ldb=N, ldc=K, a_map=(rows, S_ld, row0))
    ref2 = buf[:, row0 : row0 + rows].reshape(B * rows, N).float() @ w2.float().t()
    assert_close_bf16(out, ref2, what="a_map")
    # TN with a remapped contraction index
    g = rnd(B * rows, K, seed=4)
    dw = torch.empty((N, K), dtype=BF16, device=dev())
    ops.gemm(buf, g, dw, M=N, N=K, K=B * rows, a_kc=False, b_kc=False, lda=N, ldb=K, ldc=K, a_map=(rows, S_ld, row0))
    ref3 = buf[:, row0 : row0 + rows].reshape(B * rows, N).float().t() @ g.float()
    assert_close_bf16(dw, ref3, what="a_map on contraction rows")
    # batched heads: q [n, S, NH*HD] x k -> scores [n*NH, S, S]
    n, S, NH, HD = 2, 16, 4, 72
    E = NH * HD
    q, k = rnd(n * S, E, seed=5), rnd(n * S, E, seed=6)
    sc = torch.empty((n * NH, S, S), dtype=BF16, device=dev())
    ops.gemm(q, k, sc, M=S, N=S, K=HD, lda=E, ldb=E, ldc=S, batch=n * NH, batch_inner=NH, sA=(S * E, HD), sB=(S * E, HD),
             sC=(NH * S * S, S * S))
    qh = q.view(n, S, NH, HD).permute(0, 2, 1, 3).float()
    kh = k.view(n, S, NH, HD).permute(0, 2, 1, 3).float()
    assert_close_bf16(sc.view(n, NH, S, S), qh @ kh.transpose(-1, -2), what="two-level batch")


def test_gemm_batch_strides_may_be_negative_or_span_two_allocations(ops):
    """kai0hip.h: the batch strides are signed 64-bit element counts.  The joint-attention backward uses `&Q - &dO` — the distance
    between two separately allocated tensors, of either sign — as the outer B stride of its merged dV | dK launch."""
    from kai0_amd._lib import Kai0HipError

    Bn, M, N, K = 2, 64, 72, 136
    a = rnd(2 * Bn * K, M, seed=1)  # A stored [K][M] per entry (TN), entries (j, b) contiguous
    pool = rnd(4 * Bn * K * N + 64, seed=2)  # two "allocations" cut from one pool: the test controls their order
    lo, hi = pool[: Bn * K * N].view(Bn, K, N), pool[3 * Bn * K * N : 4 * Bn * K * N].view(Bn, K, N)
    for b0, b1 in ((lo, hi), (hi, lo)):  # positive, then negative outer stride
        out = torch.empty((2, Bn, M, N), dtype=BF16, device=dev())
        dist = (b1.data_ptr() - b0.data_ptr()) // 2
        ops.gemm(a, b0, out, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N, batch=2 * Bn, batch_inner=Bn,
                 sA=(Bn * K * M, K * M), sB=(dist, K * N), sC=(Bn * M * N, M * N))
        av = a.view(2, Bn, K, M).float()
        for j, bt in enumerate((b0, b1)):
            assert_close_bf16(out[j], av[j].transpose(-1, -2) @ bt.float(), what=f"outer stride {dist:+d}, entry {j}")
    with pytest.raises(Kai0HipError):  # a stride that would break the 16-B alignment of an entry is refused
        ops.gemm(a, lo, out, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N, batch=2 * Bn, batch_inner=Bn,
                 sA=(Bn * K * M, K * M), sB=(dist + 4, K * N), sC=(Bn * M * N, M * N))


@pytest.mark.parametrize("w8", [1, 2])  # the four-wave and the eight-wave 128 x 128 tile (kai0_gemm_desc.small_w8)
@pytest.mark.parametrize("B,P,S_ld,T", [(1, 968, 1024, 200), (2, 136, 160, 40)])
def test_gemm_rope_epilogue_is_bit_identical_to_gemm_then_rope(ops, B, P, S_ld, T, w8):
    """act 7 (kai0hip.h): the stacked q | k | v projection of the B = 1 prefix pass with the rotation in its epilogue — weight rows
    permuted so that a rotation's partners meet in one 128-column tile, output routed to the padded q buffer and the K / V caches at
    the REAL columns — against the same GEMM followed by kai0_rope_inplace2: bit for bit, padding rows untouched."""
    H, HD, D = 8, 256, 264
    NQ, M = H * HD, B * P
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(M, D, generator=g) * 1.0).to(BF16).to(dev())
    w = (torch.randn(NQ + 2 * HD, D, generator=g) * 0.06).to(BF16).to(dev())
    pos = torch.stack([torch.cumsum((torch.rand(P, generator=g) > 0.1).int(), 0) - 1 for _ in range(B)]).clamp(min=0).to(torch.int32).to(dev())
    inv_freq = (1.0 / (10000 ** (torch.arange(0, HD, 2).float() / HD))).to(BF16).float().to(dev())

    def bufs():
        return (torch.full((B, S_ld, NQ), 3.0, dtype=BF16, device=dev()), torch.full((B, S_ld, HD), 3.0, dtype=BF16, device=dev()),
                torch.full((B, S_ld, HD), 3.0, dtype=BF16, device=dev()))

    q0, k0, v0 = bufs()
    segs = lambda q, k, v: [(q, NQ, 0), (k, HD, NQ), (v, HD, NQ + HD)]  # noqa: E731
    ops.gemm(x, w, q0, M=M, N=NQ + 2 * HD, K=D, lda=D, ldb=D, ldc=NQ, c_map=(P, S_ld, 0), segs=segs(q0, k0, v0))
    ops.rope2_(q0, H, k0, 1, pos, inv_freq, B, P, S_ld, 0, HD)
    q1, k1, v1 = bufs()
    perm = ops.rope_permutation(NQ + 2 * HD, NQ + HD).to(dev())
    cos, sin = ops.rope_table(pos.reshape(-1), inv_freq)
    with ops.gemm_tuning(small_w8=w8):
        ops.gemm(x, w[perm].contiguous(), q1, M=M, N=NQ + 2 * HD, K=D, lda=D, ldb=D, ldc=NQ, c_map=(P, S_ld, 0), segs=segs(q1, k1, v1), act=7,
                 rope=(cos.to(BF16), sin.to(BF16), HD // 2, NQ + HD))
    assert torch.equal(q1, q0) and torch.equal(k1, k0) and torch.equal(v1, v0)
    assert float(q1[:, P:].float().min()) == 3.0 and float(k1[:, P:].float().min()) == 3.0  # padding rows untouched
    ref = (x.float() @ w.float().t()).to(BF16).float().view(B, P, -1)
    assert_close_bf16(v1[:, :P], ref[..., NQ + HD :], what="v segment")


def test_gemm_n_not_multiple_of_8(ops):
    M, N, K, ld = 64, 20, 72, 24
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    out = torch.full((M, ld), 7.0, dtype=BF16, device=dev())
    ops.gemm(a, b, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=ld)
    assert_close_bf16(out[:, :N], a.float() @ b.float().t(), what="N=20")
    assert float(out[:, N:].abs().max()) == 0.0


def test_gemm_rejects_bad_arguments(ops):
    from kai0_amd._lib import Kai0HipError

    a, b = rnd(64, 72), rnd(64, 72)
    out = torch.empty((64, 64), dtype=BF16, device=dev())
    with pytest.raises(Kai0HipError):
        ops.gemm(a, b, out, M=64, N=64, K=72, lda=70, ldb=72, ldc=64)  # lda not a multiple of 8
    with pytest.raises(Kai0HipError):
        ops.linear_fwd(a.cpu(), b.cpu())  # CPU tensors: no fallback


@pytest.mark.parametrize("M,N,K", [(32, 3072, 1024), (1600, 32, 1024), (100, 1152, 588), (5, 7, 3)])
def test_gemm_f32(ops, M, N, K):
    x, w, b = rnd(M, K, dtype=F32, seed=1), rnd(N, K, dtype=F32, seed=2, scale=0.05), rnd(N, dtype=F32, seed=3)
    out = torch.empty((M, N), dtype=F32, device=dev())
    ops.gemm_f32(x, K, 1, w, 1, K, out, M, N, K, bias=b)
    ref = x.double() @ w.double().t() + b.double()
    assert rel_err(out, ref) < 2e-6
    # strided: out2 = x^T-style wgrad  dw[n,k] = sum_m dy[m,n] x[m,k]
    dy = rnd(M, N, dtype=F32, seed=4)
    dw = torch.empty((N, K), dtype=F32, device=dev())
    ops.gemm_f32(dy, 1, N, x, K, 1, dw, N, K, M)
    assert rel_err(dw, dy.double().t() @ x.double()) < 2e-6
    if K >= 512:
        out2 = torch.empty((M, N), dtype=F32, device=dev())
        ops.gemm_f32(x, K, 1, w, 1, K, out2, M, N, K, bias=b, split_k=5)
        assert rel_err(out2, ref) < 2e-6


# --------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,D", [(7, 64), (968, 2048), (50, 1024), (300, 1152)])
def test_rmsnorm_fwd_bwd(ops, rows, D):
    x = rnd(rows, D, seed=1).requires_grad_(True)
    w = rnd(D, dtype=F32, seed=2, scale=0.3).requires_grad_(True)
    dy = rnd(rows, D, seed=3)
    y = ops.rmsnorm(x, w, 1e-6)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    var = xr.pow(2).mean(-1, keepdim=True)
    yr = xr * torch.rsqrt(var + 1e-6) * (1.0 + wr)
    yr.backward(dy.float())
    assert_close_bf16(y, yr, what="rmsnorm y")
    assert_close_bf16(x.grad, xr.grad, what="rmsnorm dx", tol=1e-2)
    assert rel_err(w.grad, wr.grad) < 2e-3


@pytest.mark.parametrize("B,rpb,D", [(3, 50, 1024), (2, 5, 64)])
def test_adarms_fwd_bwd(ops, B, rpb, D):
    rows = B * rpb
    x = rnd(rows, D, seed=1).requires_grad_(True)
    mod = rnd(B, 3 * D, dtype=F32, seed=2, scale=0.3).requires_grad_(True)
    dy, dgate = rnd(rows, D, seed=3), rnd(B, D, seed=4)
    y, gate = ops.adarms(x, mod, rpb, 1e-6)
    torch.autograd.backward([y, gate], [dy, dgate])
    xr = x.detach().float().requires_grad_(True)
    mr = mod.detach().clone().requires_grad_(True)
    scale, shift, gt = mr.view(B, 1, 3 * D).chunk(3, dim=-1)
    x3 = xr.view(B, rpb, D)
    var = x3.pow(2).mean(-1, keepdim=True)
    yr = x3 * torch.rsqrt(var + 1e-6) * (1 + scale) + shift
    torch.autograd.backward([yr, gt.squeeze(1)], [dy.float().view(B, rpb, D), dgate.float()])
    assert_close_bf16(y, yr.reshape(rows, D), what="adarms y")
    assert_close_bf16(gate, gt.squeeze(1), what="adarms gate")
    assert_close_bf16(x.grad, xr.grad, what="adarms dx", tol=1e-2)
    assert rel_err(mod.grad, mr.grad) < 2e-3


@pytest.mark.parametrize("rows,D", [(512, 1152), (9, 64)])
def test_layernorm_fwd_bwd(ops, rows, D):
    x = rnd(rows, D, seed=1).requires_grad_(True)
    w = (1 + rnd(D, seed=2, scale=0.2).float()).to(BF16).requires_grad_(True)
    b = rnd(D, seed=3, scale=0.2).requires_grad_(True)
    dy = rnd(rows, D, seed=4)
    y = ops.layernorm(x, w, b, 1e-6)
    y.backward(dy)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-6)
    yr.backward(dy.float())
    assert_close_bf16(y, yr, what="layernorm y")
    assert_close_bf16(x.grad, xr.grad, what="layernorm dx", tol=1e-2)
    assert rel_err(w.grad, wr.grad) < 1e-2 and rel_err(b.grad, br.grad) < 1e-2


def test_queued_final_reductions_equal_the_immediate_ones(ops):
    """Norm-weight / bias gradients whose destination is a trainer's flat buffer (`_kai0_grad_out`) are summed by the batched
    launch at the end of the backward pass: the same bits as the per-gradient launches, also past 32 queued items."""
    rows, D = 700, 1152

    def run(flat):
        torch.manual_seed(0)
        x = rnd(rows, D, seed=1).requires_grad_(True)
        lns = [((1 + rnd(D, seed=10 + i, scale=0.2).float()).to(BF16).requires_grad_(True), rnd(D, seed=40 + i, scale=0.2).requires_grad_(True))
               for i in range(14)]  # 28 LayerNorm items
        rw = [rnd(D, dtype=F32, seed=70 + i, scale=0.3).requires_grad_(True) for i in range(4)]
        ws = [rnd(n, D, seed=80 + i, scale=0.1).requires_grad_(True) for i, n in enumerate((136, 72, 72))]
        bs = [rnd(n, seed=90 + i).requires_grad_(True) for i, n in enumerate((136, 72, 72))]
        params = [t for pair in lns for t in pair] + rw + bs
        arrived = []
        if flat:
            for p_ in params:
                p_._kai0_grad_out = torch.full_like(p_, float("nan"))
                p_._kai0_grad_done = lambda p_=p_: arrived.append(p_)
        h = x
        for w, b in lns:
            h = ops.layernorm(h, w, b, 1e-6)
        for w in rw:
            h = ops.rmsnorm(h, w, 1e-6)
        q, k, v = ops.linear_multi(h, ws, bs)
        torch.autograd.backward([q, k, v], [rnd(rows, 136, seed=11), rnd(rows, 72, seed=12), rnd(rows, 72, seed=13)])
        if flat:
            assert len(arrived) == len(params) and all(p_.grad is None for p_ in params)
            return [p_._kai0_grad_out for p_ in params] + [x.grad]
        return [p_.grad for p_ in params] + [x.grad]

    ref, got = run(False), run(True)
    assert not ops._DEFERRED.get(torch.cuda.current_device())
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_norm_residual_passthrough_and_linear_multi(ops):
    """x feeds a norm AND a residual: one backward kernel returns dres + dnorm; q/k/v dgrads accumulate in the epilogue."""
    rows, D = 300, 136
    x = rnd(rows, D, seed=1).requires_grad_(True)
    w = rnd(D, dtype=F32, seed=2, scale=0.3).requires_grad_(True)
    ws = [rnd(n, D, seed=3 + i, scale=0.1).requires_grad_(True) for i, n in enumerate((136, 72, 72))]
    bs = [rnd(n, seed=7 + i).requires_grad_(True) for i, n in enumerate((136, 72, 72))]
    xr_, h = ops.rmsnorm_res(x, w, 1e-6)
    q, k, v = ops.linear_multi(h, ws, bs)
    out = xr_ + q  # residual branch + a consumer
    douts = [rnd(rows, 136, seed=11), rnd(rows, 72, seed=12), rnd(rows, 72, seed=13)]
    torch.autograd.backward([out, k, v], douts)
    X = x.detach().float().requires_grad_(True)
    W = w.detach().clone().requires_grad_(True)
    Ws = [t.detach().float().requires_grad_(True) for t in ws]
    Bs = [t.detach().float().requires_grad_(True) for t in bs]
    H = X * torch.rsqrt(X.pow(2).mean(-1, keepdim=True) + 1e-6) * (1 + W)
    Q, K_, V = (H @ Ws[i].t() + Bs[i] for i in range(3))
    torch.autograd.backward([X + Q, K_, V], [d.float() for d in douts])
    assert_close_bf16(x.grad, X.grad, what="dres + dnorm", tol=1.5e-2)
    assert rel_err(w.grad, W.grad) < 1e-2
    for a, r in zip(ws + bs, Ws + Bs):
        assert rel_err(a.grad, r.grad) < 1.5e-2


def test_linear_autograd(ops):
    M, N, K = 264, 328, 200
    x = rnd(M, K, seed=1).requires_grad_(True)
    w = rnd(N, K, seed=2, scale=0.08).requires_grad_(True)
    bias = rnd(N, seed=3).requires_grad_(True)
    res = rnd(M, N, seed=4).requires_grad_(True)
    dy = rnd(M, N, seed=5)
    out = ops.linear(x, w, bias, res, 1)
    out.backward(dy)
    xr, wr, br, rr = (t.detach().float().requires_grad_(True) for t in (x, w, bias, res))
    ref = torch.nn.functional.gelu(xr @ wr.t() + br, approximate="tanh") + rr
    ref.backward(dy.float())
    assert_close_bf16(out, ref, what="linear out", tol=1e-2)
    assert_close_bf16(x.grad, xr.grad, what="linear dx", tol=1.5e-2)
    assert_close_bf16(w.grad, wr.grad, what="linear dw", tol=1.5e-2)
    assert rel_err(bias.grad, br.grad) < 1e-2
    assert torch.equal(res.grad, dy)


def test_linear_rows_f32(ops):
    for M, N, K in [(10, 3072, 1024), (1, 1024, 1024), (16, 37, 64)]:
        x, w, b = rnd(M, K, dtype=F32, seed=1), rnd(N, K, dtype=F32, seed=2, scale=0.05), rnd(N, dtype=F32, seed=3)
        out = ops.linear_f32(x, w, b)
        assert rel_err(out, x.double() @ w.double().t() + b.double()) < 2e-6


def test_linear_f32_autograd(ops):
    M, N, K = 37, 96, 200
    x = rnd(M, K, dtype=F32, seed=1).requires_grad_(True)
    w = rnd(N, K, dtype=F32, seed=2, scale=0.1).requires_grad_(True)
    b = rnd(N, dtype=F32, seed=3).requires_grad_(True)
    dy = rnd(M, N, dtype=F32, seed=4)
    ops.silu_f32(ops.linear_f32(x, w, b)).backward(dy)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.silu(xr @ wr.t() + br).backward(dy.double())
    for a, r in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        assert rel_err(a, r) < 1e-5


# ----------------------------------------------------------------------------------------------- elementwise
def test_geglu_and_gated_residual(ops):
    g = rnd(200, 136, seed=1).requires_grad_(True)
    u = rnd(200, 136, seed=2).requires_grad_(True)
    dh = rnd(200, 136, seed=3)
    h = ops.geglu(g, u)
    h.backward(dh)
    gr, ur = (t.detach().float().requires_grad_(True) for t in (g, u))
    hr = torch.nn.functional.gelu(gr, approximate="tanh") * ur
    hr.backward(dh.float())
    assert_close_bf16(h, hr, what="geglu", tol=1e-2)
    assert_close_bf16(g.grad, gr.grad, what="geglu dg", tol=1.5e-2)
    assert_close_bf16(u.grad, ur.grad, what="geglu du", tol=1e-2)
    B, rpb, D = 3, 50, 136
    x = rnd(B * rpb, D, seed=4).requires_grad_(True)
    y = rnd(B * rpb, D, seed=5).requires_grad_(True)
    gate = rnd(B, D, seed=6).requires_grad_(True)
    do = rnd(B * rpb, D, seed=7)
    o = ops.gated_residual(x, y, gate, rpb)
    o.backward(do)
    xr, yr, gtr = (t.detach().float().requires_grad_(True) for t in (x, y, gate))
    orf = xr + yr * gtr.repeat_interleave(rpb, 0)
    orf.backward(do.float())
    assert_close_bf16(o, orf, what="gated", tol=1e-2)
    assert torch.equal(x.grad, do)
    assert_close_bf16(y.grad, yr.grad, what="gated dy", tol=1e-2)
    assert_close_bf16(gate.grad, gtr.grad, what="gated dgate", tol=1e-2)


@pytest.mark.parametrize("M,D,Fd", [(200, 64, 136), (4352, 256, 520), (300, 64, 128)])
def test_geglu_mlp_fused(ops, M, D, Fd):
    x = rnd(M, D, seed=1).requires_grad_(True)
    wg, wu = (rnd(Fd, D, seed=s, scale=0.1).requires_grad_(True) for s in (2, 3))
    wd = rnd(D, Fd, seed=4, scale=0.1).requires_grad_(True)
    res = rnd(M, D, seed=5).requires_grad_(True)
    dy = rnd(M, D, seed=6)
    out = ops.geglu_mlp(x, wg, wu, wd, res)
    out.backward(dy)
    ref_in = [t.detach().float().requires_grad_(True) for t in (x, wg, wu, wd, res)]
    xr, gr, ur, dr, rr = ref_in
    ref = (torch.nn.functional.gelu(xr @ gr.t(), approximate="tanh") * (xr @ ur.t())) @ dr.t() + rr
    ref.backward(dy.float())
    assert rel_err(out, ref) < 6e-3
    for n, t, r in zip(("dx", "dwg", "dwu", "dwd"), (x, wg, wu, wd), ref_in):
        assert rel_err(t.grad, r.grad) < 1.5e-2, f"{n}: {rel_err(t.grad, r.grad):.3e}"
    assert torch.equal(res.grad, dy)


@pytest.mark.parametrize("M,D,Fd", [(200, 64, 128), (4352, 256, 512), (968, 2048, 16384), (4100, 2048, 16384), (50, 1024, 4096)])
def test_gemm_geglu_pair_equals_gate_gemm_plus_up_gemm(ops, M, D, Fd):
    """act 6 (gate | up as one GEMM over two weights, GeGLU in registers) against the two-launch form it replaces (gate GEMM, then
    up GEMM with the act-2 epilogue): h, g and u bit for bit — in the 128x128 and the 256x256 quadrant configurations, ragged M,
    with and without the pre-activation outputs; and against an fp32 reference of the op."""
    x = rnd(M, D, seed=1)
    wg, wu = rnd(Fd, D, seed=2, scale=0.05), rnd(Fd, D, seed=3, scale=0.05)
    g_ref = torch.empty((M, Fd), dtype=BF16, device=dev())
    ops.gemm(x, wg, g_ref, M=M, N=Fd, K=D, lda=D, ldb=D, ldc=Fd)  # (no split-K: the same summation order as the pair GEMM)
    u_ref, h_ref = torch.empty_like(g_ref), torch.empty_like(g_ref)
    ops.gemm(x, wu, h_ref, M=M, N=Fd, K=D, lda=D, ldb=D, ldc=Fd, act=2, pre_out=u_ref, aux1=g_ref)
    g, u, h = (torch.full_like(g_ref, float("nan")) for _ in range(3))
    ops.gemm(x, wg, h, M=M, N=Fd, K=D, lda=D, ldb=D, ldc=Fd, act=6, B2=wu, pre_out=g, pre_out2=u)
    assert torch.equal(g, g_ref) and torch.equal(u, u_ref) and torch.equal(h, h_ref)
    h2 = torch.full_like(h_ref, float("nan"))
    ops.gemm(x, wg, h2, M=M, N=Fd, K=D, lda=D, ldb=D, ldc=Fd, act=6, B2=wu)  # inference: only h
    assert torch.equal(h2, h_ref)
    xf = x.float()
    ref = torch.nn.functional.gelu((xf @ wg.float().t()).bfloat16().float(), approximate="tanh") * (xf @ wu.float().t()).bfloat16().float()
    assert_close_bf16(h, ref, tol=1.2e-2, what="geglu pair h")
    with pytest.raises(Exception, match="act=6"):
        ops.gemm(x, wg, h2, M=M, N=Fd, K=D, lda=D, ldb=D, ldc=Fd, act=6)  # no B2


@pytest.mark.parametrize("M,D,Fd", [(200, 64, 136), (4352, 256, 520), (4100, 1152, 4304)])
def test_gelu_mlp_fused(ops, M, D, Fd):
    """SigLIP MLP as one node: padded [M, F] intermediates, GELU backward in the dgrad epilogue (act 5), bias gradients over
    the padded rows; both the small-M (NN dgrad) and the many-rows (transposed-weight NT dgrad, fast epilogue) paths."""
    x = rnd(M, D, seed=1).requires_grad_(True)
    w1 = rnd(Fd, D, seed=2, scale=0.05).requires_grad_(True)
    b1 = rnd(Fd, seed=3, scale=0.1).requires_grad_(True)
    w2 = rnd(D, Fd, seed=4, scale=0.05).requires_grad_(True)
    b2 = rnd(D, seed=5, scale=0.1).requires_grad_(True)
    res = rnd(M, D, seed=6).requires_grad_(True)
    dy = rnd(M, D, seed=7)
    out = ops.gelu_mlp(x, w1, b1, w2, b2, res)
    out.backward(dy)
    ref_in = [t.detach().float().requires_grad_(True) for t in (x, w1, b1, w2, b2, res)]
    xr, w1r, b1r, w2r, b2r, rr = ref_in
    ref = torch.nn.functional.gelu(xr @ w1r.t() + b1r, approximate="tanh") @ w2r.t() + b2r + rr
    ref.backward(dy.float())
    assert rel_err(out, ref) < 6e-3
    for n, t, r in zip(("dx", "dw1", "db1", "dw2", "db2"), (x, w1, b1, w2, b2), ref_in):
        assert rel_err(t.grad, r.grad) < 1.5e-2, f"{n}: {rel_err(t.grad, r.grad):.3e}"
    assert torch.equal(res.grad, dy)


def _rope_ref(x, pos, inv_freq, inverse=False):
    """modeling_gemma.py:149-194 in bf16: x [B, S, H, HD], pos [B, S]."""
    freqs = pos[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos().to(BF16)[:, :, None, :], emb.sin().to(BF16)[:, :, None, :]
    if inverse:
        sin = -sin
    half = x.shape[-1] // 2
    rot = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return (x * cos) + (rot * sin)


@pytest.mark.parametrize("H,HD", [(8, 256), (1, 256), (8, 16)])
def test_rope(ops, H, HD):
    B, S, S_ld, row0 = 2, 50, 64, 8
    x = rnd(B, S_ld, H * HD, seed=1)
    pos = torch.randint(0, 1100, (B, S), device=dev(), dtype=torch.int32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.int64).float() / HD)).to(dev())
    ref = x.clone()
    ref[:, row0 : row0 + S] = _rope_ref(x[:, row0 : row0 + S].reshape(B, S, H, HD), pos, inv).reshape(B, S, H * HD)
    y = x.clone()
    ops.rope_(y, pos, inv, B, S, S_ld, row0, H, HD)
    # cos/sin of a large angle may differ by one bf16 ulp between libms: allow a handful of 1-ulp flips
    mism = (y != ref).float().mean().item()
    assert mism < 2e-2, f"rope mismatch fraction {mism}"
    assert rel_err(y, ref) < 3e-3
    assert torch.equal(y[:, :row0], x[:, :row0]) and torch.equal(y[:, row0 + S :], x[:, row0 + S :])


def test_softmax_mask(ops):
    B, Sq, H, Sk, ld = 2, 24, 8, 40, 48
    M = Sq * H
    scores = rnd(B, M, ld, seed=1, scale=3.0)
    pad = torch.ones((B, Sk), dtype=torch.bool, device=dev())
    pad[0, 10:14] = False
    pad[1, 30:] = False
    att = torch.zeros((B, Sk), dtype=torch.bool, device=dev())
    att[:, 32] = True  # last 8 tokens form the "suffix"
    from kai0_amd.model import build_mask_codes

    qcode, kcode, _ = build_mask_codes(pad, att)
    q0 = Sk - Sq
    probs = torch.empty_like(scores)
    from kai0_amd import _lib

    _lib.call("kai0_softmax_mask_fwd", scores.data_ptr(), probs.data_ptr(), qcode.data_ptr(), kcode.data_ptr(), B, Sq, H,
              Sk, ld, M * ld, q0, qcode.stride(0), kcode.stride(0), ops._stream())
    cum = torch.cumsum(att.int(), 1)
    m2d = (cum[:, None, :] <= cum[:, :, None]) & (pad[:, None, :] & pad[:, :, None])  # [B, Sk, Sk]
    m2d = m2d[:, q0:, :]  # query rows
    s = scores[:, :, :Sk].float().view(B, Sq, H, Sk)
    s = s + torch.where(m2d[:, :, None, :], 0.0, -2.3819763e38)
    ref = torch.softmax(s, dim=-1)
    valid_q = pad[:, q0:]
    got = probs[:, :, :Sk].float().view(B, Sq, H, Sk)
    assert_close_bf16(got[valid_q], ref[valid_q], what="softmax", tol=1e-2)
    assert float(probs[:, :, Sk:].abs().max()) == 0.0
    assert float(got[~valid_q].abs().max()) == 0.0  # fully masked (padded) query rows are defined as zeros
    # backward
    dp = rnd(B, M, ld, seed=2)
    ds = torch.empty_like(dp)
    _lib.call("kai0_softmax_bwd", probs.data_ptr(), dp.data_ptr(), 0, ds.data_ptr(), B * M, Sk, ld, 0.5, ops._stream())
    ds32 = torch.empty_like(dp)
    _lib.call("kai0_softmax_bwd", probs.data_ptr(), dp.float().contiguous().data_ptr(), 1, ds32.data_ptr(), B * M, Sk, ld,
              0.5, ops._stream())
    assert torch.equal(ds, ds32)
    p = probs.float()[:, :, :Sk]
    d = dp.float()[:, :, :Sk]
    refd = p * (d - (p * d).sum(-1, keepdim=True)) * 0.5
    assert_close_bf16(ds[:, :, :Sk], refd, what="softmax bwd", tol=1.5e-2)


def _mqa_ref(q, k, v, pos, inv, qcode, kcode, H, HD):
    """fp32 reference of RoPE + prefix-LM masked MQA: q [B,S,H*HD], k/v [B,S,HD]."""
    B, S, _ = q.shape
    freqs = pos[:, :, None].float() * inv[None, None, :]
    emb = torch.cat((freqs, freqs), -1)
    cos, sin = emb.cos()[:, :, None, :], emb.sin()[:, :, None, :]

    def rope(x):
        half = x.shape[-1] // 2
        return x * cos + torch.cat((-x[..., half:], x[..., :half]), -1) * sin

    qh = rope(q.view(B, S, H, HD)).permute(0, 2, 1, 3)
    kh = rope(k.view(B, S, 1, HD)).permute(0, 2, 1, 3)
    vh = v.view(B, S, 1, HD).permute(0, 2, 1, 3)
    sc = (qh @ kh.transpose(-1, -2)) * HD**-0.5
    allowed = kcode[:, None, None, :] <= qcode[:, None, :, None]
    sc = sc.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(sc, -1)
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, S, H * HD)


@pytest.mark.parametrize("H,HD,P,Hs", [(8, 16, 20, 6), (8, 256, 40, 10)])
def test_joint_attention_fwd_bwd(ops, H, HD, P, Hs):
    from kai0_amd.model import build_mask_codes

    B, S = 2, P + Hs
    pad = torch.ones((B, S), dtype=torch.bool, device=dev())
    pad[0, 3:6] = False
    pad[1, P - 4 : P] = False
    att = torch.zeros((B, S), dtype=torch.bool, device=dev())
    att[:, P] = True
    qcode, kcode, pos = build_mask_codes(pad, att)
    inv = (1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.int64).float() / HD))).to(dev())
    segs = [(P, 1), (Hs, 2)]
    flat = []
    for L, sd in segs:
        flat += [rnd(B * L, H * HD, seed=sd).requires_grad_(True), rnd(B * L, HD, seed=sd + 10).requires_grad_(True),
                 rnd(B * L, HD, seed=sd + 20).requires_grad_(True)]
    outs = ops.joint_attention(pos, qcode, kcode, inv, H, HD, (P, Hs), flat)
    douts = [rnd(B * P, H * HD, seed=30), rnd(B * Hs, H * HD, seed=31)]
    valid = pad.view(-1)
    # gradients arriving at padded query rows are don't-care in the model; zero them for a clean comparison
    dfull = torch.cat([douts[0].view(B, P, -1), douts[1].view(B, Hs, -1)], 1) * pad[:, :, None]
    douts = [dfull[:, :P].reshape(B * P, -1).contiguous(), dfull[:, P:].reshape(B * Hs, -1).contiguous()]
    torch.autograd.backward(list(outs), douts)
    refs = [t.detach().float().requires_grad_(True) for t in flat]
    q = torch.cat([refs[0].view(B, P, -1), refs[3].view(B, Hs, -1)], 1)
    k = torch.cat([refs[1].view(B, P, -1), refs[4].view(B, Hs, -1)], 1)
    v = torch.cat([refs[2].view(B, P, -1), refs[5].view(B, Hs, -1)], 1)
    o = _mqa_ref(q, k, v, pos, inv, qcode, kcode, H, HD)
    o.backward(dfull.float())
    got = torch.cat([outs[0].view(B, P, -1), outs[1].view(B, Hs, -1)], 1)
    # fp32 reference vs a path that (like the reference model) rounds logits and probabilities to bf16
    assert rel_err(got[pad], o[pad]) < 1e-2
    names = ["dq_p", "dk_p", "dv_p", "dq_s", "dk_s", "dv_s"]
    for n, t, r in zip(names, flat, refs):
        e = rel_err(t.grad, r.grad)
        assert e < 2e-2, f"{n}: rel-L2 {e:.3e}"


def test_siglip_attention_fwd_bwd(ops):
    n, S, NH, HD = 3, 16, 4, 72
    E = NH * HD
    q, k, v = (rnd(n * S, E, seed=s).requires_grad_(True) for s in (1, 2, 3))
    do = rnd(n * S, E, seed=4)
    out = ops.siglip_attention(q, k, v, n, S, NH, HD)
    out.backward(do)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    sh = lambda t: t.view(n, S, NH, HD).permute(0, 2, 1, 3)  # noqa: E731
    p = torch.softmax(sh(qr) @ sh(kr).transpose(-1, -2) * HD**-0.5, -1)
    o = (p @ sh(vr)).permute(0, 2, 1, 3).reshape(n * S, E)
    o.backward(do.float())
    assert rel_err(out, o) < 1e-2
    for nme, t, r in (("dq", q, qr), ("dk", k, kr), ("dv", v, vr)):
        e = rel_err(t.grad, r.grad)
        assert e < 2e-2, f"{nme}: rel-L2 {e:.3e}"


def test_embed_and_grad(ops):
    V, D, B, T = 300, 136, 3, 20
    table = rnd(V, D, seed=1).requires_grad_(True)
    tok = torch.randint(0, V, (B, T), device=dev())
    tok[0, :5] = 7  # repeated ids
    scale = ops.sqrt_scale(D)
    out = ops.embed(table, tok, scale)
    ref = (table.detach()[tok.view(-1)].float() * scale).to(BF16)
    assert torch.equal(out, ref)
    do = rnd(B * T, D, seed=2)
    out.backward(do)
    refg = torch.zeros(V, D, device=dev())
    refg.index_add_(0, tok.view(-1), (do.float() * scale).to(BF16).float())
    assert_close_bf16(table.grad, refg, what="embed grad")


def test_patch_embed(ops):
    n, HW, P, D = 3, 56, 14, 72
    img = rnd(n, 3, HW, HW, dtype=F32, seed=1)
    w = rnd(D, 3, P, P, dtype=F32, seed=2, scale=0.05).requires_grad_(True)
    b = rnd(D, dtype=F32, seed=3).requires_grad_(True)
    pos = rnd((HW // P) ** 2, D, dtype=F32, seed=4).requires_grad_(True)
    out = ops.patch_embed(img, w, b, pos, P)
    do = rnd(*out.shape, seed=5)
    out.backward(do)
    wr, br, pr = (t.detach().clone().requires_grad_(True) for t in (w, b, pos))
    ref = torch.nn.functional.conv2d(img, wr, br, stride=P).flatten(2).transpose(1, 2) + pr[None]
    ref.reshape(-1, D).backward(do.float())
    assert_close_bf16(out, ref.reshape(-1, D), what="patch embed")
    for a, r in ((w.grad, wr.grad), (b.grad, br.grad), (pos.grad, pr.grad)):
        assert rel_err(a, r) < 1e-4


def test_time_sincos_and_flow(ops):
    from kai0_amd import _lib

    B, dim = 5, 1024
    t = torch.rand(B, device=dev()) * 0.999 + 0.001
    out = torch.empty((B, dim), dtype=F32, device=dev())
    _lib.call("kai0_time_sincos", t.data_ptr(), out.data_ptr(), B, dim, 4e-3, 4.0, ops._stream())
    frac = torch.linspace(0.0, 1.0, dim // 2, dtype=torch.float64, device=dev())
    period = 4e-3 * (4.0 / 4e-3) ** frac
    x = (1.0 / period * 2 * math.pi)[None, :] * t.double()[:, None]
    ref = torch.cat([torch.sin(x), torch.cos(x)], dim=1).float()
    assert float((out - ref).abs().max()) < 1e-6
    a, nz = rnd(B, 50, 32, dtype=F32, seed=1), rnd(B, 50, 32, dtype=F32, seed=2)
    xt, ut = ops.flow_mix(nz, a, t)
    assert torch.allclose(xt, t[:, None, None] * nz + (1 - t[:, None, None]) * a, atol=1e-6)
    assert torch.equal(ut, nz - a)
    v = rnd(B * 50, 32, dtype=F32, seed=3).requires_grad_(True)
    loss = ops.mse_loss(ut.view(B * 50, 32), v)
    loss.mean().backward()
    vr = v.detach().clone().requires_grad_(True)
    torch.nn.functional.mse_loss(ut.view(B * 50, 32), vr, reduction="none").mean().backward()
    assert torch.allclose(v.grad, vr.grad, rtol=1e-5, atol=1e-8)
    xx = xt.clone()
    ops.euler_step_(xx, ut, -0.1)
    assert torch.allclose(xx, xt - 0.1 * ut, atol=1e-6)


def test_transpose(ops):
    for R, Cc in [(64, 64), (2048, 16384), (4304, 1152), (72, 200)]:
        x = rnd(R, Cc, seed=R)
        assert torch.equal(ops.transpose(x), x.t().contiguous())


def test_linear_autograd_many_rows_uses_transposed_dgrad(ops):
    M, N, K = 4352, 264, 200
    x = rnd(M, K, seed=1).requires_grad_(True)
    w = rnd(N, K, seed=2, scale=0.08).requires_grad_(True)
    dy = rnd(M, N, seed=5)
    ops.linear(x, w).backward(dy)
    assert_close_bf16(x.grad, dy.float() @ w.detach().float(), what="dgrad via W^T", tol=1e-2)
    assert_close_bf16(w.grad, dy.float().t() @ x.detach().float(), what="wgrad", tol=1e-2)


def test_casts(ops):
    x = rnd(1000, 37, dtype=F32, seed=1)
    assert torch.equal(ops.cast(x, BF16), x.to(BF16))
    assert torch.equal(ops.cast(x.to(BF16), F32), x.to(BF16).float())


def test_adamw_and_clip(ops):
    from kai0_amd.optim import FusedAdamW

    torch.manual_seed(0)
    params = [rnd(1000, 33, seed=1).requires_grad_(True), rnd(513, dtype=F32, seed=2).requires_grad_(True)]
    ref = [p.detach().float().clone().requires_grad_(True) for p in params]
    opt = FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0)
    ropt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-2)
    for step in range(3):
        for p, r in zip(params, ref):
            g = rnd(*p.shape, dtype=F32, seed=10 + step) * 3
            p.grad = g.to(p.dtype).clone()
            r.grad = g.to(p.dtype).float().clone()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        norm = opt.step()
        assert abs(float(norm) - float(norm_ref)) / float(norm_ref) < 1e-4
    for p, r, m in zip(params, ref, opt.master_params()):
        assert rel_err(m, r) < 1e-5
        assert torch.equal(p.detach(), m.to(p.dtype))


@pytest.mark.parametrize("gdtype", [BF16, F32])
def test_adamw_rows_is_bit_identical_to_the_dense_update(ops, gdtype):
    """kai0_adamw_rows (embedding table: rows with zero moments and a zero gradient are skipped) against kai0_adamw on the same
    buffers over six steps: a different sparse set of rows receives gradients every step (so rows turn active over time and keep
    being updated with a zero gradient afterwards), clip coefficient from device memory, negative zeros among the gradients."""
    from kai0_amd import optim

    rows, rl = 3000, 264
    n = rows * rl
    g0 = torch.Generator(device=dev()).manual_seed(3)
    p0 = (torch.randn(n, device=dev(), generator=g0) * 0.02).to(BF16)
    state = {k: [p0.float().clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev()), p0.clone()] for k in ("dense", "rows")}
    active = torch.zeros(rows, dtype=torch.uint8, device=dev())
    coef = torch.tensor([0.37], device=dev())
    kw = dict(beta1=0.9, beta2=0.95, eps=1e-8, wd=1e-10)
    assert optim.sparse_rows_ok(2.5e-5, 1e-10) and not optim.sparse_rows_ok(2.5e-5, 1e-2)
    touched = torch.zeros(rows, dtype=torch.bool, device=dev())
    for step in range(1, 7):
        grad = torch.zeros(rows, rl, device=dev())
        idx = torch.randperm(rows, device=dev(), generator=g0)[: 40 + 10 * step]
        grad[idx] = torch.randn(idx.numel(), rl, device=dev(), generator=g0)
        grad[idx[:3], ::2] = -0.0
        touched[idx] = True
        grad = grad.reshape(-1).to(gdtype)
        lr = 2.5e-5 * step / 6
        m, e, v, p = state["dense"]
        optim.adamw_step_(m, e, v, grad, p, lr=lr, step=step, clip_coef=coef, **kw)
        m, e, v, p = state["rows"]
        optim.adamw_rows_step_(m, e, v, grad, p, rl, active, lr=lr, step=step, clip_coef=coef, **kw)
    for a, b in zip(state["dense"], state["rows"]):
        assert torch.equal(a, b)
    assert torch.equal(active.bool(), touched)
    assert bool((state["rows"][1].view(rows, rl)[~touched] == 0).all())  # idle rows: moments still exactly zero
    with pytest.raises(Exception, match="does not round away"):
        m, e, v, p = state["rows"]
        optim.adamw_rows_step_(m, e, v, grad, p, rl, active, lr=1e-3, step=7, clip_coef=coef, beta1=0.9, beta2=0.95, eps=1e-8, wd=1e-2)


@pytest.mark.parametrize("kind,M,N,K,split", [(2, 768, 1152, 4304, 4), (2, 768, 1152, 1152, 3), (1, 968, 2048, 16384, 6), (1, 50, 64, 512, 2),
                                             (2, 37, 1152, 4304, 12)])
def test_splitk_gemm_with_the_consumer_norm_in_its_reduction(ops, kind, M, N, K, split):
    """kai0_gemm_bf16 norm_kind: x = Linear(a) + bias + residual exactly as without it (bit-identical), and norm(x) against the
    separate norm kernels (same arithmetic; the row statistics are summed block-wide instead of wave-wide) and an fp32 reference."""
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    nw = rnd(N, dtype=F32 if kind == 1 else BF16, seed=5, scale=0.3)
    nb = None if kind == 1 else rnd(N, seed=6, scale=0.3)
    eps = 1e-6
    want = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, w, want, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, residual=res, ldr=N, split_k=split)
    got, normed = torch.empty_like(want), torch.zeros_like(want)
    ops.gemm(a, w, got, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, residual=res, ldr=N, split_k=split, norm=(kind, normed, nw, nb, eps))
    assert torch.equal(got, want)
    sep = ops.rmsnorm(want, nw, eps) if kind == 1 else ops.layernorm(want, nw, nb, eps)
    x = want.float()
    if kind == 1:
        ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * (1 + nw)
    else:
        ref = torch.nn.functional.layer_norm(x, (N,), nw.float(), nb.float(), eps)
    assert_close_bf16(normed, ref, what="fused norm vs fp32", tol=1e-2)
    mism = (normed != sep).float().mean().item()
    assert mism < 2e-3 and rel_err(normed, sep) < 1e-3, (mism, rel_err(normed, sep))
    with pytest.raises(Exception, match="split_k > 1"):
        ops.gemm(a, w, got, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, split_k=1, norm=(kind, normed, nw, nb, eps))


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("chunks,n", [(1, 4096), (4, 1024 * 37), (8, 8 * 250_001)])
def test_sum_chunks_is_an_f32_sum_with_one_rounding(ops, dtype, chunks, n):
    """kai0_sum_chunks (the local half of the all-pairs reduce-scatter, sharded.py): bit-exact against the f32 sum in slice order."""
    from kai0_amd.optim import sum_chunks_

    src = rnd(chunks, n, dtype=dtype, seed=3) * 7
    out = torch.empty(n, dtype=dtype, device=dev())
    sum_chunks_(src.reshape(-1), chunks, out)
    acc = torch.zeros(n, dtype=F32, device=dev())
    for j in range(chunks):
        acc += src[j].float()
    assert torch.equal(out, acc.to(dtype))


# ------------------------------------------------------------------------------------ skinny (denoise) GEMM
@pytest.mark.parametrize("M,K,N", [(50, 1024, 1024), (100, 1024, 512), (7, 512, 64)])
def test_skinny_plain_gate_residual(ops, M, K, N):
    """mode 0 without split: row-mapped A, gate * y + residual epilogue; same rounding points as kai0_gemm_bf16."""
    rpb, S_ld, row0 = (50 if M % 50 == 0 else M), 72, 11
    nb = M // rpb
    a_buf = rnd(nb * S_ld, K, seed=1)
    w = rnd(N, K, seed=2, scale=0.05)
    gate = rnd(nb, N, seed=3)
    res = rnd(M, N, seed=4)
    ref = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a_buf, w, ref, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_map=(rpb, S_ld, row0), gate=gate, gate_rpb=rpb, gate_ld=N,
             residual=res, ldr=N)
    out = torch.zeros(M, N, dtype=BF16, device=dev())
    ops.skinny_gemm(a_buf, w, M=M, N=N, K=K, lda=K, ldw=K, segs=[(out, N, 0, N, 0)], a_map=(rpb, S_ld, row0), gate=gate,
                    gate_rpb=rpb, gate_ld=N, residual=res, ldr=N)
    rows = torch.cat([torch.arange(rpb) + b * S_ld + row0 for b in range(nb)]).to(dev())
    y = (a_buf[rows].float() @ w.float().t()).to(BF16).float()
    y = (y * gate.float().repeat_interleave(rpb, 0)).to(BF16).float()
    y = (y + res.float()).to(BF16)
    assert_close_bf16(out, y.float(), what="skinny plain", tol=1e-2)
    assert rel_err(out, ref) < 2e-3


@pytest.mark.parametrize("M,K,N,split", [(50, 2048, 1024, 4), (50, 4096, 1024, 8), (100, 4096, 1024, 4)])
def test_skinny_splitk_partials_and_adarms_combine(ops, M, K, N, split):
    """o_proj / down_proj form: split-K partial products + kai0_adarms_combine (sum, gated residual, adaRMS) ==
    kai0_gemm_bf16 with the fused gate/residual epilogue followed by kai0_adarms_fwd; bit-stable across launches."""
    rpb, S_ld, row0 = 50, 72, 11
    nb = M // rpb
    a_buf = rnd(nb * S_ld, K, seed=1)
    w = rnd(N, K, seed=2, scale=0.05)
    gate = rnd(nb, N, seed=3)
    res = rnd(M, N, seed=4)
    mod = rnd(nb, 3 * N, dtype=F32, seed=5, scale=0.3)
    x_ref = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a_buf, w, x_ref, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_map=(rpb, S_ld, row0), gate=gate, gate_rpb=rpb,
             gate_ld=N, residual=res, ldr=N)
    y_ref, g_ref = ops.adarms(x_ref, mod, rpb, 1e-6)
    ws = ops.skinny_workspace(M, N, split, dev())
    outs = []
    for _ in range(2):
        ws.fill_(float("nan"))
        ops.skinny_gemm(a_buf, w, M=M, N=N, K=K, lda=K, ldw=K, split_k=split, workspace=ws, a_map=(rpb, S_ld, row0))
        outs.append(ops.adarms_combine(ws, gate, res, mod, rpb, 1e-6))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    x, y, g = outs[0]
    assert torch.equal(g, g_ref)
    assert rel_err(x, x_ref) < 2e-3 and rel_err(y, y_ref) < 3e-3
    # adaRMS of the combined x itself is exact
    y2, _ = ops.adarms(x, mod, rpb, 1e-6)
    assert torch.equal(y, y2)


def test_skinny_qkv_rope_segments(ops):
    """fused q|k|v projection + RoPE written through the row map into three buffers == separate GEMMs + kai0_rope_inplace."""
    B, Hs, P, S_ld, H, HD, K = 1, 50, 30, 88, 8, 256, 1024
    M = B * Hs
    x = rnd(M, K, seed=1)
    wq, wk, wv = rnd(H * HD, K, seed=2, scale=0.05), rnd(HD, K, seed=3, scale=0.05), rnd(HD, K, seed=4, scale=0.05)
    wqkv = torch.cat([wq, wk, wv], 0).contiguous()
    pos = (torch.arange(Hs, device=dev(), dtype=torch.int32) + 777).view(B, Hs).contiguous()
    inv_freq = (1.0 / (10000 ** (torch.arange(0, HD, 2, device=dev(), dtype=F32) / HD))).to(BF16).float()
    # reference path
    q_ref = torch.zeros(B, S_ld, H * HD, dtype=BF16, device=dev())
    k_ref = torch.zeros(B, S_ld, HD, dtype=BF16, device=dev())
    v_ref = torch.zeros(B, S_ld, HD, dtype=BF16, device=dev())
    for w_, dst, width in ((wq, q_ref, H * HD), (wk, k_ref, HD), (wv, v_ref, HD)):
        ops.gemm(x, w_, dst, M=M, N=width, K=K, lda=K, ldb=K, ldc=width, c_map=(Hs, S_ld, P))
    ops.rope_(q_ref, pos, inv_freq, B, Hs, S_ld, P, H, HD)
    ops.rope_(k_ref, pos, inv_freq, B, Hs, S_ld, P, 1, HD)
    # fused path
    q = torch.zeros_like(q_ref)
    k = torch.zeros_like(k_ref)
    v = torch.zeros_like(v_ref)
    cos, sin = ops.rope_table(pos, inv_freq)
    N = (H + 2) * HD
    ops.skinny_gemm(x, wqkv, M=M, N=N, K=K, lda=K, ldw=K, mode=1, pair_stride=HD // 2,
                    segs=[(q, H * HD, 0, H * HD, 1), (k, HD, H * HD, H * HD + HD, 1), (v, HD, H * HD + HD, N, 0)],
                    c_map=(Hs, S_ld, P), rope_cos=cos, rope_sin=sin, rope_half=HD // 2)
    for a, b, name in ((q, q_ref, "q"), (k, k_ref, "k"), (v, v_ref, "v")):
        assert rel_err(a, b) < 3e-3, name
        assert torch.equal(a[:, :P], b[:, :P]) and torch.equal(a[:, P + Hs:], b[:, P + Hs:]), name  # untouched rows


def test_skinny_geglu(ops):
    M, K, F = 50, 1024, 4096
    x = rnd(M, K, seed=1)
    wg, wu = rnd(F, K, seed=2, scale=0.05), rnd(F, K, seed=3, scale=0.05)
    wgu = torch.cat([wg, wu], 0).contiguous()
    h = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(x, wgu, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, segs=[(h, F, 0, F, 0)])
    g = (x.float() @ wg.float().t()).to(BF16).float()
    u = (x.float() @ wu.float().t()).to(BF16).float()
    ref = (torch.nn.functional.gelu(g, approximate="tanh").to(BF16).float() * u).to(BF16)
    assert_close_bf16(h, ref.float(), what="skinny geglu", tol=1e-2)


@pytest.mark.parametrize("B,ncam,n_img,T,Hs", [(1, 3, 256, 200, 50), (3, 2, 16, 24, 5), (2, 1, 7, 1, 1), (4, 6, 256, 200, 50)])
def test_prefix_codes_equal_build_mask_codes_bit_for_bit(ops, B, ncam, n_img, T, Hs):
    """kai0_prefix_codes (one launch) against the torch restatement of make_att_2d_masks' inputs (pi0_pytorch.py:52-81,186-235,237-314,343):
    pad = [camera masks expanded | prompt mask | ones], att = [0 ... 0 | 1 0 ... 0] -> build_mask_codes.  Integer logic: equal bits,
    with masked-out cameras, ragged prompt masks (holes included) and rows whose every prompt token is padding."""
    from kai0_amd.model import build_mask_codes

    g = torch.Generator().manual_seed(B * 131 + ncam)
    img_masks = [(torch.rand(B, generator=g) > 0.3).to(dev()) for _ in range(ncam)]
    lang = (torch.rand(B, T, generator=g) > 0.35)
    lang[0] = torch.arange(T) < max(1, T // 2)  # the usual form: valid tokens first
    if B > 1:
        lang[1] = False
    lang = lang.to(dev())
    P = ncam * n_img + T
    pad = torch.cat([m[:, None].expand(B, n_img) for m in img_masks] + [lang, torch.ones(B, Hs, dtype=torch.bool, device=dev())], dim=1)
    att = torch.zeros(B, P + Hs, dtype=torch.bool, device=dev())
    att[:, P] = True
    want = build_mask_codes(pad, att)
    got = ops.prefix_codes(img_masks, lang, n_img, Hs)
    for a, b, name in zip(got, want, ("qcode", "kcode", "pos")):
        assert a.dtype == torch.int32 and torch.equal(a, b), name


def test_rope_table_bf16_output_holds_the_same_values(ops):
    pos = torch.randint(0, 1100, (977,), dtype=torch.int32, device=dev())
    inv_freq = (1.0 / (10000 ** (torch.arange(0, 256, 2).float() / 256))).to(BF16).float().to(dev())
    c32, s32 = ops.rope_table(pos, inv_freq)
    c16, s16 = ops.rope_table(pos, inv_freq, bf16=True)
    assert c16.dtype == BF16 and torch.equal(c16.float(), c32) and torch.equal(s16.float(), s32)


@pytest.mark.parametrize("B,Hs,P", [(1, 50, 968), (2, 50, 200), (1, 3, 29)])
def test_attn_decode_matches_gemm_softmax_gemm(ops, B, Hs, P):
    """one-launch decode attention (transposed value cache) == logits GEMM + masked softmax + P V GEMM."""
    H, HD = 8, 256
    S = P + Hs
    S_ld = (S + 31) // 32 * 32
    q = rnd(B, S_ld, H * HD, seed=1)
    k = rnd(B, S_ld, HD, seed=2)
    v = rnd(B, S_ld, HD, seed=3)
    vt = v.transpose(1, 2).contiguous()
    g = torch.Generator().manual_seed(4)
    pad = torch.rand(B, S, generator=g) > 0.1
    pad[:, P:] = True
    att = torch.zeros(B, S, dtype=torch.bool)
    att[:, P] = True
    from kai0_amd.model import build_mask_codes

    qcode, kcode, _ = build_mask_codes(pad.to(dev()), att.to(dev()))
    M = Hs * H
    # reference: three launches
    scores = torch.empty(B, M, S_ld, dtype=BF16, device=dev())
    ops.gemm(q, k, scores, M=M, N=S_ld, K=HD, lda=HD, ldb=HD, ldc=S_ld, batch=B, sA=(S_ld * H * HD, 0), sB=(S_ld * HD, 0),
             sC=(M * S_ld, 0), scale=HD**-0.5, a_off_elems=P * H * HD)
    from kai0_amd import _lib

    _lib.call("kai0_softmax_mask_fwd", scores.data_ptr(), scores.data_ptr(), qcode.data_ptr(), kcode.data_ptr(), B, Hs, H, S,
              S_ld, M * S_ld, P, qcode.stride(0), kcode.stride(0), ops._stream())
    ref = torch.zeros(B, S_ld, H * HD, dtype=BF16, device=dev())
    ops.gemm(scores, v, ref, M=M, N=HD, K=S_ld, a_kc=True, b_kc=False, lda=S_ld, ldb=HD, ldc=HD, batch=B, sA=(M * S_ld, 0),
             sB=(S_ld * HD, 0), sC=(S_ld * H * HD, 0), c_off_elems=P * H * HD)
    out = torch.zeros_like(ref)
    ops.attn_decode(q, k, vt, out, qcode, kcode, batch=B, rows=M, H=H, HD=HD, Sk=S, q0=P, q_bs=S_ld * H * HD, k_bs=S_ld * HD,
                    k_ld=HD, k_rows=S_ld, vt_bs=HD * S_ld, vt_ld=S_ld, scale=HD**-0.5)
    out2 = torch.zeros_like(ref)
    ops.attn_decode(q, k, vt, out2, qcode, kcode, batch=B, rows=M, H=H, HD=HD, Sk=S, q0=P, q_bs=S_ld * H * HD, k_bs=S_ld * HD,
                    k_ld=HD, k_rows=S_ld, vt_bs=HD * S_ld, vt_ld=S_ld, scale=HD**-0.5)
    assert torch.equal(out, out2)
    assert torch.equal(out[:, :P], ref[:, :P]) and torch.equal(out[:, S:], ref[:, S:])  # only the query rows are written
    assert rel_err(out[:, P:S], ref[:, P:S]) < 3e-3
    # fp32 ground truth
    qf = q[:, P:S].float().view(B, Hs, H, HD)
    logits = (torch.einsum("bshd,bkd->bshk", qf, k[:, :S].float()).to(BF16).float() * HD**-0.5).to(BF16).float()
    allowed = kcode[:, None, :S] <= qcode[:, P:S, None]
    logits = logits.masked_fill(~allowed[:, :, None, :], float("-inf"))
    pr = torch.softmax(logits, -1).to(BF16).float()
    gt = torch.einsum("bshk,bkd->bshd", pr, v[:, :S].float()).reshape(B, Hs, H * HD)
    assert rel_err(out[:, P:S], gt) < 4e-3


@pytest.mark.parametrize("P,valid", [(968, 861), (512, 40)])
def test_attention_key_split_matches_the_one_range_kernel_and_fp32(ops, P, valid):
    """The B = 1 prefix attention as four key ranges of the one-pass kernel + kai0_attn_combine (kai0hip.h) against the same kernel over
    all keys in one range (same rounding points up to the ranges' own bf16 outputs) and against an fp32 softmax with the reference's
    rounding of the logits: prefix-LM codes with padded prompt slots — whole ranges / key tiles of invisible keys when only 40 of the
    512 positions are valid — and padded query rows (which must come out as zeros)."""
    from kai0_amd.model import build_mask_codes

    H, HD, S_ld = 8, 256, 1024
    q = rnd(1, S_ld, H * HD, seed=1)
    k = rnd(1, S_ld, HD, seed=2)
    v = rnd(1, S_ld, HD, seed=3)
    pad = torch.zeros(1, P, dtype=torch.bool)
    pad[:, :valid] = True
    pad[:, 5] = False  # a hole in the middle as well
    qcode, kcode, _ = build_mask_codes(pad.to(dev()), torch.zeros(1, P, dtype=torch.bool, device=dev()))
    M = P * H
    one = torch.zeros(1, S_ld, H * HD, dtype=BF16, device=dev())
    ops.attn_fwd(q, k, v, one, None, rows=M, Sk=P, HD=HD, H=H, q0=0, batch=1, ldq=HD, ldk=HD, ldv=HD, ldo=HD, sQ=(S_ld * H * HD, 0),
                 sK=(S_ld * HD, 0), sV=(S_ld * HD, 0), sO=(S_ld * H * HD, 0), qcode=qcode, kcode=kcode, scale=HD**-0.5, online=1)
    out = torch.full((1, S_ld, H * HD), 7.0, dtype=BF16, device=dev())
    ops.attn_fwd_keysplit(q, k, v, out, rows=M, Sk=P, HD=HD, H=H, q0=0, ldk=HD, ldv=HD, qcode=qcode[0], kcode=kcode[0], scale=HD**-0.5,
                          q_off=0, o_off=0, parts=4)
    assert float(out[:, P:].float().min()) == 7.0  # rows past the queries untouched
    qf = q[:, :P].float().view(1, P, H, HD)
    logits = (torch.einsum("bshd,bkd->bshk", qf, k[:, :P].float()).to(BF16).float() * HD**-0.5).to(BF16).float()
    allowed = kcode[:, None, :P] <= qcode[:, :P, None]
    logits = logits.masked_fill(~allowed[:, :, None, :], float("-inf"))
    pr = torch.nan_to_num(torch.softmax(logits, -1), nan=0.0)
    gt = torch.einsum("bshk,bkd->bshd", pr, v[:, :P].float()).reshape(1, P, H * HD)
    live = pad[0].to(dev())
    assert float(out[0, :P][~live].float().abs().max()) == 0.0  # padded query rows: zeros
    r_one, r_gt = rel_err(out[0, :P][live], one[0, :P][live]), rel_err(out[0, :P][live], gt[0][live])
    print(f"key split vs one range {r_one:.3e}, vs fp32 softmax {r_gt:.3e}; one range vs fp32 {rel_err(one[0, :P][live], gt[0][live]):.3e}")
    assert r_one < 4e-3 and r_gt < 4e-3


def test_skinny_transposed_value_segment(ops):
    B, Hs, P, S_ld, HD, K = 2, 50, 30, 96, 256, 1024
    M = B * Hs
    x = rnd(M, K, seed=1)
    w = rnd(2 * HD, K, seed=2, scale=0.05)
    a = torch.zeros(B, S_ld, HD, dtype=BF16, device=dev())
    bt = torch.zeros(B, HD, S_ld, dtype=BF16, device=dev())
    ops.skinny_gemm(x, w, M=M, N=2 * HD, K=K, lda=K, ldw=K, mode=1, pair_stride=HD // 2,
                    segs=[(a, HD, 0, HD, 0), (bt, S_ld, HD, 2 * HD, 2)], c_map=(Hs, S_ld, P))
    ref = torch.zeros(B, S_ld, HD, dtype=BF16, device=dev())
    ops.skinny_gemm(x, w[HD:].contiguous(), M=M, N=HD, K=K, lda=K, ldw=K, mode=1, pair_stride=HD // 2,
                    segs=[(ref, HD, 0, HD, 0)], c_map=(Hs, S_ld, P))
    assert torch.equal(bt.transpose(1, 2), ref)
    src = rnd(3, 40, 64, seed=5)
    dst = torch.zeros(3, 64, 48, dtype=BF16, device=dev())
    ops.transpose_strided(src, dst, R=40, C=64, src_ld=64, dst_ld=48, batch=3, src_bs=40 * 64, dst_bs=64 * 48)
    assert torch.equal(dst[:, :, :40], src.transpose(1, 2)) and not bool(dst[:, :, 40:].any())


def test_gemm_softmax_backward_epilogue_and_rowdot(ops):
    """act 4: dS = bf16((P * (dO V^T - D)) * scale) with D = rowdot(dO, O) == softmax backward of the f32 dP with the
    row term <dP, P> (equal up to the bf16 rounding of O = P V)."""
    from kai0_amd import _lib

    B, M, S, HD = 2, 200, 136, 64
    g = torch.Generator().manual_seed(0)
    probs = torch.softmax(torch.randn(B, M, S, generator=g) * 2, -1).to(BF16).to(dev())
    v = rnd(B, S, HD, seed=1)
    do = rnd(B, M, HD, seed=2)
    o = torch.bmm(probs.float(), v.float()).to(BF16)
    scale = HD**-0.5
    dsum = ops.rowdot(do, o, HD)
    assert rel_err(dsum, (do.float() * o.float()).sum(-1).reshape(-1)) < 1e-5
    ds = torch.empty_like(probs)
    ops.gemm(do, v, ds, M=M, N=S, K=HD, lda=HD, ldb=HD, ldc=S, batch=B, sA=(M * HD, 0), sB=(S * HD, 0), sC=(M * S, 0), act=4,
             aux1=probs, rowvec=dsum, rv=(M, 0, 1), scale=scale)
    dp = torch.bmm(do.float(), v.float().transpose(1, 2))
    ref = probs.float() * (dp - (dp * probs.float()).sum(-1, keepdim=True)) * scale
    assert rel_err(ds, ref) < 6e-3
    # and against the two-kernel path it replaces (f32 dP written, kai0_softmax_bwd)
    dp32 = torch.empty(B, M, S, dtype=F32, device=dev())
    ops.gemm(do, v, dp32, M=M, N=S, K=HD, lda=HD, ldb=HD, ldc=S, batch=B, sA=(M * HD, 0), sB=(S * HD, 0), sC=(M * S, 0))
    ds2 = torch.empty_like(probs)
    _lib.call("kai0_softmax_bwd", probs.data_ptr(), dp32.data_ptr(), 1, ds2.data_ptr(), B * M, S, S, scale, ops._stream())
    assert rel_err(ds, ds2) < 6e-3


def test_siglip_attention_backward_fused_matches_gemm_path_and_fp32(ops):
    """real tower shape (S = 256, 16 heads x 72): the one-block-per-head backward == the GEMM formulation == fp32 autograd."""
    n, S, NH, HD = 3, 256, 16, 72
    E = NH * HD
    q, k, v = rnd(n * S, E, seed=1, scale=0.6), rnd(n * S, E, seed=2, scale=0.6), rnd(n * S, E, seed=3)
    do = rnd(n * S, E, seed=4)
    outs = {}
    for mode in (True, False):
        ops._SIGLIP_BWD_FUSED = mode
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = ops.siglip_attention(qq, kk, vv, n, S, NH, HD)
        o.backward(do)
        outs[mode] = (o.detach(), qq.grad, kk.grad, vv.grad)
    ops._SIGLIP_BWD_FUSED = True
    for a, b, name in zip(outs[True][1:], outs[False][1:], ("dq", "dk", "dv")):
        assert rel_err(a, b) < 4e-3, name
    Q, K, V = (t.float().view(n, S, NH, HD).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    logits = ((Q @ K.transpose(-1, -2)).to(BF16).float() * HD**-0.5).to(BF16).float()
    O = torch.softmax(logits, -1).to(BF16).float() @ V
    O.backward(do.float().view(n, S, NH, HD).transpose(1, 2))
    for got, ref, name in zip(outs[True][1:], (Q.grad, K.grad, V.grad), ("dq", "dk", "dv")):
        assert rel_err(got.view(n, S, NH, HD).transpose(1, 2), ref) < 1.5e-2, name


@pytest.mark.parametrize("persist", [0, 2])
@pytest.mark.parametrize("case", ["plain", "bias", "bias_f32", "bias_res", "res", "gelu_pre", "gelu_res", "routed", "ragged_rows", "batched", "tn", "accum", "accum_res"])
def test_simple_epilogue_fast_path_equals_the_general_epilogue(ops, case, persist):
    """256 x 256 launches whose epilogue is a store with little else (act 0 / 1, bias, residual, column routing) take a fast path that
    skips the general epilogue's per-row checks (also when it accumulates into the destination); kai0_gemm_desc.general_epilogue = 1 sends them through the general path: same bits in
    every output, in the one-block-per-tile kernels (persist 0) and in the persistent kernel (persist 2), with ragged rows, batch
    entries (their C / residual strides) and the transpose-read layout of the weight gradients."""
    from kai0_amd import _lib

    lib = _lib.load()
    M, N, K = (4000, 4096, 320) if case == "ragged_rows" else (4096, 4096, 320)
    if persist == 2:
        N = 8192  # (512 tiles: the persistent kernel's minimum; batched / transpose-read launches never take it)
    kw, nout, batch = {}, 1, 1
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    lay = dict(lda=K, ldb=K)
    if case == "bias":
        kw = dict(bias=rnd(N, seed=3))
    elif case == "bias_f32":
        kw = dict(bias=rnd(N, seed=3).float())
    elif case in ("bias_res", "ragged_rows"):
        kw = dict(bias=rnd(N, seed=3), residual=rnd(M, N, seed=4), ldr=N)
    elif case == "res":
        kw = dict(residual=rnd(M, N, seed=4), ldr=N)
    elif case == "gelu_pre":
        kw, nout = dict(bias=rnd(N, seed=3), act=1), 2
    elif case == "gelu_res":
        kw, nout = dict(act=1, residual=rnd(M, N, seed=4), ldr=N), 2
    elif case == "batched":
        batch = 2
        A, W = rnd(batch * M, K, seed=1), rnd(batch * N, K, seed=2, scale=0.05)
        kw = dict(batch=batch, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0), sR=(M * N, 0), bias=rnd(N, seed=3), residual=rnd(batch * M, N, seed=4), ldr=N)
    elif case == "tn":  # C[M][N] = A[K][M]^T B[K][N]
        A, W = rnd(K, M, seed=1), rnd(K, N, seed=2, scale=0.05)
        lay = dict(a_kc=False, b_kc=False, lda=M, ldb=N)
    elif case == "accum":  # C += A W^T (the destination's 3.0 fill is what it accumulates into)
        kw = dict(accumulate=True)
    elif case == "accum_res":
        kw = dict(accumulate=True, bias=rnd(N, seed=3), residual=rnd(M, N, seed=4), ldr=N)
    res = {}
    for simple in (0, 1):
        with ops.gemm_tuning(persist=2 if persist == 2 else 1, general_epilogue=1 - simple):
            outs = [torch.full((batch * M, N), 3.0, dtype=BF16, device=dev()) for _ in range(nout)]
            k2 = dict(kw)
            if nout >= 2:
                k2["pre_out"] = outs[1]
            if case == "routed":  # columns [0, N/2) | [N/2, 3N/4) | [3N/4, N) to three destinations with their own row strides
                dst = [torch.full((M, w), 3.0, dtype=BF16, device=dev()) for w in (N // 2, N // 4, N // 4)]
                k2["segs"] = [(dst[0], N // 2, 0), (dst[1], N // 4, N // 2), (dst[2], N // 4, 3 * N // 4)]
                outs = outs + dst
            ops.gemm(A, W, outs[0], M=M, N=N, K=K, ldc=N, **lay, **k2)
            torch.cuda.synchronize()
            res[simple] = outs
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    if case == "plain":
        assert rel_err(res[1][0], A.float() @ W.float().t()) < 5e-3
    if case == "routed":
        assert not torch.equal(res[1][1], torch.full_like(res[1][1], 3.0))  # the routed destinations were written


@pytest.mark.parametrize("case", ["plain", "bias", "bias_res", "gelu_pre", "split", "split_norm", "routed", "rowmap", "ragged", "fc2"])
def test_eight_wave_128_tile_is_bit_identical_to_the_four_wave_tile(ops, case):
    """kai0_gemm_desc.small_w8: the 128 x 128 tile run by eight waves (4 x 2 wave tiles of 32 x 64, two waves per SIMD — round 6, for the
    B = 1 passes whose grids are at most one block per CU) against the four-wave tile: the same products in the same order, so every
    output must have the same bits — plain / bias / residual / GELU + pre-activation epilogues, split-K partials (and the norm fused into
    their reduction), column routing, output row maps, ragged edges and a K tail."""
    M, N, K = {"ragged": (700, 1160, 1160), "fc2": (768, 1152, 4304)}.get(case, (768, 1152, 1152))
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    kw, nout = {}, 1
    if case == "bias":
        kw = dict(bias=rnd(N, seed=3))
    elif case in ("bias_res", "ragged", "fc2"):
        kw = dict(bias=rnd(N, seed=3), residual=rnd(M, N, seed=4), ldr=N)
    elif case == "gelu_pre":
        kw, nout = dict(bias=rnd(N, seed=3), act=1), 2
    elif case == "split":
        kw = dict(split_k=3, bias=rnd(N, seed=3), residual=rnd(M, N, seed=4), ldr=N)
    res = {}
    for mode in (1, 2):
        with ops.gemm_tuning(small_w8=mode):
            outs = [torch.full((M, N), 3.0, dtype=BF16, device=dev()) for _ in range(nout)]
            k2 = dict(kw)
            if nout >= 2:
                k2["pre_out"] = outs[1]
            if case == "split_norm":  # LayerNorm of the rows inside the reduction launch
                nout_t = torch.full((M, N), 3.0, dtype=BF16, device=dev())
                k2.update(split_k=3, bias=rnd(N, seed=3), residual=rnd(M, N, seed=4), ldr=N,
                          norm=(2, nout_t, rnd(N, seed=5), rnd(N, seed=6), 1e-6))
                outs = outs + [nout_t]
            if case == "routed":
                dst = [torch.full((M, w), 3.0, dtype=BF16, device=dev()) for w in (N // 2, N // 4, N // 4)]
                k2["segs"] = [(dst[0], N // 2, 0), (dst[1], N // 4, N // 2), (dst[2], N // 4, 3 * N // 4)]
                outs = outs + dst
            if case == "rowmap":  # rows of 3 "batch entries" of 256 into a buffer padded to 320 rows each, from row 8
                big = torch.full((3 * 320, N), 3.0, dtype=BF16, device=dev())
                k2["c_map"] = (256, 320, 8)
                outs = [big]
            ops.gemm(A, W, outs[0], M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **k2)
            torch.cuda.synchronize()
            res[mode] = outs
    for a, b in zip(res[1], res[2]):
        assert torch.equal(a, b)
    if case == "plain":
        assert rel_err(res[2][0], A.float() @ W.float().t()) < 5e-3
    if case == "routed":
        assert not torch.equal(res[2][1], torch.full_like(res[2][1], 3.0))


@pytest.mark.parametrize("case", ["plain", "bias_res", "gelu_pre", "geglu_pair", "geglu_fwd", "geglu_bwd", "gelu_bwd", "ragged"])
def test_persistent_gemm_is_bit_identical_to_one_block_per_tile(ops, case):
    """kai0_gemm_desc.persist = 2: the persistent NT kernel (dynamic per-XCD tile queue, next tile staged before the epilogue, 16-row
    epilogue slabs) against the one-block-per-tile launches on every epilogue it serves: same bits, every output (C, pre_out,
    pre_out2), including ragged edges and more tiles than resident blocks."""
    from kai0_amd import _lib

    lib = _lib.load()
    M, N, K = (4096, 8192, 320) if case != "ragged" else (4000, 8200, 328)
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    kw, nout = {}, 1
    if case == "bias_res":
        kw = dict(bias=rnd(N, seed=3), residual=rnd(M, N, seed=4), ldr=N)
    elif case == "gelu_pre":
        kw, nout = dict(bias=rnd(N, seed=3), act=1), 2
    elif case == "geglu_pair":
        kw, nout = dict(act=6, B2=rnd(N, K, seed=5, scale=0.05)), 3
    elif case == "geglu_fwd":
        kw, nout = dict(act=2, aux1=rnd(M, N, seed=6)), 2
    elif case == "geglu_bwd":
        kw, nout = dict(act=3, aux1=rnd(M, N, seed=6), aux2=rnd(M, N, seed=7)), 2
    elif case == "gelu_bwd":
        kw = dict(act=5, aux1=rnd(M, N, seed=6))
    res = {}
    for mode in (0, 2):
        with ops.gemm_tuning(persist=2 if mode == 2 else 1):
            outs = [torch.full((M, N), 3.0, dtype=BF16, device=dev()) for _ in range(nout)]
            k2 = dict(kw)
            if nout >= 2:
                k2["pre_out"] = outs[1]
            if nout >= 3:
                k2["pre_out2"] = outs[2]
            ops.gemm(A, W, outs[0], M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **k2)
            torch.cuda.synchronize()
            res[mode] = outs
    for a, b in zip(res[0], res[2]):
        assert torch.equal(a, b)
    ref = A.float() @ W.float().t()  # and it is the right product (plain case: against fp32)
    if case == "plain":
        assert rel_err(res[2][0], ref) < 5e-3


@pytest.mark.parametrize("B", [1, 2])
def test_skinny_folded_adarms_and_producer_row_statistics(ops, B):
    """The folded form of the adaRMS projections of the denoise loop (kai0hip.h rowsq_in): the producer launch (o_proj-style mode 0
    with gate + residual) hands over per-column-tile sums of squares of the rows it stores, the consumer multiplies the RAW rows
    by W' = bf16(W (1 + scale)) and applies rstd and c = W shift to the f32 sum.  Against fp32 math of the unfolded definition
    (modeling_gemma.py:49-104 + the Linear) and against the adaRMS-prologue form it replaces (other rounding points: tolerance)."""
    Hs, K, F, Kp = 50, 1024, 4096, 2048
    M = B * Hs
    a_prev = rnd(M, Kp, seed=1)
    w_prev = rnd(K, Kp, seed=2, scale=0.03)
    gate = rnd(B, K, seed=3)
    res = rnd(M, K, seed=4)
    x = torch.empty(M, K, dtype=BF16, device=dev())
    sq = torch.zeros(K // 16, M, dtype=F32, device=dev())
    ops.skinny_gemm(a_prev, w_prev, M=M, N=K, K=Kp, lda=Kp, ldw=Kp, split_k=-1, segs=[(x, K, 0, K, 0)], gate=gate, gate_rpb=Hs,
                    gate_ld=K, residual=res, ldr=K, rowsq_out=sq)
    assert torch.allclose(sq.sum(0), x.float().pow(2).sum(1), rtol=1e-5, atol=1e-3)  # the partials add up to the rows' sums of squares
    mod = rnd(B, 3 * K, dtype=F32, seed=9, scale=0.3)
    scale, shift = mod[0, :K], mod[0, K : 2 * K]  # (the engine's modulation is the same for every sample of a step)
    modb = mod[:1].expand(B, -1).contiguous()
    wgu = rnd(2 * F, K, seed=5, scale=0.05)
    wp = (wgu.float() * (1 + scale)[None, :]).to(BF16)
    cvec = (wgu.float() @ shift).contiguous()
    h = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(x, wp, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, split_k=-1, segs=[(h, F, 0, F, 0)], eps=1e-6,
                    rowsq_in=sq, rowsq_parts=K // 16, cvec=cvec)
    h_mod = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(x, wgu, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, split_k=-1, segs=[(h_mod, F, 0, F, 0)], mod=modb,
                    mod_ld=3 * K, mod_rpb=Hs, eps=1e-6)
    xf = x.float()
    yn = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)) * (1 + scale) + shift
    gu = yn @ wgu.float().t()
    ref = torch.nn.functional.gelu(gu[:, :F], approximate="tanh") * gu[:, F:]
    e_fold, e_mod = rel_err(h, ref), rel_err(h_mod, ref)
    print(f"folded vs fp32 {e_fold:.3e}; adaRMS-prologue form vs fp32 {e_mod:.3e}; folded vs prologue form {rel_err(h, h_mod):.3e}")
    assert e_fold < 8e-3 and e_fold < 1.5 * e_mod + 1e-3
    # packed weights and a single partial (the step's first layer: the glue kernel writes one sum per row): same result
    sq1 = x.float().pow(2).sum(1)[None].contiguous()
    h2 = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(x, ops.pack_skinny_weight(wp), M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, split_k=-1,
                    segs=[(h2, F, 0, F, 0)], eps=1e-6, rowsq_in=sq1, rowsq_parts=1, cvec=cvec, w_packed=True)
    assert rel_err(h2, h) < 1e-3


@pytest.mark.parametrize("n", [3, 9])
def test_siglip_attention_forward_dedicated_kernel(ops, n):
    """kai0_siglip_attn_fwd at the real tower's shape (256 tokens, 16 heads x 72): n = 3 runs the four-blocks-per-head form (B = 1
    inference), n = 9 the head-per-block form (training); q / k / v as column slices of one stacked buffer.  Against the reference's
    choreography in torch (bf16 logits, f32 softmax, bf16 P) and the log-sum-exp the recompute backward consumes."""
    S, NH, HD = 256, 16, 72
    E = NH * HD
    qkv = rnd(n * S, 3 * E, seed=5, scale=0.7)
    q, k, v = qkv[:, :E], qkv[:, E : 2 * E], qkv[:, 2 * E :]
    out = torch.empty(n * S, E, dtype=BF16, device=dev())
    lse = torch.empty(n * NH, S, dtype=torch.float32, device=dev())
    ops.siglip_attn_fwd(q, k, v, out, n_img=n, S=S, NH=NH, HD=HD, ld_qkv=3 * E, ld_out=E, lse=lse)
    Q, K, V = (t.float().reshape(n, S, NH, HD).transpose(1, 2) for t in (q, k, v))
    logits = ((Q @ K.transpose(-1, -2)).to(BF16).float() * HD**-0.5).to(BF16).float()
    ref = (torch.softmax(logits, -1).to(BF16).float() @ V).transpose(1, 2).reshape(n * S, E)
    assert rel_err(out, ref) < 3e-3
    assert (lse.view(n, NH, S) - torch.logsumexp(logits, -1)).abs().max() < 2e-2
    # the general kernel (kai0_attn_fwd) on the same operands
    out2 = torch.empty_like(out)
    ops.attn_fwd(q, k, v, out2, None, rows=S, Sk=S, HD=HD, H=1, batch=n * NH, batch_inner=NH, ldq=3 * E, ldk=3 * E, ldv=3 * E, ldo=E,
                 sQ=(S * 3 * E, HD), sK=(S * 3 * E, HD), sV=(S * 3 * E, HD), sO=(S * E, HD), scale=HD**-0.5)
    assert rel_err(out, out2) < 4e-3


# ------------------------------------------------------------------------------- in-block skinny GEMMs (split_k = -1)
@pytest.mark.parametrize("M,K,N", [(50, 2048, 1024), (50, 4096, 1024), (100, 4096, 1024), (50, 1024, 1024), (7, 2048, 128)])
def test_skinny_inblock_plain_gate_residual(ops, M, K, N):
    """o_proj / down_proj of the denoise loop in ONE launch: whole contraction inside a block (K / 256 waves), 16-row x 16-column
    tiles, gated residual in the epilogue == kai0_gemm_bf16 with the same fused epilogue (same rounding points)."""
    rpb, S_ld, row0 = (50 if M % 50 == 0 else M), 72, 11
    nb = M // rpb
    a_buf = rnd(nb * S_ld, K, seed=1)
    w = rnd(N, K, seed=2, scale=0.05)
    gate_all = rnd(nb, 3 * N, seed=3)  # gate taken as a strided view (row stride 3 N), as the engine does
    gate = gate_all[:, 2 * N :]
    res = rnd(M, N, seed=4)
    ref = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a_buf, w, ref, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_map=(rpb, S_ld, row0), gate=gate.contiguous(), gate_rpb=rpb,
             gate_ld=N, residual=res, ldr=N)
    outs = []
    for _ in range(2):
        out = torch.full((M, N), float("nan"), dtype=BF16, device=dev())
        ops.skinny_gemm(a_buf, w, M=M, N=N, K=K, lda=K, ldw=K, split_k=-1, segs=[(out, N, 0, N, 0)], a_map=(rpb, S_ld, row0),
                        gate=gate, gate_rpb=rpb, gate_ld=3 * N, residual=res, ldr=N)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0].float()).all()
    rows = torch.cat([torch.arange(rpb) + b * S_ld + row0 for b in range(nb)]).to(dev())
    y = (a_buf[rows].float() @ w.float().t()).to(BF16).float()
    y = (y * gate.float().repeat_interleave(rpb, 0)).to(BF16).float()
    y = (y + res.float()).to(BF16)
    assert_close_bf16(outs[0], y.float(), what="skinny in-block plain", tol=1e-2)
    assert rel_err(outs[0], ref) < 2e-3


@pytest.mark.parametrize("B", [1, 2])
def test_skinny_inblock_adarms_prologue(ops, B):
    """[adaRMS -> q|k|v + RoPE] and [adaRMS -> gate|up + GeGLU] as single launches == kai0_adarms_fwd followed by the plain
    skinny GEMM of the same mode (the norm's row statistics are summed in another order: equal up to rare one-ulp flips)."""
    Hs, P, S_ld, H, HD, K, F = 50, 30, 88, 8, 256, 1024, 4096
    M = B * Hs
    x = rnd(M, K, seed=1)
    mod_all = rnd(B, 5 * 3 * K, dtype=F32, seed=9, scale=0.3)  # stacked modulations: take slot 3 as a strided view
    mod = mod_all[:, 3 * 3 * K : 4 * 3 * K]
    y_ref, _ = ops.adarms(x, mod.contiguous(), Hs, 1e-6)
    wqkv = rnd((H + 2) * HD, K, seed=2, scale=0.05)
    pos = (torch.arange(M, device=dev(), dtype=torch.int32) % Hs + 700).view(B, Hs).contiguous()
    inv_freq = (1.0 / (10000 ** (torch.arange(0, HD, 2, device=dev(), dtype=F32) / HD))).to(BF16).float()
    cos, sin = ops.rope_table(pos, inv_freq)
    N = (H + 2) * HD

    def qkv(a, **kw):
        q = torch.zeros(B, S_ld, H * HD, dtype=BF16, device=dev())
        k = torch.zeros(B, S_ld, HD, dtype=BF16, device=dev())
        vt = torch.zeros(B, HD, S_ld, dtype=BF16, device=dev())
        ops.skinny_gemm(a, wqkv, M=M, N=N, K=K, lda=K, ldw=K, mode=1, pair_stride=HD // 2,
                        segs=[(q, H * HD, 0, H * HD, 1), (k, HD, H * HD, H * HD + HD, 1), (vt, S_ld, H * HD + HD, N, 2)],
                        c_map=(Hs, S_ld, P), rope_cos=cos, rope_sin=sin, rope_half=HD // 2, **kw)
        return q, k, vt

    ref = qkv(y_ref)
    got = qkv(x, split_k=-1, mod=mod, mod_ld=mod_all.shape[1], mod_rpb=Hs, eps=1e-6)
    for a, b, name in zip(got, ref, ("q", "k", "vt")):
        assert rel_err(a, b) < 2e-3, (name, rel_err(a, b))
        assert float((a != b).float().mean()) < 2e-2, name
    assert torch.equal(got[0][:, :P], ref[0][:, :P]) and torch.equal(got[2][:, :, :P], ref[2][:, :, :P])  # untouched rows
    # and against fp32 math of the whole thing
    xf = x.float().view(B, Hs, K)
    rstd = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    yn = ((xf * rstd) * (1 + mod[:, None, :K]) + mod[:, None, K : 2 * K]).to(BF16).float().view(M, K)
    v32 = (yn @ wqkv[(H + 1) * HD :].float().t()).view(B, Hs, HD)
    assert rel_err(got[2][:, :, P : P + Hs].transpose(1, 2), v32) < 6e-3
    wgu = rnd(2 * F, K, seed=3, scale=0.05)
    h_ref = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(y_ref, wgu, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, segs=[(h_ref, F, 0, F, 0)])
    h = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(x, wgu, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, split_k=-1, segs=[(h, F, 0, F, 0)], mod=mod,
                    mod_ld=mod_all.shape[1], mod_rpb=Hs, eps=1e-6)
    assert rel_err(h, h_ref) < 3e-3 and float((h != h_ref).float().mean()) < 3e-2
    # the in-block kernel without the prologue on the already normalised input: the same products in another summation order
    h2 = torch.zeros(M, F, dtype=BF16, device=dev())
    ops.skinny_gemm(y_ref, wgu, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F, split_k=-1, segs=[(h2, F, 0, F, 0)])
    assert rel_err(h2, h_ref) < 2e-3 and float((h2 != h_ref).float().mean()) < 2e-2


def test_skinny_inblock_packed_weights_are_bit_identical(ops):
    """`w_packed` (fragment-major 1-KiB blocks, ops.pack_skinny_weight) only changes where the weight stream is read from: every
    in-block variant — plain + gated residual (K = 1024 / 2048 / 4096), adaRMS + q|k|v + RoPE, adaRMS + gate|up + GeGLU — must
    return the bits of the row-major launch."""
    Hs, P, S_ld, H, HD, F = 50, 30, 88, 8, 256, 4096
    B = 2
    M = B * Hs
    for K, N in ((1024, 1024), (2048, 1024), (4096, 1024), (2048, 128)):
        a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
        gate, res = rnd(B, N, seed=3), rnd(M, N, seed=4)
        outs = []
        for wp, packed in ((w, False), (ops.pack_skinny_weight(w), True)):
            out = torch.full((M, N), float("nan"), dtype=BF16, device=dev())
            ops.skinny_gemm(a, wp, M=M, N=N, K=K, lda=K, ldw=K, split_k=-1, segs=[(out, N, 0, N, 0)], gate=gate, gate_rpb=Hs,
                            gate_ld=N, residual=res, ldr=N, w_packed=packed)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), (K, N)
    K = 1024
    x = rnd(M, K, seed=1)
    mod = rnd(B, 3 * K, dtype=F32, seed=9, scale=0.3)
    wqkv = rnd((H + 2) * HD, K, seed=2, scale=0.05)
    wgu = rnd(2 * F, K, seed=3, scale=0.05)
    pos = (torch.arange(M, device=dev(), dtype=torch.int32) % Hs + 700).view(B, Hs).contiguous()
    inv_freq = (1.0 / (10000 ** (torch.arange(0, HD, 2, device=dev(), dtype=F32) / HD))).to(BF16).float()
    cos, sin = ops.rope_table(pos, inv_freq)
    N = (H + 2) * HD
    got = []
    for packed in (False, True):
        q = torch.zeros(B, S_ld, H * HD, dtype=BF16, device=dev())
        k = torch.zeros(B, S_ld, HD, dtype=BF16, device=dev())
        vt = torch.zeros(B, HD, S_ld, dtype=BF16, device=dev())
        ops.skinny_gemm(x, ops.pack_skinny_weight(wqkv) if packed else wqkv, M=M, N=N, K=K, lda=K, ldw=K, mode=1, pair_stride=HD // 2,
                        segs=[(q, H * HD, 0, H * HD, 1), (k, HD, H * HD, H * HD + HD, 1), (vt, S_ld, H * HD + HD, N, 2)],
                        c_map=(Hs, S_ld, P), rope_cos=cos, rope_sin=sin, rope_half=HD // 2, split_k=-1, mod=mod, mod_ld=3 * K,
                        mod_rpb=Hs, eps=1e-6, w_packed=packed)
        h = torch.zeros(M, F, dtype=BF16, device=dev())
        ops.skinny_gemm(x, ops.pack_skinny_weight(wgu) if packed else wgu, M=M, N=2 * F, K=K, lda=K, ldw=K, mode=2, pair_stride=F,
                        split_k=-1, segs=[(h, F, 0, F, 0)], mod=mod, mod_ld=3 * K, mod_rpb=Hs, eps=1e-6, w_packed=packed)
        got.append((q, k, vt, h))
    for a_, b_, name in zip(got[0], got[1], ("q", "k", "vt", "h")):
        assert torch.equal(a_, b_), name


@pytest.mark.parametrize("B,Hs", [(1, 50), (2, 7)])
def test_denoise_glue_equals_the_six_launches_it_replaces(ops, B, Hs):
    """kai0_denoise_glue (step seam of the denoise loop) against adarms -> cast -> f32 GEMM -> Euler -> f32 GEMM -> cast: the Euler
    state to f32 round-off (another summation order in the two dots), the next suffix embedding to one bf16 ulp; and the two
    one-sided forms (open only / close only)."""
    De, A, M = 1024, 32, B * Hs
    xs = rnd(M, De, seed=1)
    mod_all = rnd(B, 5 * De, dtype=F32, seed=2, scale=0.3)  # a wider row: the modulation is a strided view (leading dimension 5 De)
    mod = mod_all[:, De : 4 * De]
    w_out, b_out = rnd(A, De, dtype=F32, seed=3, scale=0.05), rnd(A, dtype=F32, seed=4, scale=0.1)
    w_in, b_in = rnd(De, A, dtype=F32, seed=5, scale=0.2), rnd(De, dtype=F32, seed=6, scale=0.1)
    x0 = rnd(M, A, dtype=F32, seed=7)
    dt = -0.1
    y, _ = ops.adarms(xs, mod.contiguous(), Hs, 1e-6)
    v = ops.linear_f32(ops.cast(y, F32), w_out, b_out)
    x_ref = x0.clone()
    ops.euler_step_(x_ref, v, dt)
    xs_ref = ops.cast(ops.linear_f32(x_ref, w_in, b_in), BF16)
    x_t, xs_next = x0.clone(), torch.full((M, De), float("nan"), dtype=BF16, device=dev())
    ops.denoise_glue(x_t, xs=xs, mod=mod, mod_ld=5 * De, rows_per_batch=Hs, eps=1e-6, w_out=w_out, b_out=b_out, dt=dt, w_in=w_in,
                     b_in=b_in, xs_next=xs_next)
    assert torch.allclose(x_t, x_ref, rtol=1e-5, atol=1e-5), float((x_t - x_ref).abs().max())
    assert rel_err(xs_next, xs_ref) < 2e-3 and float((xs_next.float() - xs_ref.float()).abs().max()) <= 2 ** -7 * float(xs_ref.float().abs().max())
    # close only: x_t updated, nothing else written; open only: x_t untouched
    x_c = x0.clone()
    ops.denoise_glue(x_c, xs=xs, mod=mod, mod_ld=5 * De, rows_per_batch=Hs, eps=1e-6, w_out=w_out, b_out=b_out, dt=dt)
    assert torch.equal(x_c, x_t)
    x_o, xs_o = x_ref.clone(), torch.empty((M, De), dtype=BF16, device=dev())
    ops.denoise_glue(x_o, w_in=w_in, b_in=b_in, xs_next=xs_o)
    assert torch.equal(x_o, x_ref) and rel_err(xs_o, xs_ref) < 2e-3


@pytest.mark.parametrize("HD,H", [(256, 8), (64, 2)])
def test_pack_rows_equals_the_separate_rope_copy_and_fill_launches(ops, HD, H):
    """kai0_pack_rows (a layer's joint-attention assembly in one launch: two segments' q / k rotated into the joint buffers, v copied,
    padding rows cleared, and the inverse rotation on the way back into column slices) == kai0_rope_copy / kai0_copy_rows, bit for bit."""
    B, lens = 3, (37, 5)
    S = sum(lens)
    S_ld = S + 6
    pos = torch.randint(0, 900, (B, S), dtype=torch.int32, device=dev())
    inv = (1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.float32) / HD))).to(dev())
    qs = [rnd(B * L, H * HD, seed=10 + i) for i, L in enumerate(lens)]
    ks = [rnd(B * L, HD, seed=20 + i) for i, L in enumerate(lens)]
    vs = [rnd(B * L, HD, seed=30 + i) for i, L in enumerate(lens)]
    nan = float("nan")
    ref = [torch.full((B, S_ld, w), nan, dtype=BF16, device=dev()) for w in (H * HD, HD, HD)]
    got = [t.clone() for t in ref]
    for t in ref:
        t[:, S:].zero_()
    parts, r0 = [], 0
    for i, L in enumerate(lens):
        ops.rope_copy(qs[i], ref[0], pos, inv, B, L, H, HD, src=(L * H * HD, H * HD, 0), dst=(S_ld * H * HD, H * HD, r0), pos_bs=S, pos_off=r0)
        ops.rope_copy(ks[i], ref[1], pos, inv, B, L, 1, HD, src=(L * HD, HD, 0), dst=(S_ld * HD, HD, r0), pos_bs=S, pos_off=r0)
        ops._copy_rows(vs[i], ref[2], B, L, HD, L * HD, 0, HD, S_ld * HD, r0, HD)
        parts.append((qs[i], 0, got[0], r0 * H * HD, (L * H * HD, H * HD), (S_ld * H * HD, H * HD), B, L, H * HD, 1, r0))
        parts.append((ks[i], 0, got[1], r0 * HD, (L * HD, HD), (S_ld * HD, HD), B, L, HD, 1, r0))
        parts.append((vs[i], 0, got[2], r0 * HD, (L * HD, HD), (S_ld * HD, HD), B, L, HD, 0, 0))
        r0 += L
    for t, w in zip(got, (H * HD, HD, HD)):
        parts.append((None, 0, t, S * w, (0, 0), (S_ld * w, w), B, S_ld - S, w, 3, 0))
    ops.pack_rows(parts, HD, pos, S, inv)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # the way back: inverse rotation while a segment's rows are gathered into column slices of one buffer
    W3 = (H + 2) * HD
    r0 = lens[0]
    L = lens[1]
    ref_b = torch.full((B * L, W3), nan, dtype=BF16, device=dev())
    got_b = ref_b.clone()
    ops.rope_copy(ref[0], ref_b, pos, inv, B, L, H, HD, src=(S_ld * H * HD, H * HD, r0), dst=(L * W3, W3, 0), pos_bs=S, pos_off=r0, inverse=True)
    ops.rope_copy(ref[1], ref_b[:, H * HD :], pos, inv, B, L, 1, HD, src=(S_ld * HD, HD, r0), dst=(L * W3, W3, 0), pos_bs=S, pos_off=r0,
                  inverse=True)
    ops._copy_rows(ref[2], ref_b[:, (H + 1) * HD :], B, L, HD, S_ld * HD, r0, HD, L * W3, 0, W3)
    ops.pack_rows([(got[0], r0 * H * HD, got_b, 0, (S_ld * H * HD, H * HD), (L * W3, W3), B, L, H * HD, 2, r0),
                   (got[1], r0 * HD, got_b[:, H * HD :], 0, (S_ld * HD, HD), (L * W3, W3), B, L, HD, 2, r0),
                   (got[2], r0 * HD, got_b[:, (H + 1) * HD :], 0, (S_ld * HD, HD), (L * W3, W3), B, L, HD, 0, 0)], HD, pos, S, inv)
    assert torch.equal(ref_b, got_b)


def test_rope_two_tensors_in_one_launch(ops):
    """kai0_rope_inplace2 (q and k of a layer, shared positions) == two kai0_rope_inplace calls, bit for bit."""
    B, S, S_ld, H, HD = 2, 37, 40, 8, 256
    q, k = rnd(B, S_ld, H * HD, seed=1), rnd(B, S_ld, HD, seed=2)
    pos = torch.randint(0, 900, (B, S), dtype=torch.int32, device=dev())
    inv = (1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.float32) / HD))).to(dev())
    q1, k1 = q.clone(), k.clone()
    ops.rope_(q1, pos, inv, B, S, S_ld, 0, H, HD)
    ops.rope_(k1, pos, inv, B, S, S_ld, 0, 1, HD)
    q2, k2 = q.clone(), k.clone()
    ops.rope2_(q2, H, k2, 1, pos, inv, B, S, S_ld, 0, HD)
    assert torch.equal(q1, q2) and torch.equal(k1, k2) and not torch.equal(q1, q)


def test_device_resize_on_the_gpu_is_pillow_bit_for_bit():
    """kai0_amd.device_resize on cuda:0 (the serve path's default when the model sits on a GPU) against Pillow on the host: three
    480 x 640 camera frames -> 224 x 224 with padding, and an upscale."""
    import numpy as np

    from kai0_amd import device_resize, image_tools

    rng = np.random.default_rng(7)
    for shape, (H, W) in (((3, 480, 640, 3), (224, 224)), ((1, 100, 120, 3), (224, 224)), ((2, 720, 1280, 3), (224, 224))):
        frames = rng.integers(0, 256, shape, dtype=np.uint8)
        got = device_resize.resize_with_pad_u8(torch.from_numpy(frames).to(dev()), H, W)
        assert got.is_cuda and got.dtype == torch.uint8
        assert np.array_equal(got.cpu().numpy(), image_tools.resize_with_pad(frames, H, W))
