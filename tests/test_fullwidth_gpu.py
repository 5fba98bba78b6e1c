"""Parity at BASELINE.json's own widths and sequence: the HIP path against the CPU oracle — not against
itself — on a full-width, depth-reduced pi0.5 (tests/fullwidth.py: Gemma-2B 2048/16384/8x256 + 300M expert, SigLIP 1152/4304,
three 224^2 cameras, 200 prompt tokens with padding, 50 x 32 actions: S = 1018, P = 968), and every kernel that carries the
training step in the exact launch configuration the real model uses:

  * joint attention forward + backward at H = 8, HD = 256, P = 968, Hs = 50 (16 key tiles, 64 row blocks, padding);
  * the GeGLU MLP at M = 4352, D = 2048, F = 16384 (256x256 tiles, both 64-row epilogue halves, act 2 forward / act 3 backward);
  * the softmax-backward epilogue (act 4) at 8144 x 1024 x 256, batched;
  * the transposed-operand weight-gradient GEMM (ring schedule) at 16384 x 2048 x 30976, on sampled output rows.

Tolerances (bf16 path vs fp32 oracle / fp32 torch reference of the same op): loss tensor rel-L2 <= 1e-2; 10-step action chunk
rel-L2 <= 3e-3 and max|d| <= 2e-2 vs the bf16-choreography oracle, rel-L2 <= 1e-2 vs the fp32 oracle; parameter gradients
rel-L2 <= 3e-2 vs fp32 autograd (about twice the worst measured on MI355X, see gpurun_out/grad_table_fullwidth.txt)."""

import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

BF16, F32 = torch.bfloat16, torch.float32
N_JOINT, N_SIG = 2, 2


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def dev():
    return torch.device("cuda:0")


def rnd(*shape, dtype=BF16, seed=0, scale=1.0):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=dev()) * scale).to(dtype)


# ===================================================================================== model level vs the oracle
@pytest.fixture(scope="module")
def fw():
    from fullwidth import build_hip, build_oracle
    from tiny import obs_to

    from oracle.pi0_oracle import synthetic_batch

    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    oracle, ocfg = build_oracle(N_JOINT, N_SIG)
    model = build_hip(oracle, N_JOINT, N_SIG, dev())
    obs, actions, noise, time = synthetic_batch(ocfg, 2, seed=3)
    o32 = copy.deepcopy(oracle)
    o32.paligemma_with_expert.to_bfloat16_for_selected_params("float32")
    return dict(model=model, oracle=oracle, o32=o32, ocfg=ocfg, obs=obs, gobs=obs_to(obs, dev()), actions=actions, noise=noise,
                time=time)  # fmt: skip


def test_fullwidth_shapes_are_the_baseline_ones(fw):
    m = fw["model"]
    pe = m.paligemma_with_expert
    assert (pe.vlm_cfg.width, pe.vlm_cfg.mlp_dim, pe.vlm_cfg.num_heads, pe.vlm_cfg.head_dim) == (2048, 16384, 8, 256)
    assert (pe.exp_cfg.width, pe.exp_cfg.mlp_dim) == (1024, 4096)
    assert (pe.siglip_cfg.hidden_size, pe.siglip_cfg.intermediate_size, pe.siglip_cfg.num_heads) == (1152, 4304, 16)
    assert fw["obs"].tokenized_prompt.shape == (2, 200) and fw["actions"].shape == (2, 50, 32)
    assert not bool(fw["obs"].tokenized_prompt_mask.all())  # padded prompts: the mask logic is exercised


def test_fullwidth_loss_matches_oracle(fw):
    m, d = fw["model"], dev()
    with torch.no_grad():
        loss = m(fw["gobs"], fw["actions"].to(d), noise=fw["noise"].to(d), time=fw["time"].to(d))
        ref = fw["oracle"](fw["obs"], fw["actions"], fw["noise"], fw["time"])
        ref32 = fw["o32"](fw["obs"], fw["actions"], fw["noise"], fw["time"])
    r, r32 = rel(loss, ref), rel(loss, ref32)
    print(f"full-width loss: rel-L2 {r:.3e} vs bf16 oracle, {r32:.3e} vs fp32 oracle (bf16 vs fp32 oracle {rel(ref, ref32):.3e})")
    assert loss.shape == (2, 50, 32) and loss.dtype == F32
    assert r <= 1e-2 and r32 <= 1e-2


def test_fullwidth_gradients_match_fp32_oracle_autograd(fw):
    m, o32, d = fw["model"], fw["o32"], dev()
    m.zero_grad(set_to_none=True)
    m(fw["gobs"], fw["actions"].to(d), noise=fw["noise"].to(d), time=fw["time"].to(d)).mean().backward()
    o32.zero_grad(set_to_none=True)
    o32(fw["obs"], fw["actions"], fw["noise"], fw["time"]).mean().backward()
    gm = {n: p.grad for n, p in m.named_parameters()}
    table, bad, checked = [], [], 0
    for n, p in o32.named_parameters():
        g = p.grad
        if g is None:
            assert gm[n] is None or float(gm[n].abs().max()) == 0.0, f"{n} must not receive a gradient"
            continue
        assert gm[n] is not None, f"no gradient for {n}"
        if float(g.norm()) < 1e-9:
            assert float(gm[n].float().norm()) < 1e-5, n
            continue
        r = rel(gm[n], g)
        table.append((r, n, float(g.norm())))
        checked += 1
        if r > 3e-2:
            bad.append((r, n))
    table.sort(reverse=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/grad_table_fullwidth.txt", "w") as f:
        for r, n, gn in table:
            f.write(f"{r:.3e}  |g|={gn:.3e}  {n}\n")
    print(f"full width: checked {checked} gradients, worst rel-L2 {table[0][0]:.3e} ({table[0][1]})")
    assert not bad, f"{len(bad)} gradient mismatches, worst: {sorted(bad, reverse=True)[:5]}"
    assert checked >= 70


@pytest.mark.parametrize("batch", [2, 1])
def test_fullwidth_action_chunk_matches_oracle(fw, batch):
    """B = 2 and the B = 1 latency configuration: prefix pass into the static cache, 10 Euler steps on the weight-streaming
    kernels, replayed from the hipGraph."""
    from test_fullsize_gpu import _take

    m, d = fw["model"], dev()
    m.eval()
    try:
        gobs, obs, noise = fw["gobs"], fw["obs"], fw["noise"]
        if batch == 1:
            gobs, obs, noise = _take(gobs, 1), _take(obs, 1), noise[1:2]
        out = m.sample_actions(d, gobs, noise=noise.to(d), num_steps=10)
        with torch.no_grad():
            ref = fw["oracle"].sample_actions(obs, noise, num_steps=10)
            ref32 = fw["o32"].sample_actions(obs, noise, num_steps=10)
        r, r32 = rel(out, ref), rel(out, ref32)
        mx = float((out.cpu() - ref).abs().max())
        print(f"full-width chunk B={batch}: rel-L2 {r:.3e} (max|d| {mx:.3e}) vs bf16 oracle, {r32:.3e} vs fp32 oracle")
        assert out.shape == (batch, 50, 32) and out.dtype == F32
        assert r <= 3e-3 and mx <= 2e-2 and r32 <= 1e-2
        assert torch.equal(out, m.sample_actions(d, gobs, noise=noise.to(d), num_steps=10))  # replay is deterministic
    finally:
        m.train()


# ===================================================================================== kernels in the launch shapes that run
def test_joint_attention_production_shape():
    """H = 8, HD = 256, P = 968, Hs = 50 (S = 1018: 16 key tiles, 64 row blocks per sample), padded prompt tokens and a masked
    camera, forward and backward vs an fp32 reference of RoPE + prefix-LM masked MQA."""
    from test_kernels_gpu import _mqa_ref

    from kai0_amd import ops
    from kai0_amd.model import build_mask_codes

    B, H, HD, P, Hs = 2, 8, 256, 968, 50
    S = P + Hs
    pad = torch.ones((B, S), dtype=torch.bool, device=dev())
    pad[0, 768 + 90 : P] = False  # prompt of 90 tokens
    pad[1, 512:768] = False  # third camera masked out
    pad[1, 768 + 120 : P] = False
    att = torch.zeros((B, S), dtype=torch.bool, device=dev())
    att[:, P] = True
    qcode, kcode, pos = build_mask_codes(pad, att)
    inv = (1.0 / (10000.0 ** (torch.arange(0, HD, 2, dtype=torch.int64).float() / HD))).to(dev())
    flat = []
    for L, sd in ((P, 1), (Hs, 2)):
        flat += [rnd(B * L, H * HD, seed=sd).requires_grad_(True), rnd(B * L, HD, seed=sd + 10).requires_grad_(True),
                 rnd(B * L, HD, seed=sd + 20).requires_grad_(True)]  # fmt: skip
    outs = ops.joint_attention(pos, qcode, kcode, inv, H, HD, (P, Hs), flat)
    dfull = torch.cat([rnd(B, P, H * HD, seed=30), rnd(B, Hs, H * HD, seed=31)], 1) * pad[:, :, None]
    douts = [dfull[:, :P].reshape(B * P, -1).contiguous(), dfull[:, P:].reshape(B * Hs, -1).contiguous()]
    torch.autograd.backward(list(outs), douts)
    refs = [t.detach().float().requires_grad_(True) for t in flat]
    q = torch.cat([refs[0].view(B, P, -1), refs[3].view(B, Hs, -1)], 1)
    k = torch.cat([refs[1].view(B, P, -1), refs[4].view(B, Hs, -1)], 1)
    v = torch.cat([refs[2].view(B, P, -1), refs[5].view(B, Hs, -1)], 1)
    o = _mqa_ref(q, k, v, pos, inv, qcode, kcode, H, HD)
    o.backward(dfull.float())
    got = torch.cat([outs[0].view(B, P, -1), outs[1].view(B, Hs, -1)], 1)
    e = rel(got[pad], o[pad])
    print(f"joint attention S=1018: out rel-L2 {e:.3e}")
    assert e < 1e-2
    for n, t, r in zip(["dq_p", "dk_p", "dv_p", "dq_s", "dk_s", "dv_s"], flat, refs):
        e = rel(t.grad, r.grad)
        print(f"  {n}: rel-L2 {e:.3e}")
        assert e < 2e-2, f"{n}: rel-L2 {e:.3e}"
    # prefix keys never see the suffix (prefix-LM): dk / dv of the prefix rows get nothing from suffix-only paths and the
    # padded rows get exactly nothing
    vk = flat[1].grad.view(B, P, HD)
    assert float(vk[0, 768 + 90 :].abs().max()) == 0.0 and float(vk[1, 512:768].abs().max()) == 0.0


def test_geglu_mlp_production_shape():
    """Gemma-2B MLP, 4352 token rows: the up-projection GEMM with the GeGLU forward epilogue (act 2) and the down-projection
    dgrad with the GeGLU backward epilogue (act 3) both take the 256x256 tile (1088 tiles) as in the B = 32 step."""
    from kai0_amd import ops

    M, D, Fd = 4352, 2048, 16384
    x = rnd(M, D, seed=1).requires_grad_(True)
    wg, wu = (rnd(Fd, D, seed=s, scale=0.03).requires_grad_(True) for s in (2, 3))
    wd = rnd(D, Fd, seed=4, scale=0.02).requires_grad_(True)
    res = rnd(M, D, seed=5).requires_grad_(True)
    dy = rnd(M, D, seed=6)
    out = ops.geglu_mlp(x, wg, wu, wd, res)
    out.backward(dy)
    ref_in = [t.detach().float().requires_grad_(True) for t in (x, wg, wu, wd, res)]
    xr, gr, ur, dr, rr = ref_in
    ref = (torch.nn.functional.gelu(xr @ gr.t(), approximate="tanh") * (xr @ ur.t())) @ dr.t() + rr
    ref.backward(dy.float())
    e = rel(out, ref)
    print(f"geglu_mlp 4352x2048x16384: out rel-L2 {e:.3e}")
    assert e < 6e-3
    for n, t, r in zip(("dx", "dwg", "dwu", "dwd"), (x, wg, wu, wd), ref_in):
        e = rel(t.grad, r.grad)
        print(f"  {n}: rel-L2 {e:.3e}")
        assert e < 1.5e-2, f"{n}: {e:.3e}"
    assert torch.equal(res.grad, dy)


def test_softmax_backward_epilogue_production_shape():
    """act 4 in the shape of the joint attention backward: dS[b] [8144, 1024] = softmax'(dO[b] [8144, 256] V[b]^T), batch 2."""
    from kai0_amd import ops

    B, M, S, HD = 2, 8144, 1024, 256
    g = torch.Generator(device="cuda:0").manual_seed(0)
    probs = torch.softmax(torch.randn(B, M, S, generator=g, device=dev()) * 2, -1).to(BF16)
    v, do = rnd(B, S, HD, seed=1), rnd(B, M, HD, seed=2)
    o = torch.bmm(probs.float(), v.float()).to(BF16)
    scale = HD**-0.5
    dsum = ops.rowdot(do, o, HD)
    ds = torch.empty_like(probs)
    ops.gemm(do, v, ds, M=M, N=S, K=HD, lda=HD, ldb=HD, ldc=S, batch=B, sA=(M * HD, 0), sB=(S * HD, 0), sC=(M * S, 0), act=4,
             aux1=probs, rowvec=dsum, rv=(M, 0, 1), scale=scale)  # fmt: skip
    dp = torch.bmm(do.float(), v.float().transpose(1, 2))
    ref = probs.float() * (dp - (dp * probs.float()).sum(-1, keepdim=True)) * scale
    e = rel(ds, ref)
    print(f"act 4 at 8144x1024x256: rel-L2 {e:.3e}")
    assert e < 6e-3


def test_wgrad_ring_gemm_production_shape():
    """The transposed-operand (TN) weight-gradient GEMM of the Gemma-2B MLP at B = 32: dW[16384, 2048] = dG[30976, 16384]^T
    X[30976, 2048] on the ring schedule; checked on 96 sampled output rows against fp32."""
    from kai0_amd import ops

    M, N, K = 16384, 2048, 30976
    dg, x = rnd(K, M, seed=1), rnd(K, N, seed=2)
    dw = torch.empty((M, N), dtype=BF16, device=dev())
    ops.gemm(dg, x, dw, M=M, N=N, K=K, a_kc=False, b_kc=False, lda=M, ldb=N, ldc=N, split_k=ops.pick_split_k_wgrad(M, N, K))
    rows = torch.cat([torch.arange(0, 32), torch.arange(8000, 8032), torch.arange(M - 32, M)]).to(dev())
    ref = dg[:, rows].float().t() @ x.float()
    e = rel(dw[rows], ref)
    print(f"TN ring 16384x2048x30976: rel-L2 {e:.3e} on {rows.numel()} rows")
    assert e < 4e-3
    assert torch.isfinite(dw.float()).all()


def test_fullwidth_production_denoise_stack_agrees_with_the_generic_path(fw):
    """At the real widths the chunk comes from the production stack (in-block weight-streaming kernels, adaRMS folded into per-step
    weights, one-launch step seams, two-launch decode attention).  The generic per-layer path (plain GEMMs, adaRMS kernels, RoPE /
    attention / Euler launches of their own — what other shapes run) must give the same chunk within the bf16 path's round-off: other
    rounding points (weights rounded after the (1 + scale) factor instead of activations after the norm), inside the chunk tolerance.
    Also: gate GEMM + up GEMM instead of the pair GEMM is bit-identical; the norms behind split-K Linears as launches of their own
    differ by the summation order of the row statistics only."""
    from test_fullsize_gpu import _take

    from kai0_amd import ops
    from kai0_amd.infer import InferenceEngine

    m, d = fw["model"], dev()
    m.eval()
    try:
        gobs, noise = _take(fw["gobs"], 1), fw["noise"][1:2].to(d)
        m.invalidate_inference_engine()
        ref = m.sample_actions(d, gobs, noise=noise, num_steps=10)
        eng = m._engine
        assert eng.fast and eng.fuse_norm and eng._fold_cache
        InferenceEngine.force_generic = True
        try:
            m.invalidate_inference_engine()
            generic = m.sample_actions(d, gobs, noise=noise, num_steps=10)
            assert not m._engine.fast
        finally:
            InferenceEngine.force_generic = False
        print(f"chunk: production stack vs generic per-layer path: rel-L2 {rel(ref, generic):.3e}")
        assert rel(ref, generic) < 3e-3, rel(ref, generic)
        # the prefix attention as four key ranges + merge (default) against logits GEMM + softmax + P V GEMM: within round-off
        assert eng.key_split
        InferenceEngine.key_split = False
        try:
            m.invalidate_inference_engine()
            three = m.sample_actions(d, gobs, noise=noise, num_steps=10)
        finally:
            InferenceEngine.key_split = True
        print(f"chunk: key-split prefix attention vs logits GEMM + softmax + P V GEMM: rel-L2 {rel(ref, three):.3e}")
        assert rel(ref, three) < 3e-3, rel(ref, three)
        old = ops.set_geglu_pair(False)
        try:
            m.invalidate_inference_engine()
            assert torch.equal(m.sample_actions(d, gobs, noise=noise, num_steps=10), ref)
        finally:
            ops.set_geglu_pair(old)
        InferenceEngine.fuse_split_norm = False
        try:
            m.invalidate_inference_engine()
            sep = m.sample_actions(d, gobs, noise=noise, num_steps=10)
            assert not m._engine.fuse_norm
        finally:
            InferenceEngine.fuse_split_norm = True
        print(f"chunk with separate norm launches vs fused: rel-L2 {rel(sep, ref):.3e}")
        assert rel(sep, ref) < 2e-3, rel(sep, ref)
    finally:
        m.invalidate_inference_engine()
        m.train()


def test_fullwidth_engine_notices_a_weight_edit_behind_autograd(fw):
    """At the real widths the denoise loop reads only packed / folded COPIES of the expert's projections: after `w.data.mul_()` (no new
    storage, no version bump) the next chunk is still the old one — and is flagged by the content stamp; the call after it runs on a
    rebuilt engine."""
    from test_fullsize_gpu import _take

    m, d = fw["model"], dev()
    w = m.paligemma_with_expert.gemma_expert.model.layers[0].mlp.down_proj.weight
    m.eval()
    try:
        gobs, noise = _take(fw["gobs"], 1), fw["noise"][1:2].to(d)
        m.invalidate_inference_engine()
        before = m.sample_actions(d, gobs, noise=noise, num_steps=10)
        eng = m._engine
        torch.cuda.synchronize()
        assert eng.fast and not m.inference_is_stale()
        saved = w.data.clone()
        w.data.mul_(1.25)
        stale_chunk = m.sample_actions(d, gobs, noise=noise, num_steps=10)
        torch.cuda.synchronize()
        assert m._engine is eng and torch.equal(stale_chunk, before) and m.inference_is_stale()
        after = m.sample_actions(d, gobs, noise=noise, num_steps=10)
        torch.cuda.synchronize()
        assert m._engine is not eng and not torch.equal(after, before) and not m.inference_is_stale()
        w.data.copy_(saved)
    finally:
        m.invalidate_inference_engine()
        m.train()
