"""Parity at BASELINE.json's REAL DEPTH (VERDICT r2 "next round" #1): the HIP path against the CPU oracle — not against itself —
on the full pi0.5: 18 joint Gemma-2B / 300M-expert layers + 27 SigLIP layers, full widths, three 224^2 cameras, 200 prompt
tokens with padding, 50 x 32 actions (S = 1018, P = 968), B = 1, identical weights (tests/fulldepth.py), inputs, noise, time:

  * the loss tensor (pi0_pytorch.py:316-373) vs the fp32 oracle and vs the bf16-choreography oracle;
  * the 10-step action chunk (pi0_pytorch.py:375-461; prefix pass into the static cache, 10 Euler steps on the weight-streaming
    kernels, hipGraph replay) vs both oracles: rel-L2 and max|d|;
  * every parameter gradient of mean(loss) vs ONE fp32 oracle backward.

Tolerances: loss rel-L2 <= 1e-2; chunk rel-L2 <= 3e-3 and max|d| <= 2e-2 vs the bf16 oracle, rel-L2 <= 1e-2 vs
the fp32 oracle.  Gradients (round 4): BASELINE.md states no tolerance, so the bar is what the REFERENCE'S OWN bf16 choreography
loses against fp32 — the bf16 oracle is run backward too and, per parameter, HIP-vs-fp32 must stay within 1.5 x (bf16-oracle-vs-fp32)
+ 2e-3 (and under the flat 5e-2 of the earlier rounds); both columns are written out.  The measured numbers go to
gpurun_out/parity_fulldepth.txt (committed as profiles/parity_r06_fulldepth.txt; earlier rounds: parity_r03 / r04 / r05*.txt) and gpurun_out/grad_table_fulldepth.txt.  Host cost on the 64-core
bench box: fp32 forward + backward ~25 s, bf16 forward + backward ~10 s, fp32 chunk ~12 s."""

import os
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

F32 = torch.float32
REPORT = os.path.join("gpurun_out", "parity_fulldepth.txt")
WITH_BF16_ORACLE = os.environ.get("KAI0_FULLDEPTH_BF16", "1") != "0"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _report(line: str):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


@pytest.fixture(scope="module")
def fd():
    from fulldepth import build_hip, host_state, oracle_from_state
    from tiny import obs_to

    from oracle.pi0_oracle import synthetic_batch

    dev = torch.device("cuda:0")
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    t0 = time.time()
    model = build_hip(dev, seed=0)
    state = host_state(model)
    o32, ocfg = oracle_from_state(state, "float32")
    obf = oracle_from_state(state, "bfloat16")[0] if WITH_BF16_ORACLE else None
    del state
    obs, actions, noise, tm = synthetic_batch(ocfg, 1, seed=5)
    if os.path.exists(REPORT):
        os.remove(REPORT)
    pe = model.paligemma_with_expert
    _report("# full-depth parity, HIP path vs CPU oracle (tests/test_fulldepth_gpu.py), B = 1, identical weights / inputs / noise / time")
    _report(f"# layers: joint {pe.vlm_cfg.depth}, SigLIP {pe.siglip_cfg.num_layers}; widths {pe.vlm_cfg.width}/{pe.vlm_cfg.mlp_dim} + "
            f"{pe.exp_cfg.width}/{pe.exp_cfg.mlp_dim} + {pe.siglip_cfg.hidden_size}/{pe.siglip_cfg.intermediate_size}; "
            f"prompt valid tokens {int(obs.tokenized_prompt_mask.sum())} of {obs.tokenized_prompt.shape[1]}; "
            f"host threads {torch.get_num_threads()}; build {time.time() - t0:.1f} s")  # fmt: skip
    return dict(model=model, o32=o32, obf=obf, ocfg=ocfg, obs=obs, gobs=obs_to(obs, dev), actions=actions, noise=noise, time=tm, dev=dev)


def test_fulldepth_is_the_baseline_architecture(fd):
    pe = fd["model"].paligemma_with_expert
    assert (pe.vlm_cfg.depth, pe.exp_cfg.depth, pe.siglip_cfg.num_layers) == (18, 18, 27)
    assert (pe.vlm_cfg.width, pe.vlm_cfg.mlp_dim, pe.exp_cfg.width, pe.exp_cfg.mlp_dim) == (2048, 16384, 1024, 4096)
    assert len(fd["o32"].paligemma_with_expert.paligemma.language_model.layers) == 18
    assert not bool(fd["obs"].tokenized_prompt_mask.all())  # padded prompt: the mask logic is exercised


def test_fulldepth_loss_and_gradients_match_oracle(fd):
    m, o32, d = fd["model"], fd["o32"], fd["dev"]
    args = (fd["gobs"], fd["actions"].to(d))
    kw = dict(noise=fd["noise"].to(d), time=fd["time"].to(d))
    m.zero_grad(set_to_none=True)
    loss = m(*args, **kw)
    loss.mean().backward()
    t0 = time.time()
    o32.zero_grad(set_to_none=True)
    ref32 = o32(fd["obs"], fd["actions"], fd["noise"], fd["time"])
    ref32.mean().backward()
    t32 = time.time() - t0
    r32 = rel(loss, ref32)
    line = f"loss tensor [1,50,32]: rel-L2 {r32:.3e} vs fp32 oracle (oracle fwd+bwd {t32:.1f} s)"
    rbf = None
    gbf = {}
    if fd["obf"] is not None:
        t0 = time.time()
        obf = fd["obf"]
        obf.zero_grad(set_to_none=True)
        refbf = obf(fd["obs"], fd["actions"], fd["noise"], fd["time"])
        refbf.mean().backward()  # the reference's own choreography through autograd: what IT loses against fp32, per parameter
        gbf = {n: p.grad for n, p in obf.named_parameters() if p.grad is not None}
        rbf = rel(loss, refbf)
        line += f"; {rbf:.3e} vs bf16 oracle (fwd+bwd {time.time() - t0:.1f} s); bf16 oracle vs fp32 oracle {rel(refbf, ref32):.3e}"
    _report(line)
    assert loss.shape == (1, 50, 32) and loss.dtype == F32
    assert r32 <= 1e-2 and (rbf is None or rbf <= 1e-2)

    gm = {n: p.grad for n, p in m.named_parameters()}
    table, bad = [], []
    for n, p in o32.named_parameters():
        g = p.grad
        if g is None:  # (the oracle's tied lm_head is a separate, unused tensor after to_empty: not a parameter of the HIP model)
            assert gm.get(n) is None or float(gm[n].abs().max()) == 0.0, f"{n} must not receive a gradient"
            continue
        assert gm.get(n) is not None, f"no gradient for {n}"
        if float(g.norm()) < 1e-9:
            assert float(gm[n].float().norm()) < 1e-5, n
            continue
        r = rel(gm[n], g)
        rb = rel(gbf[n], g) if n in gbf else float("nan")
        table.append((r, n, float(g.norm()), rb))
        if r > 5e-2 or (rb == rb and r > 1.5 * rb + 2e-3):
            bad.append((r, rb, n))
    table.sort(reverse=True)
    with_bf = [t for t in table if t[3] == t[3]]
    _report(f"gradients of mean(loss): {len(table)} parameters vs fp32 oracle autograd, worst rel-L2 {table[0][0]:.3e} ({table[0][1]}), "
            f"median {table[len(table) // 2][0]:.3e}")  # fmt: skip
    if with_bf:
        worst_ratio = max(with_bf, key=lambda t: t[0] / (1.5 * t[3] + 2e-3))
        _report(f"    the reference's bf16 choreography (bf16 oracle backward) vs the same fp32 gradients: worst {max(t[3] for t in with_bf):.3e}, "
                f"median {sorted(t[3] for t in with_bf)[len(with_bf) // 2]:.3e}; HIP / (1.5 x bf16-oracle + 2e-3) at most "
                f"{worst_ratio[0] / (1.5 * worst_ratio[3] + 2e-3):.2f} ({worst_ratio[1]}: HIP {worst_ratio[0]:.3e}, bf16 oracle {worst_ratio[3]:.3e})")  # fmt: skip
    _report("    HIP vs fp32 | bf16 oracle vs fp32 | |g| | parameter")
    for r, n, gn, rb in table[:12]:
        _report(f"    {r:.3e}  {rb:.3e}  |g|={gn:.3e}  {n}")
    with open(os.path.join("gpurun_out", "grad_table_fulldepth.txt"), "w") as f:
        f.write("# rel-L2 of d mean(loss) / d parameter against ONE fp32 oracle backward: HIP | bf16-choreography oracle | |g| | name\n")
        for r, n, gn, rb in table:
            f.write(f"{r:.3e}  {rb:.3e}  |g|={gn:.3e}  {n}\n")
    if fd["obf"] is not None:
        fd["obf"].zero_grad(set_to_none=True)
    o32.zero_grad(set_to_none=True)
    m.zero_grad(set_to_none=True)
    assert not bad, f"{len(bad)} gradient mismatches, worst: {sorted(bad, reverse=True)[:5]}"
    assert len(table) >= 700


def test_fulldepth_action_chunk_matches_oracle(fd):
    m, d = fd["model"], fd["dev"]
    m.eval()
    try:
        out = m.sample_actions(d, fd["gobs"], noise=fd["noise"].to(d), num_steps=10)
        t0 = time.time()
        with torch.no_grad():
            ref32 = fd["o32"].sample_actions(fd["obs"], fd["noise"], num_steps=10)
        t32 = time.time() - t0
        r32, mx32 = rel(out, ref32), float((out.cpu() - ref32).abs().max())
        line = (f"10-step action chunk [1,50,32] (|ref| max {float(ref32.abs().max()):.2f}): rel-L2 {r32:.3e}, max|d| {mx32:.3e} vs fp32 oracle "
                f"({t32:.1f} s)")  # fmt: skip
        rbf = mxbf = None
        if fd["obf"] is not None:
            t0 = time.time()
            with torch.no_grad():
                refbf = fd["obf"].sample_actions(fd["obs"], fd["noise"], num_steps=10)
            rbf, mxbf = rel(out, refbf), float((out.cpu() - refbf).abs().max())
            line += (f"; rel-L2 {rbf:.3e}, max|d| {mxbf:.3e} vs bf16 oracle ({time.time() - t0:.1f} s); bf16 oracle vs fp32 oracle "
                     f"{rel(refbf, ref32):.3e}")  # fmt: skip
        _report(line)
        assert out.shape == (1, 50, 32) and out.dtype == F32
        assert r32 <= 1e-2 and mx32 <= 2e-2
        assert rbf is None or (rbf <= 3e-3 and mxbf <= 2e-2)
        assert torch.equal(out, m.sample_actions(d, fd["gobs"], noise=fd["noise"].to(d), num_steps=10))  # replay is deterministic
    finally:
        m.train()


def test_fulldepth_second_case_short_ragged_prompts_masked_camera(fd):
    """A second full-depth case (one seed, B = 1, one prompt length): another seed, B = 2, the 17 / 41-token prompts
    kai0's tasks have (whole key tiles of padding: the attention kernels' dead-tile skip), and sample 1's right-wrist camera masked out
    (256 hidden keys in the middle of the prefix).  Loss tensor and 10-step chunk against the fp32 oracle, BASELINE.md section 4 tolerances."""
    from tiny import obs_to

    from oracle.pi0_oracle import synthetic_batch

    m, o32, d = fd["model"], fd["o32"], fd["dev"]
    obs, actions, noise, tm = synthetic_batch(fd["ocfg"], 2, seed=9)
    L = obs.tokenized_prompt.shape[1]
    obs.tokenized_prompt_mask = torch.arange(L)[None, :] < torch.tensor([17, 41])[:, None]
    last = list(obs.image_masks)[-1]
    obs.image_masks[last] = torch.tensor([True, False])
    gobs = obs_to(obs, d)
    with torch.no_grad():
        loss = m(gobs, actions.to(d), noise=noise.to(d), time=tm.to(d))
        t0 = time.time()
        ref = o32(obs, actions, noise, tm)
        t_l = time.time() - t0
    r_l = rel(loss, ref)
    m.eval()
    try:
        out = m.sample_actions(d, gobs, noise=noise.to(d), num_steps=10)
        t0 = time.time()
        with torch.no_grad():
            refc = o32.sample_actions(obs, noise, num_steps=10)
        t_c = time.time() - t0
    finally:
        m.train()
    r_c, mx = rel(out, refc), float((out.cpu() - refc).abs().max())
    per = [rel(out[i], refc[i]) for i in range(2)]
    _report(f"second case (seed 9, B = 2, prompts of 17 / 41 valid tokens, sample 1 without its third camera): loss [2,50,32] rel-L2 {r_l:.3e} "
            f"(oracle {t_l:.1f} s); 10-step chunk rel-L2 {r_c:.3e} (per sample {per[0]:.3e} / {per[1]:.3e}), max|d| {mx:.3e} vs fp32 oracle ({t_c:.1f} s)")  # fmt: skip
    assert loss.shape == (2, 50, 32) and r_l <= 1e-2
    assert r_c <= 1e-2 and mx <= 2e-2 and max(per) <= 1e-2


def test_fulldepth_advantage_estimator_matches_oracle(fd):
    """The Stage-Advantage estimator (pi0_pytorch.py:464-644) at full depth and width against the fp32 oracle, B = 1: six images (two
    timesteps x three cameras: prefix 6 x 256 + 200 = 1736 tokens, the longest attention shape of the path), the trunk weights of
    the pi0.5 fixture + a seeded value head; the weighted loss row [1, 50], its value term, and sample_values.  Tolerance: the loss
    bar of BASELINE.md section 4 (rel-L2 <= 1e-2); the value (a tanh output in [-1, 1]) to 2e-2 absolute."""
    import gc

    from fulldepth import host_state
    from kai0_amd.config import AdvantageEstimatorConfig
    from kai0_amd.model import AdvantageEstimator
    from oracle import pi0_oracle as O

    dev, base = fd["dev"], fd["model"]
    with torch.device(dev):
        m = AdvantageEstimator(AdvantageEstimatorConfig(vocab_size=2048, loss_value_weight=0.7, loss_action_weight=1.3))
    g = torch.Generator(device=dev).manual_seed(11)
    with torch.no_grad():
        m.load_state_dict(base.state_dict(), strict=False)
        for k, p in m.named_parameters():
            if k.startswith("value_head."):
                p.copy_(torch.randn(p.shape, generator=g, device=dev) * (0.05 if p.dim() == 2 else 0.02))
    m.train_augmentation = False
    state = host_state(m)
    cfg = O.OracleConfig(dtype="float32", vocab_size=2048)
    with torch.device("meta"):
        est = O.OracleAdvantageEstimator(cfg, loss_value_weight=0.7, loss_action_weight=1.3)
    est.to_empty(device="cpu")
    with torch.no_grad():
        est.load_state_dict(state, strict=True)
        for mod in est.modules():
            if isinstance(getattr(mod, "inv_freq", None), torch.Tensor):
                mod.inv_freq = O.rope_inv_freq(mod.inv_freq.numel() * 2).to(torch.bfloat16).float()
            if isinstance(mod, O.SiglipVisionEmbeddings):
                mod.position_ids = torch.arange(mod.num_patches).expand((1, -1))
    del state
    obs = fd["obs"]
    gen = torch.Generator().manual_seed(12)
    images = dict(obs.images)
    for name in ("base_-1_rgb", "left_wrist_-1_rgb", "right_wrist_-1_rgb"):
        images[name] = torch.rand(1, 3, 224, 224, generator=gen) * 2 - 1
    obs6 = O.SimpleObs(images=images, image_masks={k: torch.ones(1, dtype=torch.bool) for k in images}, state=obs.state,
                       tokenized_prompt=obs.tokenized_prompt, tokenized_prompt_mask=obs.tokenized_prompt_mask, token_ar_mask=None,
                       token_loss_mask=None, progress=torch.tensor([0.4]))  # fmt: skip
    from tiny import obs_to

    gobs = obs_to(obs6, dev)
    gobs.progress = obs6.progress.to(dev)
    m.train()
    loss, aux = m(gobs, fd["actions"].to(dev), noise=fd["noise"].to(dev), time=fd["time"].to(dev), return_loss_dict=True)
    m.eval()
    val = m.sample_values(dev, gobs, noise=fd["noise"].to(dev), time=fd["time"].to(dev))
    t0 = time.time()
    with torch.no_grad():
        ref, raux = est(obs6, fd["actions"], fd["noise"], fd["time"], return_loss_dict=True)
        rval = est.sample_values(obs6, fd["noise"], fd["time"])
    r = rel(loss, ref)
    dv = float((val.cpu() - rval).abs().max())
    _report(f"AdvantageEstimator (six images, S = 1786): loss row [1,50] rel-L2 {r:.3e} vs fp32 oracle; value-loss term {float(aux['loss_value']):.5f} vs "
            f"{float(raux['loss_value']):.5f}; sample_values {float(val):+.5f} vs {float(rval):+.5f} (|d| {dv:.2e}); oracle {time.time() - t0:.1f} s")  # fmt: skip
    assert loss.shape == (1, 50) and r <= 1e-2
    assert abs(float(aux["loss_value"]) - float(raux["loss_value"])) <= 2e-2 * max(1.0, abs(float(raux["loss_value"])))
    assert dv <= 2e-2
    del m, est
    gc.collect()
    torch.cuda.empty_cache()
