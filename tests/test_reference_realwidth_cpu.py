"""The oracle against the reference's own layers AT THE WIDTHS THE KERNELS RUN AT.

tests/test_reference_blocks_cpu.py pins the oracle to reference-executed vectors at toy widths (width 48-64, head_dim 16).
Here the reference's definitions are lifted from /root/reference when the test runs (tests/reflift.py; the test skips where
the reference is absent, i.e. on the GPU box) and executed next to the oracle at the real shapes of BASELINE.json's model:

  * a whole action-expert GemmaDecoderLayer: width 1024, 8 query heads / 1 KV head of head_dim 256, MLP 4096, adaRMS
    conditioning, cached prefix K/V (modeling_gemma.py:344-384, 282-329);
  * a whole Gemma-2B prefix GemmaDecoderLayer: width 2048, MLP 16384;
  * a whole SiglipEncoderLayer: hidden 1152, 16 heads of head_dim 72, MLP 4304, 256 patch tokens (modeling_siglip.py:435-480);
  * the JOINT prefix + expert forward of PaliGemmaWithExpertModel (gemma_pytorch.py:126-279) with both towers at real width.

Same torch ops in the same order => the bf16 results must be bit-identical."""

import os
import sys
import types

import pytest
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import reflift  # noqa: E402

from oracle import pi0_oracle as O  # noqa: E402

pytestmark = pytest.mark.skipif(not reflift.available(), reason="needs /root/reference (build container only)")
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ref():
    gem = reflift.lift(reflift.GEMMA_PY, ["GemmaRMSNorm", "GemmaMLP", "rotate_half", "apply_rotary_pos_emb", "repeat_kv",
                                          "_gated_residual", "eager_attention_forward", "GemmaAttention", "GemmaDecoderLayer"],
                       reflift.base_ns())  # fmt: skip
    sig = reflift.lift(reflift.SIGLIP_PY, ["eager_attention_forward", "SiglipAttention", "SiglipMLP", "SiglipEncoderLayer"],
                       reflift.base_ns())  # fmt: skip
    pi0 = reflift.lift(reflift.PI0_PY, ["make_att_2d_masks"], reflift.base_ns())
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    return types.SimpleNamespace(gem=gem, sig=sig, pi0=pi0)


def _rnd(g, *shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def _set_params(g, mod, scale):
    for n, p in mod.named_parameters():
        p.data = _rnd(g, *p.shape, scale=(scale if p.dim() >= 2 else 0.1)).to(p.dtype)


def _selected_bf16(mod):
    """gemma_pytorch.py:63-83 on a single layer: everything bf16, the norms (incl. the adaRMS dense) back to f32"""
    mod.to(BF)
    for name, p in mod.named_parameters():
        if "layernorm" in name:
            p.data = p.data.to(torch.float32)


def _gemma_cfg(width, mlp, adarms):
    return types.SimpleNamespace(hidden_size=width, num_attention_heads=8, num_key_value_heads=1, head_dim=256, attention_bias=False,
                                 attention_dropout=0.0, _attn_implementation="eager", intermediate_size=mlp,
                                 hidden_act="gelu_pytorch_tanh", rms_norm_eps=1e-6, use_adarms=adarms, adarms_cond_dim=width)  # fmt: skip


def _copy_layer(dst, src):
    sd = dict(src.named_parameters())
    assert set(sd) == {n for n, _ in dst.named_parameters()}
    for n, p in dst.named_parameters():
        p.data = sd[n].data.clone()


def _tables(g, B, S, HD=256):
    ang = _rnd(g, B, S, HD // 2, scale=3.0)
    return torch.cat([ang, ang], -1).cos().to(BF), torch.cat([ang, ang], -1).sin().to(BF)


@pytest.mark.parametrize("kind", ["expert", "prefix"])
def test_decoder_layer_at_real_width(ref, kind):
    g = torch.Generator().manual_seed(7 if kind == "expert" else 8)
    adaptive = kind == "expert"
    width, mlp = (1024, 4096) if adaptive else (2048, 16384)
    B, S, Pk = 2, 24, (40 if adaptive else 0)
    layer = ref.gem["GemmaDecoderLayer"](_gemma_cfg(width, mlp, adaptive), 0).eval()
    _set_params(g, layer, 0.03)
    _selected_bf16(layer)
    h = _rnd(g, B, S, width, dtype=BF)
    cos, sin = _tables(g, B, S)
    past = [(_rnd(g, B, 1, Pk, 256, dtype=BF), _rnd(g, B, 1, Pk, 256, dtype=BF))] if adaptive else None
    allowed = torch.rand(B, 1, S, Pk + S, generator=g) > 0.25
    allowed[..., 0] = True
    mask = torch.where(allowed, 0.0, O.MASK_VALUE)
    cond = _rnd(g, B, width) if adaptive else None
    with torch.no_grad():
        want = layer(h, attention_mask=mask, position_ids=None, past_key_value=past, use_cache=False,
                     position_embeddings=(cos, sin), adarms_cond=cond)[0]  # fmt: skip
    cfg = O.GemmaCfg(width=width, depth=1, mlp_dim=mlp, num_heads=8, num_kv_heads=1, head_dim=256)
    model = O.GemmaModel(cfg, vocab=8, use_adarms=adaptive, with_embed=False)
    _selected_bf16(model.layers[0])
    _copy_layer(model.layers[0], layer)
    if adaptive:
        _set_params(g, model.norm, 0.03)
    real = O.rotary_cos_sin
    O.rotary_cos_sin = lambda inv, pos, dt: (cos, sin)  # the stored tables instead of position ids
    try:
        with torch.no_grad():
            got, _ = model.forward_single(h, mask, torch.zeros(B, S, dtype=torch.long), past, False, cond)
            want_n, _ = model.norm(want, cond)  # forward_single ends with the model's final norm
    finally:
        O.rotary_cos_sin = real
    assert want.dtype == BF and float(want.float().abs().mean()) > 1e-2
    assert torch.equal(got, want_n)


def test_siglip_layer_at_real_width(ref):
    g = torch.Generator().manual_seed(9)
    scfg = types.SimpleNamespace(hidden_size=1152, num_attention_heads=16, intermediate_size=4304, layer_norm_eps=1e-6,
                                 hidden_act="gelu_pytorch_tanh", attention_dropout=0.0, _attn_implementation="eager")
    sl = ref.sig["SiglipEncoderLayer"](scfg).eval()
    _set_params(g, sl, 0.03)
    sl.to(BF)
    x = _rnd(g, 2, 256, 1152, dtype=BF)
    with torch.no_grad():
        want = sl(x, attention_mask=None)[0]
    layer = O.SiglipEncoderLayer(O.SiglipCfg(num_layers=1)).to(BF)  # So400m/14 defaults: 1152 / 16 x 72 / 4304
    _copy_layer(layer, sl)
    with torch.no_grad():
        got = layer(x)
    assert float(want.float().abs().mean()) > 1e-2
    assert torch.equal(got, want)


def test_joint_forward_at_real_width(ref):
    """PaliGemmaWithExpertModel.forward (both inputs) from the reference on a stub `self` holding reference layers: Gemma-2B
    width 2048 / 16384 next to the expert's 1024 / 4096, head_dim 256 MQA, padded prompt tokens, prefix-LM mask, one joint
    layer + the final norms; rotary from the vendored GemmaRotaryEmbedding with the bf16-rounded inv_freq."""
    import pytest as _pytest

    g = torch.Generator().manual_seed(10)
    gem = ref.gem
    gns = reflift.base_ns()
    gns.update({"dynamic_rope_update": (lambda f: f), "ROPE_INIT_FUNCTIONS": {}})
    reflift.lift(reflift.GEMMA_PY, ["GemmaRotaryEmbedding"], gns)
    Rot = gns["GemmaRotaryEmbedding"]
    rot = Rot.__new__(Rot)
    nn.Module.__init__(rot)
    inv_freq = O.rope_inv_freq(256).to(BF)
    rot.register_buffer("inv_freq", inv_freq, persistent=False)
    rot.attention_scaling = 1.0
    L = 1
    vl = [gem["GemmaDecoderLayer"](_gemma_cfg(2048, 16384, False), i).eval() for i in range(L)]
    el = [gem["GemmaDecoderLayer"](_gemma_cfg(1024, 4096, True), i).eval() for i in range(L)]
    vnorm, enorm = gem["GemmaRMSNorm"](2048), gem["GemmaRMSNorm"](1024, cond_dim=1024)
    for m in (*vl, *el):
        _set_params(g, m, 0.03)
        _selected_bf16(m)
    vnorm.weight.data = _rnd(g, 2048, scale=0.3)
    _set_params(g, enorm, 0.03)
    lm = types.SimpleNamespace(layers=vl, norm=vnorm, rotary_emb=rot)
    ex = types.SimpleNamespace(layers=el, norm=enorm, gradient_checkpointing=False)
    stub = types.SimpleNamespace(
        paligemma=types.SimpleNamespace(language_model=lm, model=types.SimpleNamespace(language_model=lm),
                                        config=types.SimpleNamespace(text_config=types.SimpleNamespace(num_hidden_layers=L))),
        gemma_expert=types.SimpleNamespace(model=ex), training=False)  # fmt: skip
    jns = reflift.base_ns()
    jns.update({"pytest": _pytest,
                "modeling_gemma": types.SimpleNamespace(**{k: gem[k] for k in ("apply_rotary_pos_emb", "eager_attention_forward", "_gated_residual")})})  # fmt: skip
    joint = reflift.lift_method(reflift.GEMMA_PT_PY, "PaliGemmaWithExpertModel", "forward", jns)
    B, P_, S_ = 2, 48, 16
    pad = torch.ones(B, P_ + S_, dtype=torch.bool)
    pad[0, 40:48] = False
    att = torch.zeros(B, P_ + S_, dtype=torch.bool)
    att[:, P_] = True
    mask = torch.where(ref.pi0["make_att_2d_masks"](pad, att)[:, None, :, :], 0.0, O.MASK_VALUE)
    pos = torch.cumsum(pad, dim=1) - 1
    pre, suf = _rnd(g, B, P_, 2048, dtype=BF), _rnd(g, B, S_, 1024, dtype=BF)
    cond = _rnd(g, B, 1024)
    with torch.no_grad():
        (wp, ws), _ = joint(stub, attention_mask=mask, position_ids=pos, past_key_values=None, inputs_embeds=[pre, suf],
                            use_cache=False, adarms_cond=[None, cond])  # fmt: skip

    vlm = O.GemmaCfg(width=2048, depth=L, mlp_dim=16384, num_heads=8, num_kv_heads=1, head_dim=256)
    exp = O.GemmaCfg(width=1024, depth=L, mlp_dim=4096, num_heads=8, num_kv_heads=1, head_dim=256)
    sc = O.SiglipCfg(hidden_size=16, num_layers=1, num_heads=2, intermediate_size=32, patch_size=14, image_size=28, projection_dim=2048)
    model = O.PaliGemmaWithExpertModel(vlm, exp, use_adarms=[False, True], precision="bfloat16", vocab=16, sc=sc)
    olm, oex = model.paligemma.language_model, model.gemma_expert.model
    for tower, layers in ((olm, vl), (oex, el)):
        for dst, src in zip(tower.layers, layers, strict=True):
            for (n, p), (n2, p2) in zip(dst.named_parameters(), src.named_parameters(), strict=True):
                assert n == n2 and p.dtype == p2.dtype, (n, n2)  # to_bfloat16_for_selected_params chose the same dtypes
                p.data = p2.data.clone()
    olm.norm.weight.data = vnorm.weight.data.clone()
    oex.norm.dense.weight.data, oex.norm.dense.bias.data = enorm.dense.weight.data.clone(), enorm.dense.bias.data.clone()
    olm.inv_freq = inv_freq.clone()
    with torch.no_grad():
        (gp, gs), _ = model(mask, pos, None, [pre, suf], False, [None, cond])
    assert float(ws.float().abs().mean()) > 1e-2
    assert torch.equal(gp, wp) and torch.equal(gs, ws)
