"""JAX parameter tree <-> torch state-dict (kai0_amd/convert.py).  No JAX here, so the axis conventions are
checked against the contraction each JAX module WRITES (the einsum strings / matmuls quoted from the reference source):
weights taken from the oracle, mapped to the JAX layout, pushed through that contraction in numpy, compared with the torch
layer of the oracle on the same input.  Plus shapes of the JAX tree, exact round trip, and loading into the model."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny import tiny_cfgs  # noqa: E402

from kai0_amd import convert  # noqa: E402
from oracle import pi0_oracle as O  # noqa: E402

LLM, IMG = "PaliGemma/llm/", "PaliGemma/img/"


def _oracle_f32():
    _, ocfg = tiny_cfgs()
    m = O.OraclePI0(ocfg)
    O.synthetic_weights_(m, seed=3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.mul_(4.0)
    m.paligemma_with_expert.to_bfloat16_for_selected_params("float32")
    return m, ocfg


def test_jax_tree_shapes_and_exact_round_trip():
    m, ocfg = _oracle_f32()
    sd = m.state_dict()
    jp = convert.torch_to_jax(sd, siglip_heads=ocfg.siglip.num_heads)
    L, D, H, F = 4, 64, 16, 128  # models/gemma.py:60-68 ("dummy"): width 64, depth 4, mlp 128, 8 / 1 heads of 16
    want = {LLM + "embedder/input_embedding": (ocfg.vocab_size, D), LLM + "layers/attn/q_einsum/w": (L, 8, D, H),
            LLM + "layers/attn/kv_einsum/w": (L, 2, 1, D, H), LLM + "layers/attn/attn_vec_einsum/w": (L, 8, H, D),
            LLM + "layers/mlp/gating_einsum": (L, 2, D, F), LLM + "layers/mlp/linear": (L, F, D),
            LLM + "layers/pre_attention_norm/scale": (L, D), LLM + "layers/pre_ffw_norm/scale": (L, D), LLM + "final_norm/scale": (D,),
            LLM + "layers/attn/q_einsum_1/w": (L, 8, D, H), LLM + "layers/mlp_1/gating_einsum": (L, 2, D, F),
            LLM + "layers/pre_attention_norm_1/Dense_0/kernel": (L, D, 3 * D), LLM + "layers/pre_ffw_norm_1/Dense_0/bias": (L, 3 * D),
            LLM + "final_norm_1/Dense_0/kernel": (D, 3 * D), IMG + "embedding/kernel": (14, 14, 3, 64), IMG + "pos_embedding": (1, 16, 64),
            IMG + "Transformer/encoderblock/MultiHeadDotProductAttention_0/query/kernel": (2, 64, 4, 16),
            IMG + "Transformer/encoderblock/MultiHeadDotProductAttention_0/query/bias": (2, 4, 16),
            IMG + "Transformer/encoderblock/MultiHeadDotProductAttention_0/out/kernel": (2, 4, 16, 64),
            IMG + "Transformer/encoderblock/MlpBlock_0/Dense_0/kernel": (2, 64, 136), IMG + "Transformer/encoderblock/LayerNorm_1/scale": (2, 64),
            IMG + "Transformer/encoder_norm/bias": (64,), IMG + "head/kernel": (64, 64), "action_in_proj/kernel": (32, D),
            "action_out_proj/kernel": (D, 32), "time_mlp_in/bias": (D,)}  # fmt: skip
    for k, shp in want.items():
        assert jp[k].shape == shp, (k, jp[k].shape, shp)
    back = convert.jax_to_torch(jp, fill_missing=True)
    assert set(back) == set(sd)  # every key of the contract, incl. the tied lm_head and the dead expert lm_head
    for k, v in sd.items():
        if k.endswith("gemma_expert.lm_head.weight"):
            assert back[k].shape == v.shape and not back[k].any()
        else:
            assert back[k].dtype == v.dtype and torch.equal(back[k], v), k
    assert back[convert.PWE + "paligemma.lm_head.weight"] is back[convert.LM + "embed_tokens.weight"]
    # nested input with the wrappers restore_params may leave around the leaves (model.py:360-364)
    nested = {"params": {}}
    for k, v in jp.items():
        d = nested["params"]
        for part in k.split("/"):
            d = d.setdefault(part, {})
        d["value"] = v
    again = convert.jax_to_torch(nested)
    assert all(torch.equal(again[k], back[k]) for k in again)


def test_axis_conventions_against_the_jax_contractions():
    m, ocfg = _oracle_f32()
    jp = convert.torch_to_jax(m.state_dict(), siglip_heads=ocfg.siglip.num_heads)
    g = torch.Generator().manual_seed(0)
    lm = m.paligemma_with_expert.paligemma.language_model
    ex = m.paligemma_with_expert.gemma_expert.model
    x = torch.randn(2, 5, 64, generator=g)
    xn = x.numpy()
    close = lambda a, b: np.allclose(a, b.detach().numpy() if isinstance(b, torch.Tensor) else b, atol=1e-5)  # noqa: E731
    for sfx, tower in (("", lm), ("_1", ex)):
        for i in (0, 3):
            att, mlp = tower.layers[i].self_attn, tower.layers[i].mlp
            q = np.einsum("BTD,NDH->BTNH", xn, jp[f"{LLM}layers/attn/q_einsum{sfx}/w"][i])                 # gemma.py:191
            assert close(q, att.q_proj(x).view(2, 5, 8, 16))
            k, v = np.einsum("BSD,CKDH->CBSKH", xn, jp[f"{LLM}layers/attn/kv_einsum{sfx}/w"][i])           # gemma.py:198 "BSD,2KDH->2BSKH" (numpy wants a letter for the axis JAX calls '2')
            assert close(k, att.k_proj(x).view(2, 5, 1, 16)) and close(v, att.v_proj(x).view(2, 5, 1, 16))
            enc = torch.randn(2, 5, 8, 16, generator=g)
            o = np.einsum("BTNH,NHD->BTD", enc.numpy(), jp[f"{LLM}layers/attn/attn_vec_einsum{sfx}/w"][i])  # gemma.py:245
            assert close(o, att.o_proj(enc.reshape(2, 5, 128)))
            w = jp[f"{LLM}layers/mlp{sfx}/gating_einsum"][i]
            assert close(xn @ w[0], mlp.gate_proj(x)) and close(xn @ w[1], mlp.up_proj(x))                 # gemma.py:266-269
            a = torch.randn(2, 5, 128, generator=g)
            assert close(a.numpy() @ jp[f"{LLM}layers/mlp{sfx}/linear"][i], mlp.down_proj(a))              # gemma.py:278
    cond = torch.randn(2, 64, generator=g)
    mod = cond.numpy() @ jp[LLM + "layers/pre_ffw_norm_1/Dense_0/kernel"][2] + jp[LLM + "layers/pre_ffw_norm_1/Dense_0/bias"][2]
    assert close(mod, ex.layers[2].post_attention_layernorm.dense(cond))                                   # gemma.py:128
    assert close(cond.numpy() @ jp[LLM + "final_norm_1/Dense_0/kernel"] + jp[LLM + "final_norm_1/Dense_0/bias"], ex.norm.dense(cond))
    assert np.array_equal(jp[LLM + "layers/pre_attention_norm/scale"][1], lm.layers[1].input_layernorm.weight.detach().numpy())
    tok = torch.tensor([[3, 17, 200]])
    assert close(jp[LLM + "embedder/input_embedding"][tok.numpy()], lm.embed_tokens(tok))                  # gemma.py:148-149

    vt = m.paligemma_with_expert.paligemma.model.vision_tower.vision_model
    img = torch.randn(2, 3, 56, 56, generator=g)
    nhwc = img.permute(0, 2, 3, 1).numpy()
    patches = nhwc.reshape(2, 4, 14, 4, 14, 3).transpose(0, 1, 3, 2, 4, 5)                                   # [n, gh, gw, ph, pw, c]
    stem = np.einsum("nhwpqc,pqcd->nhwd", patches, jp[IMG + "embedding/kernel"]) + jp[IMG + "embedding/bias"]  # nn.Conv VALID, stride = patch
    assert close(stem.reshape(2, 16, 64), vt.embeddings.patch_embedding(img).flatten(2).transpose(1, 2))
    assert np.array_equal(jp[IMG + "pos_embedding"][0], vt.embeddings.position_embedding.weight.detach().numpy())
    y = torch.randn(2, 16, 64, generator=g)
    blk = IMG + "Transformer/encoderblock/"
    layer = vt.encoder.layers[1]
    for jn, lin in (("query", layer.self_attn.q_proj), ("key", layer.self_attn.k_proj), ("value", layer.self_attn.v_proj)):
        proj = np.einsum("nld,dhk->nlhk", y.numpy(), jp[f"{blk}MultiHeadDotProductAttention_0/{jn}/kernel"][1])  # flax DenseGeneral
        proj = proj + jp[f"{blk}MultiHeadDotProductAttention_0/{jn}/bias"][1]
        assert close(proj, lin(y).view(2, 16, 4, 16))
    heads = torch.randn(2, 16, 4, 16, generator=g)
    o = np.einsum("nlhk,hkd->nld", heads.numpy(), jp[blk + "MultiHeadDotProductAttention_0/out/kernel"][1]) + jp[blk + "MultiHeadDotProductAttention_0/out/bias"][1]
    assert close(o, layer.self_attn.out_proj(heads.reshape(2, 16, 64)))
    assert close(y.numpy() @ jp[blk + "MlpBlock_0/Dense_0/kernel"][1] + jp[blk + "MlpBlock_0/Dense_0/bias"][1], layer.mlp.fc1(y))
    z = torch.randn(2, 16, 136, generator=g)
    assert close(z.numpy() @ jp[blk + "MlpBlock_0/Dense_1/kernel"][1] + jp[blk + "MlpBlock_0/Dense_1/bias"][1], layer.mlp.fc2(z))
    assert close(y.numpy() @ jp[IMG + "head/kernel"] + jp[IMG + "head/bias"], m.paligemma_with_expert.paligemma.model.multi_modal_projector.linear(y))
    act = torch.randn(2, 10, 32, generator=g)
    assert close(act.numpy() @ jp["action_in_proj/kernel"] + jp["action_in_proj/bias"], m.action_in_proj(act))  # nnx.Linear, pi0.py:93


def test_converted_weights_drive_the_model_to_the_same_loss():
    """torch -> JAX layout -> torch -> load_state_dict(strict) -> identical loss tensor (the mapping loses nothing)."""
    _, ocfg = tiny_cfgs()
    a = O.OraclePI0(ocfg)
    O.synthetic_weights_(a, seed=5)
    sd = {k: v for k, v in a.state_dict().items()}
    back = convert.jax_to_torch(convert.torch_to_jax(sd, siglip_heads=ocfg.siglip.num_heads), fill_missing=True)
    b = O.OraclePI0(ocfg)
    want = b.state_dict()
    b.load_state_dict({k: v.to(want[k].dtype) for k, v in back.items()}, strict=True)  # bf16 -> f32 widening is exact, so is the way back
    obs, actions, noise, time = O.synthetic_batch(ocfg, 2, seed=1)
    with torch.no_grad():
        assert torch.equal(a(obs, actions, noise, time), b(obs, actions, noise, time))


def test_restore_params_and_convert_checkpoint_with_a_stub_orbax(tmp_path, monkeypatch):
    """`restore_params` / `convert_checkpoint` against a stand-in for orbax.checkpoint (absent in this image) that serves a
    training-style tree (every leaf wrapped in {"value": ...}, as nnx.State saves it): the suffix is stripped, the tree goes
    through jax_to_torch into the model, and the written directory is what create_trained_policy reads."""
    import sys
    import types

    import numpy as np
    from tiny import tiny_cfgs

    from kai0_amd import convert
    from kai0_amd.model import PI0Pytorch

    pcfg, _ = tiny_cfgs()
    torch.manual_seed(3)
    src_model = PI0Pytorch(pcfg)
    tree_flat = convert.torch_to_jax(src_model.state_dict(), num_heads=8, num_kv_heads=1, siglip_heads=pcfg.siglip.num_heads)
    nested: dict = {}
    for k, v in tree_flat.items():
        node = nested
        parts = k.split("/")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = {"value": v}

    class _Ckptr:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def metadata(self, path):
            assert str(path).endswith("params")
            return {"params": nested, "opt_state": {}}

        def restore(self, path, args):
            assert set(args.item) == {"params"}
            return {"params": args.item["params"]}

    stub = types.ModuleType("orbax.checkpoint")
    stub.PyTreeCheckpointer = _Ckptr
    stub.ArrayRestoreArgs = lambda **kw: kw
    stub.args = types.SimpleNamespace(PyTreeRestore=lambda item, restore_args: types.SimpleNamespace(item=item, restore_args=restore_args))
    pkg = types.ModuleType("orbax")
    pkg.checkpoint = stub
    monkeypatch.setitem(sys.modules, "orbax", pkg)
    monkeypatch.setitem(sys.modules, "orbax.checkpoint", stub)
    (tmp_path / "jax" / "params").mkdir(parents=True)
    (tmp_path / "jax" / "assets" / "robot").mkdir(parents=True)
    (tmp_path / "jax" / "assets" / "robot" / "norm_stats.json").write_text("{}")
    tree = convert.restore_params(tmp_path / "jax" / "params")
    assert isinstance(tree["PaliGemma"]["llm"]["embedder"]["input_embedding"], np.ndarray)  # "value" level removed
    out = convert.convert_checkpoint(tmp_path / "jax", tmp_path / "torch", config=pcfg)
    assert sorted(p.name for p in (tmp_path / "torch").iterdir()) == ["assets", "model.safetensors"]
    from kai0_amd.checkpoint import load_model_safetensors

    m2 = PI0Pytorch(pcfg)
    load_model_safetensors(m2, out + "/model.safetensors")
    dead = "paligemma_with_expert.gemma_expert.lm_head.weight"
    for (k, a), (_, b) in zip(src_model.state_dict().items(), m2.state_dict().items()):
        assert a.dtype == b.dtype, k
        if k != dead:  # the JAX tree has no entry for the dead expert lm_head: zero-filled
            assert torch.equal(a, b), k
    monkeypatch.delitem(sys.modules, "orbax.checkpoint")
    monkeypatch.delitem(sys.modules, "orbax")
    with pytest.raises(ImportError, match="orbax-checkpoint"):
        convert.restore_params(tmp_path / "jax" / "params")
