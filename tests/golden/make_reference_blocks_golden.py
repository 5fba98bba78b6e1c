"""Golden vectors for the building blocks of the hot path, produced by EXECUTING THE REFERENCE'S OWN CODE.

The reference package cannot be imported here (python 3.10, no flax/jax, transformers 5.x instead of the patched 4.53.2),
but the functions and classes the hot path is made of are plain torch.  This script lifts exactly those definitions out
of the reference source files with `ast` (nothing is copied into the repository — the source is read from
/root/reference at run time), executes them on seeded inputs and stores inputs, weights and outputs in
tests/golden/reference_blocks.safetensors.  tests/test_reference_blocks_cpu.py then holds the oracle to these vectors.

    python tests/golden/make_reference_blocks_golden.py      # build container only; needs /root/reference

Third-party pieces the lifted code touches: `ACT2FN["gelu_pytorch_tanh"]` is taken from the installed transformers
(un-vendored dependency, = F.gelu(approximate="tanh")); type-annotation-only names are bound to `typing.Any`.
"""
import ast
import os
import types
import typing

import torch
import torch.nn.functional as F
import typing_extensions
from safetensors.torch import save_file
from torch import nn
from transformers.activations import ACT2FN

REF = "/root/reference/src/openpi/models_pytorch"
HERE = os.path.dirname(os.path.abspath(__file__))


def lift(path, names, ns):
    tree = ast.parse(open(path).read())
    found = set()
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            found.add(node.name)
    assert found == set(names), set(names) - found
    return ns


def lift_method(path, cls, name, ns):
    """a method of a class, compiled as a free function taking `self` explicitly"""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    exec(compile(ast.Module([sub], []), path, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


def base_ns():
    import math

    return {"torch": torch, "nn": nn, "F": F, "math": math, "Optional": typing.Optional, "Union": typing.Union,
            "Callable": typing.Callable, "Tensor": torch.Tensor, "Cache": typing.Any, "GemmaConfig": typing.Any,
            "SiglipVisionConfig": typing.Any, "SiglipTextConfig": typing.Any, "FlashAttentionKwargs": dict,
            "Unpack": typing_extensions.Unpack, "GradientCheckpointingLayer": nn.Module, "ACT2FN": ACT2FN}  # fmt: skip


pi0 = lift(f"{REF}/pi0_pytorch.py", ["get_safe_dtype", "create_sinusoidal_pos_embedding", "make_att_2d_masks"], base_ns())
gem = lift(f"{REF}/transformers_replace/models/gemma/modeling_gemma.py",
           ["GemmaRMSNorm", "GemmaMLP", "rotate_half", "apply_rotary_pos_emb", "repeat_kv", "_gated_residual",
            "eager_attention_forward", "GemmaAttention", "GemmaDecoderLayer"], base_ns())  # fmt: skip
sig = lift(f"{REF}/transformers_replace/models/siglip/modeling_siglip.py",
           ["eager_attention_forward", "SiglipAttention", "SiglipMLP", "SiglipEncoderLayer"], base_ns())

g = torch.Generator().manual_seed(20240925)
BF = torch.bfloat16
out = {}


def rnd(*shape, scale=1.0, dtype=torch.float32):
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def put(prefix, **tensors):
    for k, v in tensors.items():
        t = v.detach().clone().contiguous()
        out[f"{prefix}.{k}"] = t.to(torch.uint8) if t.dtype == torch.bool else t


def set_params(mod, scale=0.2):
    for p in mod.parameters():
        p.data = rnd(*p.shape, scale=scale).to(p.dtype)


def selected_bf16(mod):
    """gemma_pytorch.py:63-83 on a single layer: everything bf16, the norms (incl. adaRMS dense) back to f32."""
    mod.to(BF)
    for name, p in mod.named_parameters():
        if "layernorm" in name:
            p.data = p.data.to(torch.float32)


# ---- A. attention mask construction + the docstring's own examples (pi0_pytorch.py:52-81) -----------------------
pad = torch.rand(4, 12, generator=g) > 0.2
att = torch.rand(4, 12, generator=g) > 0.6
att[0] = torch.tensor([1] * 12, dtype=torch.bool)  # pure causal
att[1] = torch.tensor([0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1], dtype=torch.bool)  # prefix-lm
att[2] = torch.tensor([1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0], dtype=torch.bool)  # blocks
pad[0] = pad[1] = True
put("masks", pad=pad, att=att, att2d=pi0["make_att_2d_masks"](pad, att), position_ids=torch.cumsum(pad, dim=1) - 1)

# ---- B. time embedding (pi0_pytorch.py:25-42) ----------------------------------------------------------------
t = torch.tensor([1.0, 0.9, 0.5, 0.001, 0.3333, 0.7], dtype=torch.float32)
put("sincos", time=t, out=pi0["create_sinusoidal_pos_embedding"](t, 64, 4e-3, 4.0, torch.device("cpu")))

# ---- C/D. RMSNorm, plain and adaptive (modeling_gemma.py:49-104) -------------------------------------------------
norm = gem["GemmaRMSNorm"](48)
norm.weight.data = rnd(48, scale=0.3)
x = rnd(2, 5, 48, dtype=BF)
put("rms_plain", x=x, w=norm.weight.data, y_bf16=norm(x)[0], y_f32=norm(x.float())[0])
ada = gem["GemmaRMSNorm"](48, cond_dim=32)
set_params(ada, 0.3)
cond = rnd(2, 32)
y, gate = ada(x, cond)
put("rms_ada", x=x, cond=cond, dense_w=ada.dense.weight.data, dense_b=ada.dense.bias.data, y=y, gate=gate)

# ---- E. rotary application (modeling_gemma.py:163-194) --------------------------------------------------------
q, k = rnd(2, 8, 6, 16, dtype=BF), rnd(2, 1, 6, 16, dtype=BF)
ang = rnd(2, 6, 8, scale=3.0)
cos, sin = torch.cat([ang, ang], -1).cos().to(BF), torch.cat([ang, ang], -1).sin().to(BF)
qe, ke = gem["apply_rotary_pos_emb"](q, k, cos, sin)
put("rope", q=q, k=k, cos=cos, sin=sin, q_out=qe, k_out=ke)

# ---- F. masked MQA attention (modeling_gemma.py:230-253) ---------------------------------------------------------
mod = types.SimpleNamespace(num_key_value_groups=8, training=False)
kk, vv = rnd(2, 1, 9, 16, dtype=BF), rnd(2, 1, 9, 16, dtype=BF)
allowed = torch.rand(2, 1, 6, 9, generator=g) > 0.3
allowed[..., 0] = True
mask = torch.where(allowed, 0.0, -2.3819763e38)
o_bf, w_bf = gem["eager_attention_forward"](mod, q, kk, vv, mask, 16**-0.5)
o_f32, _ = gem["eager_attention_forward"](mod, q.float(), kk.float(), vv.float(), mask, 16**-0.5)
put("attn", q=q, k=kk, v=vv, mask=mask, out_bf16=o_bf, probs_bf16=w_bf, out_f32=o_f32)

# ---- G. gated residual (modeling_gemma.py:209-227) ---------------------------------------------------------------
a_, b_, g_ = rnd(2, 5, 48, dtype=BF), rnd(2, 5, 48, dtype=BF), rnd(2, 1, 48, dtype=BF)
put("gated", x=a_, y=b_, gate=g_, out=gem["_gated_residual"](a_, b_, g_), out_nogate=gem["_gated_residual"](a_, b_, None))

# ---- H. GeGLU MLP (modeling_gemma.py:113-126) --------------------------------------------------------------------
cfg = types.SimpleNamespace(hidden_size=48, intermediate_size=96, hidden_act="gelu_pytorch_tanh")
mlp = gem["GemmaMLP"](cfg)
set_params(mlp)
mlp.to(BF)
put("mlp", x=x, gate_w=mlp.gate_proj.weight.data, up_w=mlp.up_proj.weight.data, down_w=mlp.down_proj.weight.data, y=mlp(x))

# ---- I. one decoder layer of the action expert: adaRMS, rotary, cached prefix K/V, gated residuals ----------------------
lcfg = types.SimpleNamespace(hidden_size=48, num_attention_heads=8, num_key_value_heads=1, head_dim=16, attention_bias=False,
                             attention_dropout=0.0, _attn_implementation="eager", intermediate_size=96,
                             hidden_act="gelu_pytorch_tanh", rms_norm_eps=1e-6, use_adarms=True, adarms_cond_dim=48)  # fmt: skip
layer = gem["GemmaDecoderLayer"](lcfg, 0).eval()
set_params(layer)
selected_bf16(layer)
h = rnd(2, 5, 48, dtype=BF)
ang = rnd(2, 5, 8, scale=3.0)
cos5, sin5 = torch.cat([ang, ang], -1).cos().to(BF), torch.cat([ang, ang], -1).sin().to(BF)
past = [(rnd(2, 1, 4, 16, dtype=BF), rnd(2, 1, 4, 16, dtype=BF))]
allowed = torch.rand(2, 1, 5, 9, generator=g) > 0.25
allowed[..., 0] = True
lmask = torch.where(allowed, 0.0, -2.3819763e38)
lcond = rnd(2, 48)
with torch.no_grad():
    lout = layer(h, attention_mask=lmask, position_ids=None, past_key_value=past, use_cache=False,
                 position_embeddings=(cos5, sin5), adarms_cond=lcond)[0]
put("layer_expert", h=h, cos=cos5, sin=sin5, past_k=past[0][0], past_v=past[0][1], mask=lmask, cond=lcond, out=lout,
    **{"w." + n: p.data for n, p in layer.named_parameters()})

# ---- I'. one plain (prefix) decoder layer ------------------------------------------------------------------------
pcfg = types.SimpleNamespace(**{**lcfg.__dict__, "use_adarms": False})
player = gem["GemmaDecoderLayer"](pcfg, 0).eval()
set_params(player)
selected_bf16(player)
pmask = lmask[..., :5].contiguous()
with torch.no_grad():
    pout = player(h, attention_mask=pmask, position_ids=None, past_key_value=None, use_cache=False,
                  position_embeddings=(cos5, sin5), adarms_cond=None)[0]
put("layer_prefix", h=h, cos=cos5, sin=sin5, mask=pmask, out=pout, **{"w." + n: p.data for n, p in player.named_parameters()})

# ---- J. one SigLIP encoder layer (modeling_siglip.py:325-480) ----------------------------------------------------
scfg = types.SimpleNamespace(hidden_size=48, num_attention_heads=4, intermediate_size=96, layer_norm_eps=1e-6,
                             hidden_act="gelu_pytorch_tanh", attention_dropout=0.0, _attn_implementation="eager")
sl = sig["SiglipEncoderLayer"](scfg).eval()
set_params(sl)
sl.to(BF)
sx = rnd(2, 7, 48, dtype=BF)
with torch.no_grad():
    sy = sl(sx, attention_mask=None)[0]
put("siglip_layer", x=sx, y=sy, **{"w." + n: p.data for n, p in sl.named_parameters()})

# ---- K. the joint (prefix + expert) layer loop and final norms: PaliGemmaWithExpertModel.forward, both inputs given
#         (gemma_pytorch.py:126-279), executed on a stub `self` that holds reference GemmaDecoderLayer / GemmaRMSNorm objects.
#         The rotary table comes from the vendored GemmaRotaryEmbedding.forward (modeling_gemma.py:149-161); its inv_freq is
#         the un-vendored transformers default, restated here and bf16-rounded as `.to(bfloat16)` leaves it.
import pytest  # noqa: E402  (the reference signature mentions pytest.Cache)

gns = base_ns()
gns.update({"dynamic_rope_update": (lambda f: f), "ROPE_INIT_FUNCTIONS": {}})
lift(f"{REF}/transformers_replace/models/gemma/modeling_gemma.py", ["GemmaRotaryEmbedding"], gns)
Rot = gns["GemmaRotaryEmbedding"]
rot = Rot.__new__(Rot)
nn.Module.__init__(rot)
inv_freq = (1.0 / (10000.0 ** (torch.arange(0, 16, 2, dtype=torch.int64).to(torch.float) / 16))).to(BF)
rot.register_buffer("inv_freq", inv_freq, persistent=False)
rot.attention_scaling = 1.0

L = 2
vcfg = types.SimpleNamespace(**{**lcfg.__dict__, "use_adarms": False})
ecfg = types.SimpleNamespace(**{**lcfg.__dict__, "hidden_size": 32, "intermediate_size": 64, "adarms_cond_dim": 32})
vl = [gem["GemmaDecoderLayer"](vcfg, i).eval() for i in range(L)]
el = [gem["GemmaDecoderLayer"](ecfg, i).eval() for i in range(L)]
vnorm, enorm = gem["GemmaRMSNorm"](48), gem["GemmaRMSNorm"](32, cond_dim=32)
for m in (*vl, *el):
    set_params(m)
    selected_bf16(m)
vnorm.weight.data = rnd(48, scale=0.3)
set_params(enorm, 0.3)
lm = types.SimpleNamespace(layers=vl, norm=vnorm, rotary_emb=rot)
ex = types.SimpleNamespace(layers=el, norm=enorm, gradient_checkpointing=False)
stub = types.SimpleNamespace(
    paligemma=types.SimpleNamespace(language_model=lm, model=types.SimpleNamespace(language_model=lm),
                                    config=types.SimpleNamespace(text_config=types.SimpleNamespace(num_hidden_layers=L))),
    gemma_expert=types.SimpleNamespace(model=ex), training=False)  # fmt: skip
jns = base_ns()
jns.update({"pytest": pytest, "modeling_gemma": types.SimpleNamespace(**{k: gem[k] for k in ("apply_rotary_pos_emb", "eager_attention_forward", "_gated_residual")})})
joint = lift_method(f"{REF}/gemma_pytorch.py", "PaliGemmaWithExpertModel", "forward", jns)
P_, S_ = 7, 5
jpad = torch.ones(2, P_ + S_, dtype=torch.bool)
jpad[0, 5:7] = False  # padded prompt tokens
jatt = torch.zeros(2, P_ + S_, dtype=torch.bool)
jatt[:, P_] = True
att2d = pi0["make_att_2d_masks"](jpad, jatt)
jmask = torch.where(att2d[:, None, :, :], 0.0, -2.3819763e38)
jpos = torch.cumsum(jpad, dim=1) - 1
pre, suf = rnd(2, P_, 48, dtype=BF), rnd(2, S_, 32, dtype=BF)
jcond = rnd(2, 32)
with torch.no_grad():
    (jp, js), _ = joint(stub, attention_mask=jmask, position_ids=jpos, past_key_values=None, inputs_embeds=[pre, suf],
                        use_cache=False, adarms_cond=[None, jcond])
w = {}
for i in range(L):
    w.update({f"w.vlm.{i}.{n}": p_.data for n, p_ in vl[i].named_parameters()})
    w.update({f"w.exp.{i}.{n}": p_.data for n, p_ in el[i].named_parameters()})
put("joint", pad=jpad, att=jatt, mask=jmask, pos=jpos, x_prefix=pre, x_suffix=suf, cond=jcond, prefix_out=jp, suffix_out=js,
    inv_freq=inv_freq, vnorm_w=vnorm.weight.data, enorm_w=enorm.dense.weight.data, enorm_b=enorm.dense.bias.data, **w)

# ---- L. embed_suffix, pi0.5 branch (pi0_pytorch.py:237-314): time embedding -> time MLP (swish) -> adaRMS condition; action
#         projection; suffix pad / att masks
sns = base_ns()
sns["create_sinusoidal_pos_embedding"] = pi0["create_sinusoidal_pos_embedding"]
embed_suffix = lift_method(f"{REF}/pi0_pytorch.py", "PI0Pytorch", "embed_suffix", sns)
ain, tin, tout = nn.Linear(6, 32), nn.Linear(32, 32), nn.Linear(32, 32)
for m in (ain, tin, tout):
    set_params(m, 0.3)
sself = types.SimpleNamespace(pi05=True, action_in_proj=ain, time_mlp_in=tin, time_mlp_out=tout,
                              config=types.SimpleNamespace(action_horizon=5), _apply_checkpoint=lambda f, *a: f(*a))
na, tt = rnd(2, 5, 6), torch.tensor([0.7, 0.05], dtype=torch.float32)
with torch.no_grad():
    embs, spad, satt, scond = embed_suffix(sself, None, na, tt)
put("suffix", noisy_actions=na, time=tt, embs=embs, pad=spad, att=satt, cond=scond, ain_w=ain.weight.data, ain_b=ain.bias.data,
    tin_w=tin.weight.data, tin_b=tin.bias.data, tout_w=tout.weight.data, tout_b=tout.bias.data)

path = os.path.join(HERE, "reference_blocks.safetensors")
save_file(out, path)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "tensors")
