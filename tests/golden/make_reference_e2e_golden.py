"""End-to-end golden vectors produced by EXECUTING THE REFERENCE'S OWN CODE on the tiny configuration of the parity tests.

Same method as make_reference_blocks_golden.py (definitions lifted out of /root/reference with `ast`, nothing copied), but
here the whole path is assembled from reference pieces on stub `self` objects:
  SiglipVisionTransformer (embeddings + encoder + post-LN)  -> PaliGemmaModel.get_image_features + projector
  GemmaModel.forward (prefix pass with a KV cache; suffix pass against it) over reference GemmaDecoderLayer objects
  PaliGemmaWithExpertModel.forward (all three branches)
  PI0Pytorch.embed_prefix / embed_suffix / forward (flow-matching loss) / denoise_step / sample_actions (Euler loop)
Weights = the oracle's synthetic state dict (identical key names), inputs = the oracle's synthetic batch, so the stored
loss tensor and action chunk are what THE REFERENCE computes for the very inputs the parity tests use.

Un-vendored third-party behaviour restated here: transformers' DynamicCache (append on the sequence axis),
create_causal_mask (a 4-D mask passes through), the default rotary inv_freq, ACT2FN["gelu_pytorch_tanh"] (installed copy).

    python tests/golden/make_reference_e2e_golden.py      # build container only; needs /root/reference
"""
import functools
import logging
import os
import sys
import types
import typing

import torch
import torch.nn.functional as F
from safetensors.torch import save_file
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_reference_blocks_golden as B  # noqa: E402  (lift / lift_method / base_ns; regenerates the block fixture too)
from tiny import tiny_cfgs  # noqa: E402

from oracle import pi0_oracle as O  # noqa: E402

REF = B.REF
BF = torch.bfloat16
ident = lambda f: f  # noqa: E731


class Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class StubCache:  # transformers DynamicCache, the two calls the reference makes
    def __init__(self):
        self.k, self.v = [], []

    def get_seq_length(self, layer_idx=0):
        return self.k[0].shape[2] if self.k else 0

    def update(self, k, v, layer_idx, cache_kwargs=None):
        if layer_idx == len(self.k):
            self.k.append(k)
            self.v.append(v)
        else:
            self.k[layer_idx] = torch.cat([self.k[layer_idx], k], dim=2)
            self.v[layer_idx] = torch.cat([self.v[layer_idx], v], dim=2)
        return self.k[layer_idx], self.v[layer_idx]

    def __getitem__(self, i):
        return self.k[i], self.v[i]


def assign(module, sd, prefix):
    names = {n for n, _ in module.named_parameters()}
    got = {k[len(prefix):] for k in sd if k.startswith(prefix)}
    assert names <= got, (prefix, names - got)
    for n, p in module.named_parameters():
        p.data = sd[prefix + n].clone()  # keeps the stored dtype (bf16 weights, f32 selected params)


# ---------------------------------------------------------------------------------------------------------------- oracle side
_, ocfg = tiny_cfgs()
oracle = O.OraclePI0(ocfg)
O.synthetic_weights_(oracle, seed=0)
with torch.no_grad():
    for n, p in oracle.named_parameters():
        if p.dim() >= 2:
            p.mul_(0.08 / 0.02)  # same rescale as tests/tiny.build_pair(std=0.08)
sd = {k: v.detach().clone() for k, v in oracle.state_dict().items()}
obs, actions, noise, time = O.synthetic_batch(ocfg, 2, seed=0)
vlm, exp, sc = O.get_gemma_config("dummy"), O.get_gemma_config("dummy"), ocfg.siglip
PWE = "paligemma_with_expert."

# ------------------------------------------------------------------------------------------------------------- reference side
sns = B.base_ns()
sns.update({"can_return_tuple": ident, "auto_docstring": ident, "BaseModelOutput": Out, "BaseModelOutputWithPooling": Out,
            "torch_int": int, "SiglipConfig": typing.Any, "PaliGemmaConfig": typing.Any})  # fmt: skip
B.lift(f"{REF}/transformers_replace/models/siglip/modeling_siglip.py",
       ["eager_attention_forward", "SiglipAttention", "SiglipMLP", "SiglipEncoderLayer", "SiglipVisionEmbeddings", "SiglipEncoder",
        "SiglipVisionTransformer"], sns)  # fmt: skip
scfg = types.SimpleNamespace(hidden_size=sc.hidden_size, num_hidden_layers=sc.num_layers, num_attention_heads=sc.num_heads,
                             intermediate_size=sc.intermediate_size, patch_size=sc.patch_size, image_size=sc.image_size,
                             num_channels=3, layer_norm_eps=sc.layer_norm_eps, hidden_act="gelu_pytorch_tanh",
                             attention_dropout=0.0, _attn_implementation="eager", vision_use_head=False, output_attentions=False,
                             output_hidden_states=False, projection_dim=sc.projection_dim)  # fmt: skip
vt = sns["SiglipVisionTransformer"](scfg).eval()
assign(vt, sd, PWE + "paligemma.model.vision_tower.vision_model.")
pns = B.base_ns()
pns.update({"PaliGemmaConfig": typing.Any, "can_return_tuple": ident, "auto_docstring": ident})
B.lift(f"{REF}/transformers_replace/models/paligemma/modeling_paligemma.py", ["PaliGemmaMultiModalProjector"], pns)
proj = pns["PaliGemmaMultiModalProjector"](types.SimpleNamespace(vision_config=scfg)).eval()
assign(proj, sd, PWE + "paligemma.model.multi_modal_projector.")
get_image_features = B.lift_method(f"{REF}/transformers_replace/models/paligemma/modeling_paligemma.py", "PaliGemmaModel",
                                   "get_image_features", pns)

gns = B.base_ns()
gns.update({"dynamic_rope_update": ident, "can_return_tuple": ident, "auto_docstring": ident, "ROPE_INIT_FUNCTIONS": {},
            "DynamicCache": StubCache,
            "create_causal_mask": lambda **kw: kw["attention_mask"], "BaseModelOutputWithPast": Out,
            "logger": logging.getLogger("ref")})  # fmt: skip
B.lift(f"{REF}/transformers_replace/models/gemma/modeling_gemma.py",
       ["GemmaRMSNorm", "GemmaMLP", "GemmaRotaryEmbedding", "rotate_half", "apply_rotary_pos_emb", "repeat_kv", "_gated_residual",
        "eager_attention_forward", "GemmaAttention", "GemmaDecoderLayer"], gns)  # fmt: skip
gemma_forward = B.lift_method(f"{REF}/transformers_replace/models/gemma/modeling_gemma.py", "GemmaModel", "forward", gns)


def gemma_model(cfg, adaptive, prefix, with_embed):
    c = types.SimpleNamespace(hidden_size=cfg.width, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                              head_dim=cfg.head_dim, attention_bias=False, attention_dropout=0.0, _attn_implementation="eager",
                              intermediate_size=cfg.mlp_dim, hidden_act="gelu_pytorch_tanh", rms_norm_eps=1e-6,
                              use_adarms=adaptive, adarms_cond_dim=cfg.width if adaptive else None, num_hidden_layers=cfg.depth,
                              output_attentions=False, output_hidden_states=False, use_cache=False)  # fmt: skip
    layers = [gns["GemmaDecoderLayer"](c, i).eval() for i in range(cfg.depth)]
    for i, layer in enumerate(layers):
        assign(layer, sd, f"{prefix}layers.{i}.")
    norm = gns["GemmaRMSNorm"](cfg.width, cond_dim=cfg.width if adaptive else None)
    assign(norm, sd, prefix + "norm.")
    Rot = gns["GemmaRotaryEmbedding"]
    rot = Rot.__new__(Rot)
    nn.Module.__init__(rot)
    inv = 1.0 / (10000.0 ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.int64).to(torch.float) / cfg.head_dim))
    rot.register_buffer("inv_freq", inv.to(BF), persistent=False)  # `.to(bfloat16)` on the module rounds the buffer
    rot.attention_scaling = 1.0
    m = types.SimpleNamespace(config=c, gradient_checkpointing=False, training=False, layers=layers, norm=norm, rotary_emb=rot,
                              embed_tokens=None)  # fmt: skip
    if with_embed:
        m.embed_tokens = nn.Embedding(ocfg.vocab_size, cfg.width)
        m.embed_tokens.weight.data = sd[prefix + "embed_tokens.weight"].clone()
    m.forward = functools.partial(gemma_forward, m)
    return m


lm = gemma_model(vlm, False, PWE + "paligemma.model.language_model.", True)
ex = gemma_model(exp, True, PWE + "gemma_expert.model.", False)

jns = B.base_ns()
import pytest  # noqa: E402

jns.update({"pytest": pytest, "modeling_gemma": types.SimpleNamespace(**{k: gns[k] for k in ("apply_rotary_pos_emb", "eager_attention_forward", "_gated_residual")})})
pwe = types.SimpleNamespace(training=False)
pwe.paligemma = types.SimpleNamespace(
    language_model=lm,
    model=types.SimpleNamespace(language_model=lm, vision_tower=vt, multi_modal_projector=proj),
    config=types.SimpleNamespace(text_config=types.SimpleNamespace(num_hidden_layers=vlm.depth)))
pwe.paligemma.model.get_image_features = functools.partial(get_image_features, pwe.paligemma.model)
pwe.gemma_expert = types.SimpleNamespace(model=ex)
for name in ("forward", "embed_image", "embed_language_tokens"):
    setattr(pwe, name, functools.partial(B.lift_method(f"{REF}/gemma_pytorch.py", "PaliGemmaWithExpertModel", name, jns), pwe))

mns = B.base_ns()
B.lift(f"{REF}/pi0_pytorch.py", ["get_safe_dtype", "create_sinusoidal_pos_embedding", "make_att_2d_masks"], mns)
pol = types.SimpleNamespace(config=types.SimpleNamespace(action_horizon=ocfg.action_horizon, action_dim=ocfg.action_dim), pi05=True,
                            gradient_checkpointing_enabled=False, training=False, paligemma_with_expert=pwe)  # fmt: skip
for head, (i, o) in {"action_in_proj": (ocfg.action_dim, exp.width), "action_out_proj": (exp.width, ocfg.action_dim),
                     "time_mlp_in": (exp.width, exp.width), "time_mlp_out": (exp.width, exp.width)}.items():
    lin = nn.Linear(i, o)
    assign(lin, sd, head + ".")
    setattr(pol, head, lin)
for name in ("_apply_checkpoint", "_prepare_attention_masks_4d", "embed_prefix", "embed_suffix", "denoise_step", "forward",
             "sample_actions"):
    setattr(pol, name, functools.partial(B.lift_method(f"{REF}/pi0_pytorch.py", "PI0Pytorch", name, mns), pol))
# preprocessing with train=False on images already at resolution is the identity (preprocessing_pytorch.py:20-173)
pol._preprocess_observation = lambda o, train=True: (list(o.images.values()), list(o.image_masks.values()), o.tokenized_prompt,
                                                     o.tokenized_prompt_mask, o.state)  # fmt: skip

with torch.no_grad():
    ref_loss = pol.forward(obs, actions, noise=noise, time=time)
    ref_actions = pol.sample_actions(torch.device("cpu"), obs, noise=noise.clone(), num_steps=10)
    feats = pwe.embed_image(obs.images["base_0_rgb"])
# gradients of mean(loss) through the reference code (autograd over the lifted modules), for a spread of parameters
GRAD_KEYS = ["action_in_proj.weight", "action_out_proj.bias", "time_mlp_out.weight",
             PWE + "gemma_expert.model.layers.0.self_attn.q_proj.weight", PWE + "gemma_expert.model.layers.3.input_layernorm.dense.bias",
             PWE + "gemma_expert.model.norm.dense.weight", PWE + "paligemma.model.language_model.layers.2.mlp.down_proj.weight",
             PWE + "paligemma.model.language_model.layers.0.input_layernorm.weight",
             PWE + "paligemma.model.language_model.embed_tokens.weight",
             PWE + "paligemma.model.vision_tower.vision_model.encoder.layers.1.mlp.fc1.weight",
             PWE + "paligemma.model.vision_tower.vision_model.embeddings.patch_embedding.weight",
             PWE + "paligemma.model.multi_modal_projector.linear.bias"]  # fmt: skip
ref_params = {}
for prefix, mod in ((PWE + "paligemma.model.vision_tower.vision_model.", vt), (PWE + "paligemma.model.multi_modal_projector.", proj),
                    ("action_in_proj.", pol.action_in_proj), ("action_out_proj.", pol.action_out_proj),
                    ("time_mlp_in.", pol.time_mlp_in), ("time_mlp_out.", pol.time_mlp_out)):
    ref_params.update({prefix + n: p_ for n, p_ in mod.named_parameters()})
for tower, prefix in ((lm, PWE + "paligemma.model.language_model."), (ex, PWE + "gemma_expert.model.")):
    for i, layer in enumerate(tower.layers):
        ref_params.update({f"{prefix}layers.{i}.{n}": p_ for n, p_ in layer.named_parameters()})
    ref_params.update({prefix + "norm." + n: p_ for n, p_ in tower.norm.named_parameters()})
ref_params[PWE + "paligemma.model.language_model.embed_tokens.weight"] = lm.embed_tokens.weight
for p_ in ref_params.values():
    p_.requires_grad_(True)
pol.forward(obs, actions, noise=noise, time=time).mean().backward()
ref_grads = {"grad." + k: ref_params[k].grad.detach().clone() for k in GRAD_KEYS}
oracle.zero_grad(set_to_none=True)
oracle(obs, actions, noise, time).mean().backward()
ograds = dict(oracle.named_parameters())
print("oracle vs reference gradients: max|d|", max(float((ograds[k].grad - ref_grads["grad." + k]).abs().max()) for k in GRAD_KEYS))

# ---- AdvantageEstimator (pi0_pytorch.py:464-644): forward with progress targets and sample_values, six images -----------
#      (two timesteps x three cameras, deliberately inserted out of order), executed on the same stub with the class's own
#      forward / sample_values lifted; preprocessing (apply_aug=False, native resolution) is the reference's key sort.
AE = {}
torch.manual_seed(99)
vh = nn.Sequential(nn.Linear(exp.width, exp.width), nn.SiLU(), nn.Linear(exp.width, exp.width), nn.SiLU(), nn.Linear(exp.width, 1), nn.Tanh())
for p_ in vh.parameters():
    p_.data = torch.randn(p_.shape) * 0.06
    p_.requires_grad_(False)
g6 = torch.Generator().manual_seed(6)
extra = {k: torch.rand(2, 3, sc.image_size, sc.image_size, generator=g6) * 2 - 1 for k in ("right_wrist_-1_rgb", "base_-1_rgb", "left_wrist_-1_rgb")}
ims6 = {"left_wrist_0_rgb": obs.images["left_wrist_0_rgb"], **extra, "base_0_rgb": obs.images["base_0_rgb"],
        "right_wrist_0_rgb": obs.images["right_wrist_0_rgb"]}
progress = torch.tensor([0.35, -1.7])
obs6 = types.SimpleNamespace(images=ims6, image_masks={k: torch.ones(2, dtype=torch.bool) for k in ims6}, state=obs.state,
                             tokenized_prompt=obs.tokenized_prompt, tokenized_prompt_mask=obs.tokenized_prompt_mask,
                             token_ar_mask=None, token_loss_mask=None, progress=progress, frame_index=None, episode_length=None,
                             image_original=None, episode_index=None)  # fmt: skip
cns = B.base_ns()
cns.update({"image_tools": types.SimpleNamespace(resize_with_pad_torch=None), "logger": logging.getLogger("ref"),
            "Sequence": __import__("collections.abc").abc.Sequence, "IMAGE_RESOLUTION": (sc.image_size, sc.image_size)})
B.lift("/root/reference/src/openpi/models_pytorch/preprocessing_pytorch.py", ["preprocess_observation_pytorch_custom"], cns)
ans = B.base_ns()
ans.update({k: mns[k] for k in ("create_sinusoidal_pos_embedding", "make_att_2d_masks")})
ans["_preprocessing"] = types.SimpleNamespace(preprocess_observation_pytorch_custom=functools.partial(
    cns["preprocess_observation_pytorch_custom"], image_resolution=(sc.image_size, sc.image_size)))
est = types.SimpleNamespace(**pol.__dict__)
est.value_head, est.loss_value_weight, est.loss_action_weight, est.training = vh, 0.7, 1.3, False
for name in ("_apply_checkpoint", "_prepare_attention_masks_4d", "embed_prefix", "embed_suffix"):
    setattr(est, name, functools.partial(B.lift_method(f"{REF}/pi0_pytorch.py", "PI0Pytorch", name, ans), est))
for name in ("_preprocess_observation", "forward", "sample_values"):
    setattr(est, name, functools.partial(B.lift_method(f"{REF}/pi0_pytorch.py", "AdvantageEstimator", name, ans), est))
sv_noise = torch.randn(2, ocfg.action_horizon, ocfg.action_dim, generator=g6)
sv_time = torch.tensor([0.62, 0.11])
est.sample_noise = lambda shape, device: sv_noise.clone()
est.sample_time = lambda bsize, device: sv_time.clone()
with torch.no_grad():
    ae_loss, ae_aux = est.forward(obs6, actions, noise=noise, time=time, return_loss_dict=True)
    ae_values = est.sample_values(torch.device("cpu"), obs6)
oest = O.OracleAdvantageEstimator(ocfg, loss_value_weight=0.7, loss_action_weight=1.3)
oest.load_state_dict({**sd, **{"value_head." + k: v for k, v in vh.state_dict().items()}}, strict=True)
with torch.no_grad():
    o_loss6, o_aux = oest(obs6, actions, noise, time, return_loss_dict=True)
    o_val = oest.sample_values(obs6, sv_noise, sv_time)
print("advantage estimator, oracle vs reference: loss max|d|", float((o_loss6 - ae_loss).abs().max()), " values max|d|",
      float((o_val - ae_values).abs().max()), "| values", ae_values.flatten().tolist())
AE.update({"ae.loss": ae_loss, "ae.values": ae_values, "ae.loss_action": ae_aux["loss_action"].reshape(1),
           "ae.loss_value": ae_aux["loss_value"].reshape(1), "ae.progress": progress, "ae.sv_noise": sv_noise, "ae.sv_time": sv_time,
           **{"ae.img." + k: v for k, v in extra.items()}, **{"ae.value_head." + k: v for k, v in vh.state_dict().items()}})

print("reference loss", tuple(ref_loss.shape), float(ref_loss.mean()), "| actions", tuple(ref_actions.shape), float(ref_actions.abs().mean()))
with torch.no_grad():
    o_loss = oracle(obs, actions, noise, time)
    o_act = oracle.sample_actions(obs, noise.clone(), num_steps=10)
print("oracle vs reference: loss max|d|", float((o_loss - ref_loss).abs().max()), " actions max|d|", float((o_act - ref_actions).abs().max()))
save_file({**{k: v.contiguous() for k, v in ref_grads.items()}, **{k: v.contiguous() for k, v in AE.items()}, "loss": ref_loss.contiguous(), "actions": ref_actions.contiguous(), "image_features_cam0": feats.contiguous(),
           "noise": noise.contiguous(), "time": time.contiguous(), "in_actions": actions.contiguous()},
          os.path.join(HERE, "reference_e2e.safetensors"),
          metadata={"config": "tests/tiny.tiny_cfgs()", "weights": "oracle.synthetic_weights_(seed=0), matrices x4 (std 0.08)",
                    "batch": "oracle.synthetic_batch(cfg, 2, seed=0)", "num_steps": "10"})  # fmt: skip
print("wrote reference_e2e.safetensors")
