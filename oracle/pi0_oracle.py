"""CPU ORACLE (test infrastructure only) — a pure-torch restatement of the reference pi0.5 PyTorch path.

THIS FILE IS NOT PART OF THE PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import it; nothing under kai0_amd/ does.  It exists to check the HIP path, never to serve it.

Parity status: the reference ships no golden vectors and no test of its PyTorch model (SURVEY.md §4/§8c:
model_test.py:12-24 asserts shapes only), and the package cannot be imported in this image (python 3.10 < 3.11,
transformers 5.x != patched 4.53.2, no jax/flax).  What pins this restatement to the reference instead:
  * **vectors produced by executing the reference's own code** (tests/golden/make_reference_blocks_golden.py lifts the
    definitions out of the reference source with `ast` and runs them in the build container; fixture
    tests/golden/reference_blocks.safetensors; checked bit-exactly in bf16 by tests/test_reference_blocks_cpu.py):
    make_att_2d_masks + position ids, create_sinusoidal_pos_embedding, GemmaRMSNorm (plain and adaptive),
    apply_rotary_pos_emb, eager_attention_forward, _gated_residual, GemmaMLP, a whole GemmaDecoderLayer (expert with
    adaRMS + cached K/V; plain prefix layer), a whole SiglipEncoderLayer, the JOINT prefix+expert forward of
    PaliGemmaWithExpertModel (gemma_pytorch.py:126-279, two layers, padded prompt, prefix-LM mask) and
    PI0Pytorch.embed_suffix (pi0.5 branch);
  * the integer/boolean logic additionally against the hand-worked examples of the reference docstring;
  * the state-dict key set against SURVEY.md §8a16.
  * **end to end**: tests/golden/make_reference_e2e_golden.py assembles the WHOLE path from lifted reference code on stub
    `self` objects (SiglipVisionTransformer, get_image_features, GemmaModel.forward with a KV cache, all three branches of
    PaliGemmaWithExpertModel.forward, PI0Pytorch.embed_prefix / embed_suffix / forward / denoise_step / sample_actions) and
    runs it on the tiny test configuration with this oracle's synthetic weights and batch: the reference's loss tensor,
    10-step action chunk and parameter gradients equal this oracle's BIT FOR BIT (tests/golden/reference_e2e.safetensors).
Restated rather than executed (un-vendored third parties): transformers' DynamicCache.update (append),
create_causal_mask (4-D pass-through), the default rotary inv_freq, ACT2FN["gelu_pytorch_tanh"]; and observation
preprocessing, which is the identity for train=False at native resolution.

Module tree and parameter names mirror the reference exactly so a state_dict moves between this oracle and
the HIP model unchanged:
  paligemma_with_expert.paligemma.model.{vision_tower.vision_model, multi_modal_projector, language_model}
  paligemma_with_expert.paligemma.lm_head (tied), paligemma_with_expert.gemma_expert.{model, lm_head}
  action_in_proj, action_out_proj, time_mlp_in, time_mlp_out
"""

from __future__ import annotations

import dataclasses
import math

import torch
import torch.nn.functional as F  # noqa: N812
from torch import nn

PALIGEMMA_VOCAB_SIZE = 257_152  # models/gemma.py:40
MASK_VALUE = -2.3819763e38  # pi0_pytorch.py:159


# ------------------------------------------------------------------------------------------------ configs
@dataclasses.dataclass
class GemmaCfg:  # models/gemma.py:43-52
    width: int
    depth: int
    mlp_dim: int
    num_heads: int
    num_kv_heads: int
    head_dim: int


def get_gemma_config(variant: str) -> GemmaCfg:  # models/gemma.py:58-87
    if variant == "dummy":
        return GemmaCfg(64, 4, 128, 8, 1, 16)
    if variant == "gemma_300m":
        return GemmaCfg(1024, 18, 4096, 8, 1, 256)
    if variant == "gemma_2b":
        return GemmaCfg(2048, 18, 16_384, 8, 1, 256)
    raise ValueError(f"Unknown variant: {variant}")


@dataclasses.dataclass
class SiglipCfg:
    """SigLIP So400m/14 defaults (models/siglip.py:318-363; gemma_pytorch.py:38-41; HF SiglipVisionConfig).
    Tests shrink it; the reference PyTorch path always builds the full tower."""

    hidden_size: int = 1152
    num_layers: int = 27
    num_heads: int = 16
    intermediate_size: int = 4304
    patch_size: int = 14
    image_size: int = 224
    projection_dim: int = 2048
    layer_norm_eps: float = 1e-6


@dataclasses.dataclass
class OracleConfig:  # models/pi0_config.py:19-40
    dtype: str = "bfloat16"
    paligemma_variant: str = "gemma_2b"
    action_expert_variant: str = "gemma_300m"
    action_dim: int = 32
    action_horizon: int = 50
    max_token_len: int = 200
    pi05: bool = True
    vocab_size: int = PALIGEMMA_VOCAB_SIZE
    siglip: SiglipCfg = dataclasses.field(default_factory=SiglipCfg)


# ----------------------------------------------------------------------------------------- layer library
class GemmaRMSNorm(nn.Module):  # modeling_gemma.py:49-104
    def __init__(self, dim: int, eps: float = 1e-6, cond_dim: int | None = None):
        super().__init__()
        self.eps, self.dim, self.cond_dim = eps, dim, cond_dim
        if cond_dim is not None:
            self.dense = nn.Linear(cond_dim, dim * 3, bias=True)
            nn.init.zeros_(self.dense.weight)
        else:
            self.weight = nn.Parameter(torch.zeros(dim))
            self.dense = None

    def forward(self, x, cond=None):
        dtype = x.dtype
        var = torch.mean(torch.square(x.float()), dim=-1, keepdim=True)  # :66-68
        normed = x * torch.rsqrt(var + self.eps)  # bf16 * f32 -> f32  (:70)
        if cond is None or self.dense is None:
            normed = normed * (1.0 + self.weight.float())  # :79
            return normed.to(dtype), None
        modulation = self.dense(cond)  # :89
        if x.dim() == 3:
            modulation = modulation.unsqueeze(1)
        scale, shift, gate = torch.chunk(modulation, 3, dim=-1)  # :94
        normed = normed * (1 + scale.to(torch.float32)) + shift.to(torch.float32)  # :103
        return normed.to(dtype), gate.to(dtype)


class GemmaMLP(nn.Module):  # modeling_gemma.py:113-126
    def __init__(self, width: int, mlp_dim: int):
        super().__init__()
        self.gate_proj = nn.Linear(width, mlp_dim, bias=False)
        self.up_proj = nn.Linear(width, mlp_dim, bias=False)
        self.down_proj = nn.Linear(mlp_dim, width, bias=False)

    def forward(self, x):
        return self.down_proj(F.gelu(self.gate_proj(x), approximate="tanh") * self.up_proj(x))


class GemmaAttention(nn.Module):  # modeling_gemma.py:256-280
    def __init__(self, cfg: GemmaCfg):
        super().__init__()
        self.head_dim = cfg.head_dim
        self.num_heads = cfg.num_heads
        self.num_key_value_groups = cfg.num_heads // cfg.num_kv_heads
        self.scaling = cfg.head_dim**-0.5
        self.q_proj = nn.Linear(cfg.width, cfg.num_heads * cfg.head_dim, bias=False)
        self.k_proj = nn.Linear(cfg.width, cfg.num_kv_heads * cfg.head_dim, bias=False)
        self.v_proj = nn.Linear(cfg.width, cfg.num_kv_heads * cfg.head_dim, bias=False)
        self.o_proj = nn.Linear(cfg.num_heads * cfg.head_dim, cfg.width, bias=False)


class GemmaDecoderLayer(nn.Module):  # modeling_gemma.py:332-342
    def __init__(self, cfg: GemmaCfg, cond_dim: int | None):
        super().__init__()
        self.self_attn = GemmaAttention(cfg)
        self.mlp = GemmaMLP(cfg.width, cfg.mlp_dim)
        self.input_layernorm = GemmaRMSNorm(cfg.width, cond_dim=cond_dim)
        self.post_attention_layernorm = GemmaRMSNorm(cfg.width, cond_dim=cond_dim)


def rope_inv_freq(head_dim: int, base: float = 10000.0) -> torch.Tensor:
    """ROPE_INIT_FUNCTIONS["default"] of transformers 4.53.2 (un-vendored; modeling_gemma.py:141-143)."""
    return 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(dtype=torch.float) / head_dim))


def rotary_cos_sin(inv_freq, position_ids, dtype):  # modeling_gemma.py:149-162
    inv = inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
    pos = position_ids[:, None, :].float()
    freqs = (inv.float() @ pos.float()).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):  # modeling_gemma.py:165-169
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rope(q, k, cos, sin):  # modeling_gemma.py:172-194 (unsqueeze_dim=1)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def repeat_kv(x, n_rep):  # modeling_gemma.py:197-206
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def eager_attention(q, k, v, mask, scaling, n_rep):  # modeling_gemma.py:230-253
    k = repeat_kv(k, n_rep)
    v = repeat_kv(v, n_rep)
    w = torch.matmul(q, k.transpose(2, 3)) * scaling
    if mask is not None:
        w = w + mask[:, :, :, : k.shape[-2]]
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    out = torch.matmul(w, v)
    return out.transpose(1, 2).contiguous()


def gated_residual(x, y, gate):  # modeling_gemma.py:209-227
    if gate is None:
        return x + y
    return x + y * gate


class GemmaModel(nn.Module):  # modeling_gemma.py:419-433
    def __init__(self, cfg: GemmaCfg, vocab: int, use_adarms: bool, with_embed: bool):
        super().__init__()
        self.cfg = cfg
        cond_dim = cfg.width if use_adarms else None
        # gemma_pytorch.py:59 sets gemma_expert.model.embed_tokens = None (drops the key)
        self.embed_tokens = nn.Embedding(vocab, cfg.width) if with_embed else None
        self.layers = nn.ModuleList([GemmaDecoderLayer(cfg, cond_dim) for _ in range(cfg.depth)])
        self.norm = GemmaRMSNorm(cfg.width, cond_dim=cond_dim)
        self.register_buffer("inv_freq", rope_inv_freq(cfg.head_dim), persistent=False)

    def forward_single(self, inputs_embeds, attention_mask, position_ids, past_key_values, use_cache, adarms_cond):
        """GemmaModel.forward / GemmaDecoderLayer.forward / GemmaAttention.forward for one expert
        (modeling_gemma.py:446-555, 344-384, 282-329).  past_key_values: list of (k, v) or None."""
        h = inputs_embeds
        if self.layers[0].self_attn.q_proj.weight.dtype == torch.bfloat16:  # :506-507
            h = h.to(torch.bfloat16)
        cos, sin = rotary_cos_sin(self.inv_freq, position_ids, h.dtype)  # :510
        new_cache = [] if use_cache else None
        for li, layer in enumerate(self.layers):
            residual = h
            x, gate = layer.input_layernorm(h, adarms_cond)
            at = layer.self_attn
            shp = (*x.shape[:-1], -1, at.head_dim)
            q = at.q_proj(x).view(shp).transpose(1, 2)
            k = at.k_proj(x).view(shp).transpose(1, 2)
            v = at.v_proj(x).view(shp).transpose(1, 2)
            q, k = apply_rope(q, k, cos, sin)
            if use_cache:  # DynamicCache.update == append on the sequence dim (:303-307)
                new_cache.append((k, v))
            elif past_key_values is not None:  # :308-310
                k = torch.cat([past_key_values[li][0], k], dim=2)
                v = torch.cat([past_key_values[li][1], v], dim=2)
            a = eager_attention(q, k, v, attention_mask, at.scaling, at.num_key_value_groups)
            a = a.reshape(*x.shape[:-1], -1).contiguous()
            a = at.o_proj(a)
            h = gated_residual(residual, a, gate)
            residual = h
            x, gate = layer.post_attention_layernorm(h, adarms_cond)
            x = layer.mlp(x)
            h = gated_residual(residual, x, gate)
        h, _ = self.norm(h, adarms_cond)
        return h, new_cache


class GemmaForCausalLM(nn.Module):  # modeling_gemma.py:567-571 (only the containers)
    def __init__(self, cfg: GemmaCfg, vocab: int, use_adarms: bool):
        super().__init__()
        self.model = GemmaModel(cfg, vocab, use_adarms, with_embed=False)
        self.lm_head = nn.Linear(cfg.width, vocab, bias=False)  # dead weight kept for the state-dict contract


# ------------------------------------------------------------------------------------------------ SigLIP
class SiglipVisionEmbeddings(nn.Module):  # modeling_siglip.py:212-281
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.patch_embedding = nn.Conv2d(3, c.hidden_size, kernel_size=c.patch_size, stride=c.patch_size, padding="valid")
        self.num_patches = (c.image_size // c.patch_size) ** 2
        self.position_embedding = nn.Embedding(self.num_patches, c.hidden_size)
        self.register_buffer("position_ids", torch.arange(self.num_patches).expand((1, -1)), persistent=False)

    def forward(self, pixel_values):
        target_dtype = self.patch_embedding.weight.dtype
        patch = self.patch_embedding(pixel_values.to(dtype=target_dtype))
        emb = patch.flatten(2).transpose(1, 2)
        return emb + self.position_embedding(self.position_ids)


class SiglipAttention(nn.Module):  # modeling_siglip.py:348-417
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.num_heads = c.num_heads
        self.head_dim = c.hidden_size // c.num_heads
        self.scale = self.head_dim**-0.5
        self.k_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.v_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.q_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.out_proj = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, x):
        b, s, e = x.shape
        q = self.q_proj(x).view(b, s, self.num_heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(x).view(b, s, self.num_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(x).view(b, s, self.num_heads, self.head_dim).transpose(1, 2)
        w = torch.matmul(q, k.transpose(-1, -2)) * self.scale  # :334
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)  # :339
        o = torch.matmul(w, v).transpose(1, 2).contiguous()
        return self.out_proj(o.reshape(b, s, e).contiguous())


class SiglipMLP(nn.Module):  # modeling_siglip.py:420-432
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.fc1 = nn.Linear(c.hidden_size, c.intermediate_size)
        self.fc2 = nn.Linear(c.intermediate_size, c.hidden_size)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x), approximate="tanh"))


class SiglipEncoderLayer(nn.Module):  # modeling_siglip.py:435-480
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.self_attn = SiglipAttention(c)
        self.layer_norm2 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp = SiglipMLP(c)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class SiglipEncoder(nn.Module):
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.layers = nn.ModuleList([SiglipEncoderLayer(c) for _ in range(c.num_layers)])


class SiglipVisionTransformer(nn.Module):  # modeling_siglip.py:748-796 (vision_use_head=False)
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.embeddings = SiglipVisionEmbeddings(c)
        self.encoder = SiglipEncoder(c)
        self.post_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)

    def forward(self, pixel_values):
        h = self.embeddings(pixel_values)
        if self.encoder.layers[0].self_attn.q_proj.weight.dtype == torch.bfloat16:  # :777-778
            h = h.to(torch.bfloat16)
        for layer in self.encoder.layers:
            h = layer(h)
        return self.post_layernorm(h)


class SiglipVisionModel(nn.Module):
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.vision_model = SiglipVisionTransformer(c)


class PaliGemmaMultiModalProjector(nn.Module):  # modeling_paligemma.py:91-99
    def __init__(self, c: SiglipCfg):
        super().__init__()
        self.linear = nn.Linear(c.hidden_size, c.projection_dim, bias=True)


class PaliGemmaModel(nn.Module):  # modeling_paligemma.py:138-145
    def __init__(self, cfg: GemmaCfg, vocab: int, sc: SiglipCfg):
        super().__init__()
        self.vision_tower = SiglipVisionModel(sc)
        self.multi_modal_projector = PaliGemmaMultiModalProjector(sc)
        self.language_model = GemmaModel(cfg, vocab, use_adarms=False, with_embed=True)

    def get_image_features(self, pixel_values):  # :232-245 (no 1/sqrt(D) scaling in the patched file)
        feats = self.vision_tower.vision_model(pixel_values)
        return self.multi_modal_projector.linear(feats)


class PaliGemmaForConditionalGeneration(nn.Module):  # modeling_paligemma.py:389-392
    def __init__(self, cfg: GemmaCfg, vocab: int, sc: SiglipCfg):
        super().__init__()
        self.model = PaliGemmaModel(cfg, vocab, sc)
        self.lm_head = nn.Linear(cfg.width, vocab, bias=False)
        self.lm_head.weight = self.model.language_model.embed_tokens.weight  # tied (post_init)

    @property
    def language_model(self):
        return self.model.language_model


KEEP_F32_SELECTORS = (  # gemma_pytorch.py:72-79
    "vision_tower.vision_model.embeddings.patch_embedding.weight",
    "vision_tower.vision_model.embeddings.patch_embedding.bias",
    "vision_tower.vision_model.embeddings.position_embedding.weight",
    "input_layernorm",
    "post_attention_layernorm",
    "model.norm",
)


class PaliGemmaWithExpertModel(nn.Module):  # gemma_pytorch.py:12-281
    def __init__(self, vlm: GemmaCfg, expert: GemmaCfg, use_adarms, precision: str, vocab: int, sc: SiglipCfg):
        super().__init__()
        assert not use_adarms[0]
        self.paligemma = PaliGemmaForConditionalGeneration(vlm, vocab, sc)
        self.gemma_expert = GemmaForCausalLM(expert, vocab, use_adarms[1])
        self.num_heads = vlm.num_heads
        self.to_bfloat16_for_selected_params(precision)

    def to_bfloat16_for_selected_params(self, precision: str = "bfloat16"):  # :63-83
        if precision == "float32":
            self.to(dtype=torch.float32)
            return
        if precision != "bfloat16":
            raise ValueError(f"Invalid precision: {precision}")
        self.to(dtype=torch.bfloat16)
        for name, param in self.named_parameters():
            if any(sel in name for sel in KEEP_F32_SELECTORS):
                param.data = param.data.to(dtype=torch.float32)

    def embed_image(self, image):
        return self.paligemma.model.get_image_features(image)

    def embed_language_tokens(self, tokens):
        return self.paligemma.language_model.embed_tokens(tokens)

    def forward(self, attention_mask, position_ids, past_key_values, inputs_embeds, use_cache=False, adarms_cond=None):
        if adarms_cond is None:
            adarms_cond = [None, None]
        if inputs_embeds[1] is None:  # prefix only (:102-113)
            out, cache = self.paligemma.language_model.forward_single(
                inputs_embeds[0], attention_mask, position_ids, past_key_values, use_cache, adarms_cond[0]
            )
            return [out, None], cache
        if inputs_embeds[0] is None:  # suffix only with cached prefix KV (:114-125)
            out, _ = self.gemma_expert.model.forward_single(
                inputs_embeds[1], attention_mask, position_ids, past_key_values, use_cache, adarms_cond[1]
            )
            return [None, out], None
        # joint, layer-interleaved (:126-279)
        models = [self.paligemma.language_model, self.gemma_expert.model]
        for li in range(len(models[0].layers)):
            inputs_embeds = self._joint_layer(li, models, inputs_embeds, attention_mask, position_ids, adarms_cond)
        outs = []
        for i, h in enumerate(inputs_embeds):  # final norms (:262-275)
            o, _ = models[i].norm(h, cond=adarms_cond[i])
            outs.append(o)
        return outs, None

    def _joint_layer(self, li, models, inputs_embeds, attention_mask, position_ids, adarms_cond):  # :160-237
        qs, ks, vs, gates = [], [], [], []
        for i, h in enumerate(inputs_embeds):
            layer = models[i].layers[li]
            x, gate = layer.input_layernorm(h, cond=adarms_cond[i])
            gates.append(gate)
            shp = (*x.shape[:-1], -1, layer.self_attn.head_dim)
            qs.append(layer.self_attn.q_proj(x).view(shp).transpose(1, 2))
            ks.append(layer.self_attn.k_proj(x).view(shp).transpose(1, 2))
            vs.append(layer.self_attn.v_proj(x).view(shp).transpose(1, 2))
        q, k, v = torch.cat(qs, dim=2), torch.cat(ks, dim=2), torch.cat(vs, dim=2)
        cos, sin = rotary_cos_sin(models[0].inv_freq, position_ids, q.dtype)  # rotary of the 2B model (:192)
        q, k = apply_rope(q, k, cos, sin)
        at0 = models[0].layers[li].self_attn
        att = eager_attention(q, k, v, attention_mask, at0.scaling, at0.num_key_value_groups)
        bsz = q.shape[0]
        att = att.reshape(bsz, -1, self.num_heads * at0.head_dim)  # `1 * 8 * head_dim` (:212)
        outs, start = [], 0
        for i, h in enumerate(inputs_embeds):
            layer = models[i].layers[li]
            end = start + h.shape[1]
            if att.dtype != layer.self_attn.o_proj.weight.dtype:
                att = att.to(layer.self_attn.o_proj.weight.dtype)
            o = layer.self_attn.o_proj(att[:, start:end])
            o = gated_residual(h, o, gates[i])
            after_first = o.clone()
            o, gate = layer.post_attention_layernorm(o, cond=adarms_cond[i])
            if layer.mlp.up_proj.weight.dtype == torch.bfloat16:
                o = o.to(dtype=torch.bfloat16)
            o = layer.mlp(o)
            outs.append(gated_residual(after_first, o, gate))
            start = end
        return outs


# ---------------------------------------------------------------------------------------------- PI0 model
def create_sinusoidal_pos_embedding(time, dimension, min_period, max_period):  # pi0_pytorch.py:25-42
    if dimension % 2 != 0:
        raise ValueError(f"dimension ({dimension}) must be divisible by 2")
    if time.ndim != 1:
        raise ValueError("The time tensor is expected to be of shape `(batch_size, )`.")
    fraction = torch.linspace(0.0, 1.0, dimension // 2, dtype=torch.float64, device=time.device)
    period = min_period * (max_period / min_period) ** fraction
    scaling = 1.0 / period * 2 * math.pi
    sin_input = scaling[None, :] * time[:, None]
    return torch.cat([torch.sin(sin_input), torch.cos(sin_input)], dim=1)


def make_att_2d_masks(pad_masks, att_masks):  # pi0_pytorch.py:52-81
    if att_masks.ndim != 2:
        raise ValueError(att_masks.ndim)
    if pad_masks.ndim != 2:
        raise ValueError(pad_masks.ndim)
    cumsum = torch.cumsum(att_masks, dim=1)
    att_2d = cumsum[:, None, :] <= cumsum[:, :, None]
    pad_2d = pad_masks[:, None, :] * pad_masks[:, :, None]
    return att_2d & pad_2d


def masks_4d(att_2d):  # pi0_pytorch.py:156-159
    return torch.where(att_2d[:, None, :, :], 0.0, MASK_VALUE)


IMAGE_KEYS = ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb")  # preprocessing_pytorch.py:11-15


class OraclePI0(nn.Module):  # pi0_pytorch.py:84-461 (pi05=True branch only; train aug is off => train=False path)
    def __init__(self, config: OracleConfig):
        super().__init__()
        assert config.pi05, "only the pi0.5 branch is restated"
        self.config = config
        vlm = get_gemma_config(config.paligemma_variant)
        exp = get_gemma_config(config.action_expert_variant)
        self.paligemma_with_expert = PaliGemmaWithExpertModel(
            vlm, exp, use_adarms=[False, True], precision=config.dtype, vocab=config.vocab_size, sc=config.siglip
        )
        self.action_in_proj = nn.Linear(config.action_dim, exp.width)
        self.action_out_proj = nn.Linear(exp.width, config.action_dim)
        self.time_mlp_in = nn.Linear(exp.width, exp.width)
        self.time_mlp_out = nn.Linear(exp.width, exp.width)

    # -- observation plumbing (preprocessing_pytorch.py:20-173 with train=False, images already at resolution)
    @staticmethod
    def _unpack(obs):
        images = [obs.images[k] for k in IMAGE_KEYS]
        masks = [obs.image_masks[k] for k in IMAGE_KEYS]
        return images, masks, obs.tokenized_prompt, obs.tokenized_prompt_mask

    def embed_prefix(self, images, img_masks, lang_tokens, lang_masks):  # :186-235
        embs, pads, att = [], [], []
        for img, m in zip(images, img_masks, strict=True):
            e = self.paligemma_with_expert.embed_image(img)
            b, n = e.shape[:2]
            embs.append(e)
            pads.append(m[:, None].expand(b, n))
            att += [0] * n
        le = self.paligemma_with_expert.embed_language_tokens(lang_tokens)
        le = le * math.sqrt(le.shape[-1])  # :215-216
        embs.append(le)
        pads.append(lang_masks)
        att += [0] * le.shape[1]
        embs = torch.cat(embs, dim=1)
        pads = torch.cat(pads, dim=1)
        att = torch.tensor(att, dtype=torch.bool, device=pads.device)
        return embs, pads, att[None, :].expand(pads.shape[0], len(att))

    def embed_suffix(self, noisy_actions, timestep):  # :237-314 (pi05 branch)
        width = self.action_in_proj.out_features
        te = create_sinusoidal_pos_embedding(timestep, width, min_period=4e-3, max_period=4.0)
        te = te.type(dtype=timestep.dtype)
        action_emb = self.action_in_proj(noisy_actions)
        x = F.silu(self.time_mlp_in(te))
        adarms_cond = F.silu(self.time_mlp_out(x))
        b, n = action_emb.shape[:2]
        pad = torch.ones(b, n, dtype=torch.bool, device=timestep.device)
        att = torch.tensor([1] + [0] * (self.config.action_horizon - 1), dtype=action_emb.dtype, device=action_emb.device)
        return action_emb, pad, att[None, :].expand(b, n), adarms_cond

    def _is_bf16(self):
        return self.paligemma_with_expert.paligemma.language_model.layers[0].self_attn.q_proj.weight.dtype == torch.bfloat16

    def forward(self, observation, actions, noise, time):  # :316-373 (noise/time injected; no augmentation)
        images, img_masks, lang_tokens, lang_masks = self._unpack(observation)
        t = time[:, None, None]
        x_t = t * noise + (1 - t) * actions
        u_t = noise - actions
        pe, ppad, patt = self.embed_prefix(images, img_masks, lang_tokens, lang_masks)
        se, spad, satt, cond = self.embed_suffix(x_t, time)
        if self._is_bf16():
            se = se.to(dtype=torch.bfloat16)
            pe = pe.to(dtype=torch.bfloat16)
        pad = torch.cat([ppad, spad], dim=1)
        att = torch.cat([patt, satt], dim=1)
        att_2d = make_att_2d_masks(pad, att)
        position_ids = torch.cumsum(pad, dim=1) - 1
        (_, suffix_out), _ = self.paligemma_with_expert.forward(
            attention_mask=masks_4d(att_2d), position_ids=position_ids, past_key_values=None,
            inputs_embeds=[pe, se], use_cache=False, adarms_cond=[None, cond],
        )  # fmt: skip
        suffix_out = suffix_out[:, -self.config.action_horizon :].to(dtype=torch.float32)
        v_t = self.action_out_proj(suffix_out)
        return F.mse_loss(u_t, v_t, reduction="none")

    @torch.no_grad()
    def sample_actions(self, observation, noise, num_steps: int = 10):  # :375-419
        images, img_masks, lang_tokens, lang_masks = self._unpack(observation)
        bsize = noise.shape[0]
        pe, ppad, patt = self.embed_prefix(images, img_masks, lang_tokens, lang_masks)
        p2d = make_att_2d_masks(ppad, patt)
        ppos = torch.cumsum(ppad, dim=1) - 1
        _, cache = self.paligemma_with_expert.forward(
            attention_mask=masks_4d(p2d), position_ids=ppos, past_key_values=None, inputs_embeds=[pe, None], use_cache=True
        )
        dt = torch.tensor(-1.0 / num_steps, dtype=torch.float32, device=noise.device)
        x_t = noise
        time = torch.tensor(1.0, dtype=torch.float32, device=noise.device)
        while time >= -dt / 2:
            v_t = self.denoise_step(ppad, cache, x_t, time.expand(bsize))
            x_t = x_t + dt * v_t
            time = time + dt
        return x_t

    def denoise_step(self, prefix_pad_masks, past_key_values, x_t, timestep):  # :421-461
        se, spad, satt, cond = self.embed_suffix(x_t, timestep)
        b, plen = prefix_pad_masks.shape
        slen = spad.shape[1]
        p2d = prefix_pad_masks[:, None, :].expand(b, slen, plen)
        s2d = make_att_2d_masks(spad, satt)
        full = torch.cat([p2d, s2d], dim=2)
        offsets = torch.sum(prefix_pad_masks, dim=-1)[:, None]
        position_ids = offsets + torch.cumsum(spad, dim=1) - 1
        outs, _ = self.paligemma_with_expert.forward(
            attention_mask=masks_4d(full), position_ids=position_ids, past_key_values=past_key_values,
            inputs_embeds=[None, se], use_cache=False, adarms_cond=[None, cond],
        )  # fmt: skip
        suffix_out = outs[1][:, -self.config.action_horizon :].to(dtype=torch.float32)
        return self.action_out_proj(suffix_out)


# ------------------------------------------------------------------------------------ synthetic data / weights
class OracleAdvantageEstimator(OraclePI0):  # pi0_pytorch.py:464-644
    def __init__(self, config: OracleConfig, loss_value_weight: float = 0.0, loss_action_weight: float = 1.0):
        super().__init__(config)
        self.loss_value_weight, self.loss_action_weight = loss_value_weight, loss_action_weight
        w = get_gemma_config(config.action_expert_variant).width
        self.value_head = nn.Sequential(nn.Linear(w, w), nn.SiLU(), nn.Linear(w, w), nn.SiLU(), nn.Linear(w, 1), nn.Tanh())  # :471-481

    @staticmethod
    def _unpack_sorted(obs):  # preprocessing_pytorch.py:193-202 (apply_aug=False, native resolution => identity otherwise)
        order = {"base": 0, "left_wrist": 1, "right_wrist": 2}

        def key(k):
            part, ts, _ = k.rsplit("_", 2)
            return int(ts), order[part]

        keys = sorted(obs.images.keys(), key=key)
        b = obs.state.shape[:-1]
        masks = [obs.image_masks[k] if k in obs.image_masks else torch.ones(b, dtype=torch.bool) for k in keys]
        return [obs.images[k] for k in keys], masks, obs.tokenized_prompt, obs.tokenized_prompt_mask

    def _suffix_out(self, observation, x_t, time):  # the shared middle of forward (:515-552) and sample_values (:610-636)
        images, img_masks, lang_tokens, lang_masks = self._unpack_sorted(observation)
        prefix_embs, prefix_pad, prefix_att = self.embed_prefix(images, img_masks, lang_tokens, lang_masks)
        suffix_embs, suffix_pad, suffix_att, adarms_cond = self.embed_suffix(x_t, time)
        if self._is_bf16():
            suffix_embs, prefix_embs = suffix_embs.to(torch.bfloat16), prefix_embs.to(torch.bfloat16)
        pad = torch.cat([prefix_pad, suffix_pad], dim=1)
        att = torch.cat([prefix_att, suffix_att], dim=1)
        att_2d = make_att_2d_masks(pad, att)
        position_ids = torch.cumsum(pad, dim=1) - 1
        (_, suffix_out), _ = self.paligemma_with_expert(masks_4d(att_2d), position_ids, None, [prefix_embs, suffix_embs], False,
                                                        [None, adarms_cond])  # fmt: skip
        return suffix_out

    def forward(self, observation, actions, noise, time, return_loss_dict=False):  # :500-592
        te = time[:, None, None]
        x_t = te * noise + (1 - te) * actions
        u_t = noise - actions
        suffix_out_full = self._suffix_out(observation, x_t, time)
        v_t = self.action_out_proj(suffix_out_full[:, -self.config.action_horizon :].to(dtype=torch.float32))
        loss_action = F.mse_loss(u_t, v_t, reduction="none").mean(dim=-1)  # :563
        loss = loss_action * self.loss_action_weight
        value_pred = self.value_head(suffix_out_full[:, 0, :].to(dtype=torch.float32))  # :571-572
        target = torch.clamp(observation.progress.float(), -1.0, 1.0).unsqueeze(1)  # :574-576
        value_loss = F.mse_loss(value_pred, target, reduction="none").to(loss.dtype) * self.loss_value_weight
        aux = {"loss_action": loss_action.detach().mean(), "loss_value": value_loss.detach().mean()}
        loss = loss + value_loss
        return (loss, aux) if return_loss_dict else loss

    @torch.no_grad()
    def sample_values(self, observation, noise, time):  # :596-644 with the sampled noise / time injected
        suffix_out = self._suffix_out(observation, noise, time)
        return self.value_head(suffix_out[:, 0, :].to(dtype=torch.float32))


class SimpleObs:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def synthetic_weights_(model: nn.Module, seed: int = 0) -> None:
    """Explicit seeded weights: N(0, 0.02) linears/embeddings/conv, plain-RMSNorm weights 0,
    LayerNorm weight 1 / bias N(0,0.02), adaRMS dense weight N(0, 0.02) / bias N(0,0.02) so modulation and
    gates are non-trivial.  Never rely on module init for parity (the reference init sets RMSNorm weight 1.0)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        seen = set()
        for name, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            if name.endswith("layernorm.weight") and p.dim() == 1 and "dense" not in name and "vision" not in name:
                p.zero_()
            elif name.endswith("norm.weight") and "vision" not in name and "dense" not in name:
                p.zero_()
            elif "vision" in name and ("layer_norm" in name or "post_layernorm" in name) and name.endswith("weight"):
                p.copy_((1.0 + 0.02 * torch.randn(p.shape, generator=g)).to(p.dtype))
            else:
                p.copy_((0.02 * torch.randn(p.shape, generator=g, dtype=torch.float32)).to(p.dtype))


def synthetic_batch(cfg: OracleConfig, batch: int, seed: int = 0, device="cpu", image_size: int | None = None):
    """Seeded synthetic observation/actions/noise/time of SURVEY.md §8d."""
    g = torch.Generator().manual_seed(1000 + seed)
    hw = image_size or cfg.siglip.image_size
    images, masks = {}, {}
    for k in IMAGE_KEYS:
        u8 = torch.randint(0, 256, (batch, hw, hw, 3), generator=g, dtype=torch.uint8)
        images[k] = (u8.to(torch.float32) / 255.0 * 2.0 - 1.0).permute(0, 3, 1, 2).contiguous().to(device)  # model.py:132-133
        masks[k] = torch.ones(batch, dtype=torch.bool, device=device)
    L = cfg.max_token_len
    vocab_hi = min(2048, cfg.vocab_size)
    tokens = torch.randint(0, vocab_hi, (batch, L), generator=g, dtype=torch.int64)
    n_valid = (L * 3) // 8 + torch.randint(0, max(1, L // 3), (batch,), generator=g)
    tmask = torch.arange(L)[None, :] < n_valid[:, None]
    state = torch.zeros(batch, cfg.action_dim)
    state[:, :14] = torch.rand(batch, 14, generator=g) * 2 - 1
    actions = torch.zeros(batch, cfg.action_horizon, cfg.action_dim)
    actions[..., :14] = torch.randn(batch, cfg.action_horizon, 14, generator=g)
    noise = torch.randn(batch, cfg.action_horizon, cfg.action_dim, generator=g)
    beta = torch.distributions.Beta(torch.tensor(1.5), torch.tensor(1.0))
    torch.manual_seed(2000 + seed)
    time = beta.sample((batch,)) * 0.999 + 0.001
    obs = SimpleObs(
        images=images, image_masks=masks, state=state.to(device), tokenized_prompt=tokens.to(device),
        tokenized_prompt_mask=tmask.to(device), token_ar_mask=None, token_loss_mask=None,
    )  # fmt: skip
    return obs, actions.to(device), noise.to(device), time.to(torch.float32).to(device)
