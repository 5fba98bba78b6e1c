"""PI0Pytorch-shaped pi0.5 model whose arithmetic runs on the MI355X HIP kernels (libkai0hip.so).

Boundary mirrored: `PI0Pytorch(config)`, `.forward(observation, actions, noise=None, time=None)
-> loss [B, H, A]`, `.sample_actions(device, observation, noise=None, num_steps=10) -> [B, H, A]`,
`.gradient_checkpointing_enable()`, `.paligemma_with_expert.to_bfloat16_for_selected_params(...)`, and an
nn.Module parameter tree whose state_dict keys / shapes / dtypes equal the reference's
(pi0_pytorch.py:84-461, gemma_pytorch.py:12-281; key list in SURVEY.md §8a16).

Layout choices (MI355X-first, not a translation of the HF modules):
  * activations are flat [B*S, D] bf16 matrices so every Linear is one large MFMA GEMM (M = B*S rows);
  * the 3 cameras go through SigLIP as one batch of 3B images;
  * the joint attention folds the 8 query heads of the single KV head into the GEMM M dimension
    (Q viewed as [S*8, 256] per sample), so QK^T / PV are plain batched GEMMs with K/V read once per sample;
  * the prefix-LM mask is never materialised (no [S,S] tensor): two int32 codes per token feed the softmax;
  * 288 GB HBM: full-batch activations of all 45 layers stay resident, so rematerialisation is off by default
    (`gradient_checkpointing_enable()` keeps the reference API and turns it on).
"""

from __future__ import annotations

import dataclasses

import contextlib
import os

import logging
import math

import torch
from torch import nn

from . import _lib, ops
from .config import Pi0Config, SiglipConfig, get_config
from .preprocessing import IMAGE_KEYS, preprocess_observation

BF16, F32 = torch.bfloat16, torch.float32
INT_MAX = 2**31 - 1
logger = logging.getLogger("kai0_amd")


# ------------------------------------------------------------------------------------- parameter containers
class Linear(nn.Module):
    """nn.Linear-shaped parameter holder (weight [out, in], optional bias)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.normal_(self.weight, std=0.02)
        if bias:
            nn.init.zeros_(self.bias)


class Embedding(nn.Module):
    def __init__(self, num: int, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num, dim))
        nn.init.normal_(self.weight, std=0.02)


class Conv2dParams(nn.Module):
    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.normal_(self.weight, std=0.02)


class LayerNormParams(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class GemmaRMSNorm(nn.Module):
    """Parameters of GemmaRMSNorm (modeling_gemma.py:49-64): `weight` (plain) or `dense` (adaRMS)."""

    def __init__(self, dim: int, eps: float = 1e-6, cond_dim: int | None = None):
        super().__init__()
        self.eps, self.dim, self.cond_dim = eps, dim, cond_dim
        if cond_dim is not None:
            self.dense = Linear(cond_dim, dim * 3, bias=True)
            nn.init.zeros_(self.dense.weight)
        else:
            self.weight = nn.Parameter(torch.zeros(dim))
            self.dense = None


class GemmaMLP(nn.Module):
    def __init__(self, width: int, mlp_dim: int):
        super().__init__()
        self.gate_proj = Linear(width, mlp_dim, bias=False)
        self.up_proj = Linear(width, mlp_dim, bias=False)
        self.down_proj = Linear(mlp_dim, width, bias=False)


class GemmaAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.head_dim = cfg.head_dim
        self.num_heads = cfg.num_heads
        self.num_key_value_groups = cfg.num_heads // cfg.num_kv_heads
        self.scaling = cfg.head_dim**-0.5
        self.q_proj = Linear(cfg.width, cfg.num_heads * cfg.head_dim, bias=False)
        self.k_proj = Linear(cfg.width, cfg.num_kv_heads * cfg.head_dim, bias=False)
        self.v_proj = Linear(cfg.width, cfg.num_kv_heads * cfg.head_dim, bias=False)
        self.o_proj = Linear(cfg.num_heads * cfg.head_dim, cfg.width, bias=False)


class GemmaDecoderLayer(nn.Module):
    def __init__(self, cfg, cond_dim):
        super().__init__()
        self.self_attn = GemmaAttention(cfg)
        self.mlp = GemmaMLP(cfg.width, cfg.mlp_dim)
        self.input_layernorm = GemmaRMSNorm(cfg.width, cond_dim=cond_dim)
        self.post_attention_layernorm = GemmaRMSNorm(cfg.width, cond_dim=cond_dim)


class GemmaModel(nn.Module):
    def __init__(self, cfg, vocab: int, use_adarms: bool, with_embed: bool):
        super().__init__()
        self.cfg = cfg
        cond_dim = cfg.width if use_adarms else None
        self.embed_tokens = Embedding(vocab, cfg.width) if with_embed else None
        self.layers = nn.ModuleList([GemmaDecoderLayer(cfg, cond_dim) for _ in range(cfg.depth)])
        self.norm = GemmaRMSNorm(cfg.width, cond_dim=cond_dim)
        self.gradient_checkpointing = False
        # ROPE_INIT_FUNCTIONS["default"] (transformers 4.53.2): 1 / 10000^(arange(0, hd, 2) / hd), f32
        hd = cfg.head_dim
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.int64).to(dtype=torch.float) / hd))
        self.register_buffer("inv_freq", inv, persistent=False)

    def rope_inv_freq(self) -> torch.Tensor:
        """f32 table handed to the RoPE kernel.  NOTE (reference quirk, mirrored on purpose): the reference casts
        the whole module with `self.to(dtype=torch.bfloat16)` (gemma_pytorch.py:64-65), which also rounds the
        non-persistent `inv_freq` buffer to bf16; GemmaRotaryEmbedding.forward then upcasts it with `.float()`
        (modeling_gemma.py:151).  So in bf16 mode the rotary frequencies ARE the bf16-rounded ones."""
        return self.inv_freq.to(torch.float32).contiguous()


class GemmaForCausalLM(nn.Module):
    def __init__(self, cfg, vocab: int, use_adarms: bool):
        super().__init__()
        self.model = GemmaModel(cfg, vocab, use_adarms, with_embed=False)
        # never used by the forward (263 M dead parameters) but part of the reference state dict (§8a16)
        self.lm_head = Linear(cfg.width, vocab, bias=False)


class SiglipVisionEmbeddings(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.patch_size = c.patch_size
        self.patch_embedding = Conv2dParams(3, c.hidden_size, c.patch_size)
        self.num_patches = (c.image_size // c.patch_size) ** 2
        self.position_embedding = Embedding(self.num_patches, c.hidden_size)


class SiglipAttention(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.num_heads = c.num_heads
        self.head_dim = c.hidden_size // c.num_heads
        self.k_proj = Linear(c.hidden_size, c.hidden_size)
        self.v_proj = Linear(c.hidden_size, c.hidden_size)
        self.q_proj = Linear(c.hidden_size, c.hidden_size)
        self.out_proj = Linear(c.hidden_size, c.hidden_size)


class SiglipMLP(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.fc1 = Linear(c.hidden_size, c.intermediate_size)
        self.fc2 = Linear(c.intermediate_size, c.hidden_size)


class SiglipEncoderLayer(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.layer_norm1 = LayerNormParams(c.hidden_size, c.layer_norm_eps)
        self.self_attn = SiglipAttention(c)
        self.layer_norm2 = LayerNormParams(c.hidden_size, c.layer_norm_eps)
        self.mlp = SiglipMLP(c)


class SiglipEncoder(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.layers = nn.ModuleList([SiglipEncoderLayer(c) for _ in range(c.num_layers)])


class SiglipVisionTransformer(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.embeddings = SiglipVisionEmbeddings(c)
        self.encoder = SiglipEncoder(c)
        self.post_layernorm = LayerNormParams(c.hidden_size, c.layer_norm_eps)


class SiglipVisionModel(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.vision_model = SiglipVisionTransformer(c)
        self.gradient_checkpointing = False


class PaliGemmaMultiModalProjector(nn.Module):
    def __init__(self, c: SiglipConfig):
        super().__init__()
        self.linear = Linear(c.hidden_size, c.projection_dim, bias=True)


class PaliGemmaModel(nn.Module):
    def __init__(self, cfg, vocab: int, sc: SiglipConfig):
        super().__init__()
        self.vision_tower = SiglipVisionModel(sc)
        self.multi_modal_projector = PaliGemmaMultiModalProjector(sc)
        self.language_model = GemmaModel(cfg, vocab, use_adarms=False, with_embed=True)


class PaliGemmaForConditionalGeneration(nn.Module):
    def __init__(self, cfg, vocab: int, sc: SiglipConfig):
        super().__init__()
        self.model = PaliGemmaModel(cfg, vocab, sc)
        self.lm_head = Linear(cfg.width, vocab, bias=False)
        self.lm_head.weight = self.model.language_model.embed_tokens.weight  # tied, as post_init does

    @property
    def language_model(self):
        return self.model.language_model

    @property
    def vision_tower(self):
        return self.model.vision_tower


KEEP_F32_SELECTORS = (  # gemma_pytorch.py:72-79 (substring match; also catches the adaRMS `dense` layers)
    "vision_tower.vision_model.embeddings.patch_embedding.weight",
    "vision_tower.vision_model.embeddings.patch_embedding.bias",
    "vision_tower.vision_model.embeddings.position_embedding.weight",
    "input_layernorm",
    "post_attention_layernorm",
    "model.norm",
)


# ---------------------------------------------------------------------------------------- compute helpers
def _lin(x, mod: Linear, residual=None, act=0):
    return ops.linear(x, mod.weight, mod.bias, residual, act)


def build_mask_codes(pad_masks: torch.Tensor, att_masks: torch.Tensor):
    """Integer restatement of make_att_2d_masks (pi0_pytorch.py:52-81): token j is visible from token i iff
    cumsum(att)[j] <= cumsum(att)[i] and both are valid.  Encoded as kcode[j] <= qcode[i] with
    qcode = pad ? cumsum : -1 and kcode = pad ? cumsum : INT_MAX (bit-exact, no [S,S] tensor).
    Also returns position_ids = cumsum(pad) - 1 (pi0_pytorch.py:343)."""
    cum = torch.cumsum(att_masks.to(torch.int32), dim=1)
    pad = pad_masks.to(torch.bool)
    qcode = torch.where(pad, cum, torch.full_like(cum, -1)).to(torch.int32).contiguous()
    kcode = torch.where(pad, cum, torch.full_like(cum, INT_MAX)).to(torch.int32).contiguous()
    pos = (torch.cumsum(pad.to(torch.int32), dim=1) - 1).to(torch.int32).contiguous()
    return qcode, kcode, pos


_EXPERT_STREAM = os.environ.get("KAI0_EXPERT_STREAM", "1") != "0"  # the action expert's chain on a second HIP stream
_SKIP_DEAD_PREFIX = True  # last layer: no prefix o_proj / MLP (dead values); set_skip_dead_prefix(False) computes them (tests)


def set_expert_stream(on: bool) -> bool:
    """Switch the second stream of `forward_joint` on / off (bench.py times every GEMM launch alone); returns the old setting."""
    global _EXPERT_STREAM
    old, _EXPERT_STREAM = _EXPERT_STREAM, bool(on)
    return old


def set_skip_dead_prefix(on: bool) -> bool:
    """Switch the elimination of the last layer's dead prefix o_proj / MLP on / off (tests); returns the old setting."""
    global _SKIP_DEAD_PREFIX
    old, _SKIP_DEAD_PREFIX = _SKIP_DEAD_PREFIX, bool(on)
    return old


class _NoUnitHooks:
    """Default `unit_hooks`: the model announces each sharding unit (kai0_amd.sharded) to nobody."""

    def pre_forward(self, unit: str) -> bool:
        return False  # (the sharded engine returns True when the current stream had to wait for a parameter gather)

    def post_forward(self, unit: str, *tensors):
        return tensors


class PaliGemmaWithExpertModel(nn.Module):
    """Parameter tree + kernels-backed compute of gemma_pytorch.py:12-281."""

    def __init__(self, vlm_config, action_expert_config, use_adarms=None, precision: str = "bfloat16",
                 vocab: int = 257_152, siglip: SiglipConfig | None = None):  # fmt: skip
        super().__init__()
        if use_adarms is None:
            use_adarms = [False, False]
        if use_adarms[0]:
            raise NotImplementedError("adaRMS on the PaliGemma tower is not part of pi0.5")
        self.vlm_cfg, self.exp_cfg = vlm_config, action_expert_config
        self.siglip_cfg = siglip or SiglipConfig()
        if self.siglip_cfg.projection_dim != vlm_config.width:
            # The image tokens join the prompt tokens in ONE prefix sequence: the projector must end at the PaliGemma width.  The
            # reference's torch model hard-codes 2048 (gemma_pytorch.py:39: right for gemma_2b, a shape error in its torch.cat for
            # the "dummy" variants of debug / debug_pi05); its JAX model sizes the head by the tower (pi0.py:83
            # `num_classes=paligemma_config.width`), which is what is done here instead of failing.
            self.siglip_cfg = dataclasses.replace(self.siglip_cfg, projection_dim=vlm_config.width)
        self.paligemma = PaliGemmaForConditionalGeneration(vlm_config, vocab, self.siglip_cfg)
        self.gemma_expert = GemmaForCausalLM(action_expert_config, vocab, use_adarms[1])
        self.remat = False
        # the sharded trainer hangs its parameter gather / release logic here (Trainer -> PI0Pytorch.set_unit_hooks)
        self.unit_hooks = _NoUnitHooks()
        self.to_bfloat16_for_selected_params(precision)

    def to_bfloat16_for_selected_params(self, precision: str = "bfloat16"):
        """gemma_pytorch.py:63-83. The HIP path computes in bf16; "float32" storage is accepted for loading
        f32 checkpoints (model_arithmetic saves f32) but must be followed by a "bfloat16" call before compute."""
        if precision == "bfloat16":
            self.to(dtype=BF16)
        elif precision == "float32":
            self.to(dtype=F32)
            return
        else:
            raise ValueError(f"Invalid precision: {precision}")
        for name, param in self.named_parameters():
            if any(sel in name for sel in KEEP_F32_SELECTORS):
                param.data = param.data.to(dtype=F32)

    # ---- SigLIP + projector (modeling_siglip.py:763-796; modeling_paligemma.py:232-245) -------------------
    def embed_image(self, image: torch.Tensor) -> torch.Tensor:
        """image f32 [N, 3, HW, HW] in [-1, 1] -> bf16 [N, n_patches, projection_dim]."""
        n = image.shape[0]
        vt = self.paligemma.model.vision_tower.vision_model
        sc = self.siglip_cfg
        S = vt.embeddings.num_patches
        NH, HD = sc.num_heads, sc.hidden_size // sc.num_heads
        emb = vt.embeddings
        hk = self.unit_hooks
        hk.pre_forward("siglip.embed")
        x = ops.patch_embed(image.contiguous(), emb.patch_embedding.weight, emb.patch_embedding.bias,
                            emb.position_embedding.weight, sc.patch_size)  # fmt: skip
        (x,) = hk.post_forward("siglip.embed", x)

        def layer_fn(x, layer):
            x, h = ops.layernorm_res(x, layer.layer_norm1.weight, layer.layer_norm1.bias, layer.layer_norm1.eps)
            at = layer.self_attn
            q, k, v = ops.linear_multi(h, [at.q_proj.weight, at.k_proj.weight, at.v_proj.weight],
                                       [at.q_proj.bias, at.k_proj.bias, at.v_proj.bias])
            a = ops.siglip_attention(q, k, v, n, S, NH, HD)
            x = _lin(a, at.out_proj, residual=x)
            x, h = ops.layernorm_res(x, layer.layer_norm2.weight, layer.layer_norm2.bias, layer.layer_norm2.eps)
            m = layer.mlp
            return ops.gelu_mlp(h, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, residual=x)

        for l, layer in enumerate(vt.encoder.layers):
            hk.pre_forward(f"siglip.{l}")
            x = self._maybe_remat(layer_fn, x, layer)
            (x,) = hk.post_forward(f"siglip.{l}", x)
        hk.pre_forward("prefix")  # post-LN, projector, token / action / time embeddings: closed by PI0Pytorch._trunk
        x = ops.layernorm(x, vt.post_layernorm.weight, vt.post_layernorm.bias, vt.post_layernorm.eps)
        x = _lin(x, self.paligemma.model.multi_modal_projector.linear)
        return x.view(n, S, -1)

    def embed_language_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """Raw embedding lookup (gemma_pytorch.py:88-89), [B, T] int64 -> bf16 [B, T, D]."""
        table = self.paligemma.model.language_model.embed_tokens.weight
        return ops.embed(table, tokens, 1.0).view(*tokens.shape, -1)

    def _maybe_remat(self, fn, *args):
        if self.remat and self.training and torch.is_grad_enabled():
            return torch.utils.checkpoint.checkpoint(fn, *args, use_reentrant=False, preserve_rng_state=False)
        return fn(*args)

    # ---- joint, layer-interleaved forward (gemma_pytorch.py:126-279) ---------------------------------------
    def forward_joint(self, prefix: torch.Tensor, suffix: torch.Tensor, qcode, kcode, pos, cond: torch.Tensor,
                      B: int, P: int, Hs: int) -> torch.Tensor:  # fmt: skip
        """prefix bf16 [B*P, Dp], suffix bf16 [B*Hs, De], cond f32 [B, De] -> suffix output after the expert's
        final adaRMS norm, bf16 [B*Hs, De].  (The prefix's final norm has no consumer in training.)"""
        lm, ex = self.paligemma.model.language_model, self.gemma_expert.model
        cfg = self.vlm_cfg
        H, HD = cfg.num_heads, cfg.head_dim
        inv_freq = lm.rope_inv_freq()

        # The two towers meet only in the joint attention.  The action expert's side of a layer is ~15 launches over B*50 rows
        # (10-25 us each: pure launch latency, a few dozen tiles), the PaliGemma side a handful of GEMMs over B*968 rows
        # (0.2-2.3 ms each): the expert's chain runs on a second HIP stream next to them, joined before and forked after the
        # attention.  autograd replays every backward node on the stream its forward ran on (with the event hand-offs and
        # allocator bookkeeping between them), so the backward overlaps the same way.  Not with rematerialisation (the
        # recomputation would run inside another node's backward).
        dual = (_EXPERT_STREAM and prefix.is_cuda and not (self.remat and self.training)
                and getattr(self.unit_hooks, "mode", "zero2") != "fsdp")  # fsdp frees a unit's parameters behind the main stream only
        main = torch.cuda.current_stream() if dual else None
        side = ops.side_stream(prefix.device) if dual else None

        def on_side():
            return torch.cuda.stream(side) if dual else contextlib.nullcontext()

        def hand(ts, to):  # tensors allocated under one stream and read under the other
            if dual:
                for t in ts:
                    t.record_stream(to)

        n_layers = len(lm.layers)

        def layer_fn(xp, xs, lp, le, last=False):
            ap, ae = lp.self_attn, le.self_attn
            with on_side():
                mod1 = ops.linear_f32(cond, le.input_layernorm.dense.weight, le.input_layernorm.dense.bias)
                xs, hs, gate1 = ops.adarms_res(xs, mod1, Hs, le.input_layernorm.eps)
                qkv_s = ops.linear_multi(hs, [ae.q_proj.weight, ae.k_proj.weight, ae.v_proj.weight])
            xp, hp = ops.rmsnorm_res(xp, lp.input_layernorm.weight, lp.input_layernorm.eps)
            qkv_p = ops.linear_multi(hp, [ap.q_proj.weight, ap.k_proj.weight, ap.v_proj.weight])
            if dual:
                main.wait_stream(side)
                hand(qkv_s, main)
            att_p, att_s = ops.joint_attention(pos, qcode, kcode, inv_freq, H, HD, (P, Hs), (*qkv_p, *qkv_s))
            if dual:
                side.wait_stream(main)
                hand((att_s,), side)
            # suffix (action expert): gated residuals (modeling_gemma.py:209-227)
            with on_side():
                xs = ops.gated_residual(xs, _lin(att_s, ae.o_proj), gate1, Hs)
                mod2 = ops.linear_f32(cond, le.post_attention_layernorm.dense.weight, le.post_attention_layernorm.dense.bias)
                xs, hs, gate2 = ops.adarms_res(xs, mod2, Hs, le.post_attention_layernorm.eps)
                ys = ops.geglu_mlp(hs, le.mlp.gate_proj.weight, le.mlp.up_proj.weight, le.mlp.down_proj.weight)
                xs = ops.gated_residual(xs, ys, gate2, Hs)
            if last and _SKIP_DEAD_PREFIX:
                # nothing reads the prefix stream after the last joint attention (the model's output is the suffix; the prefix's
                # final norm has no consumer either): its o_proj, post-attention norm and MLP in the LAST layer are dead values —
                # 0.2 TFLOP per sample that XLA removes from the reference's jitted JAX step and the inference engine never
                # computed.  Their parameters receive no gradient either way (tests/test_model_gpu.py).
                return xp, xs
            # prefix: o_proj + residual fused in the GEMM epilogue, then RMSNorm -> GeGLU MLP -> residual
            xp = _lin(att_p, ap.o_proj, residual=xp)
            xp, hp = ops.rmsnorm_res(xp, lp.post_attention_layernorm.weight, lp.post_attention_layernorm.eps)
            xp = ops.geglu_mlp(hp, lp.mlp.gate_proj.weight, lp.mlp.up_proj.weight, lp.mlp.down_proj.weight, residual=xp)
            return xp, xs

        xp, xs = prefix, suffix
        hk = self.unit_hooks
        if dual:
            side.wait_stream(main)
            hand((xs, cond), side)
        for l, (lp, le) in enumerate(zip(lm.layers, ex.layers, strict=True)):
            gathered = hk.pre_forward(f"joint.{l}")
            if dual and gathered:
                # the unit's parameters were just completed behind a wait on the MAIN stream: the second stream must see them
                # too.  Without a gather nothing orders layer l+1's expert chain behind the main stream's o_proj / MLP of layer l
                # (its longest kernels) except the data it really needs.
                side.wait_stream(main)
            xp, xs = self._maybe_remat(layer_fn, xp, xs, lp, le, l == n_layers - 1)
            xp, xs = hk.post_forward(f"joint.{l}", xp, xs)
        hk.pre_forward("head")  # final adaRMS norm, action_out_proj (and the estimator's value head): never released early
        if dual:
            main.wait_stream(side)
            hand((xs,), main)
        modf = ops.linear_f32(cond, ex.norm.dense.weight, ex.norm.dense.bias)
        out, _ = ops.adarms(xs, modf, Hs, ex.norm.eps)
        return out


class PI0Pytorch(nn.Module):
    """Drop-in for openpi's `PI0Pytorch` (pi0_pytorch.py:84-461), pi0.5 branch."""

    def __init__(self, config: Pi0Config):
        super().__init__()
        self.config = config
        self.pi05 = config.pi05
        if not self.pi05:
            raise NotImplementedError("only the pi0.5 branch (pi05=True) is on the hot path")
        vlm = get_config(config.paligemma_variant)
        exp = get_config(config.action_expert_variant)
        siglip = getattr(config, "siglip", None) or SiglipConfig()
        self.paligemma_with_expert = PaliGemmaWithExpertModel(
            vlm, exp, use_adarms=[False, True], precision=config.dtype, vocab=getattr(config, "vocab_size", 257_152),
            siglip=siglip,
        )  # fmt: skip
        self.action_in_proj = Linear(config.action_dim, exp.width)
        self.action_out_proj = Linear(exp.width, config.action_dim)
        self.time_mlp_in = Linear(exp.width, exp.width)
        self.time_mlp_out = Linear(exp.width, exp.width)
        self.gradient_checkpointing_enabled = False
        # the reference hard-codes train=True (random crop/rotate/colour) in forward (pi0_pytorch.py:318);
        # parity tests switch it off and inject noise/time.
        self.train_augmentation = True
        self.trim_prompt_padding = False  # training forward: drop the prompt slots no sample of the batch uses (_trim_prompt)
        # sample_actions: the same for the request's prompt (the tokenizer pads every prompt to max_token_len; a task prompt fills
        # 10-40 of pi0.5's 200 slots).  Off here — the model computes what the reference computes; policy.create_trained_policy turns
        # it on for the serve path.  Engines are kept per (batch, prompt slots, cameras), a few at a time (_ENGINE_SLOTS).
        self.trim_prompt_padding_infer = False
        self.trim_prompt_granule_infer = 64  # serve path: prompt lengths are bucketed (see _trim_prompt)
        self._engine = None

    # ---- reference API ------------------------------------------------------------------------------------
    # The inference engine holds stacked weight copies and a captured hipGraph: anything that can change or move the
    # parameters drops it (it is rebuilt on the next sample_actions call).
    _ENGINE_SLOTS = 4

    @property
    def _engine(self):
        """the engine of the last sample_actions call (None: nothing built, or dropped)"""
        return self.__dict__.get("_engine_cur")

    @_engine.setter
    def _engine(self, eng):
        if eng is None:  # dropping the engine drops every cached one: they all hold copies of the same weights
            self.__dict__["_engine_lru"] = {}
        self.__dict__["_engine_cur"] = eng

    def invalidate_inference_engine(self):
        self._engine = None

    def inference_is_stale(self) -> bool:
        """True if the last action chunk (the caller has synchronised with it) was computed while a source weight of the engine's
        derived copies had been edited in place behind autograd's back (infer.InferenceEngine._content_init)."""
        eng = self._engine
        return eng is not None and eng.stale()

    def train(self, mode: bool = True):
        if mode:
            self._engine = None
        return super().train(mode)

    def _apply(self, fn, recurse=True):
        self._engine = None
        return super()._apply(fn, recurse)

    def load_state_dict(self, *args, **kwargs):
        self._engine = None
        return super().load_state_dict(*args, **kwargs)

    # ---- sharded training (kai0_amd.sharded / kai0_amd.train) ---------------------------------------------------
    def sharding_units(self):
        """[(unit name, [parameters])] in forward-use order: what is gathered, used and released together.  The names are
        the ones the forward announces through `unit_hooks.pre_forward / post_forward`."""
        pe = self.paligemma_with_expert
        vt = pe.paligemma.model.vision_tower.vision_model
        lm, ex = pe.paligemma.model.language_model, pe.gemma_expert.model
        units = [("siglip.embed", list(vt.embeddings.parameters()))]
        def siglip_layer(layer):
            # q | k | v weights (and biases) back to back in that order: the trainer's flat buffers then hold the STACKED projection
            # weight / bias / gradient of ops.linear_multi as views (ops._as_rows) — HF's registration order is k, v, q with the
            # biases in between, which cost two concatenations per layer and forward and three gradient copies per backward
            at = layer.self_attn
            first = [at.q_proj.weight, at.k_proj.weight, at.v_proj.weight, at.q_proj.bias, at.k_proj.bias, at.v_proj.bias]
            ids = {id(p) for p in first}
            return first + [p for p in layer.parameters() if id(p) not in ids]

        units += [(f"siglip.{l}", siglip_layer(layer)) for l, layer in enumerate(vt.encoder.layers)]
        units.append(("prefix", [*vt.post_layernorm.parameters(), *pe.paligemma.model.multi_modal_projector.parameters(),
                                 lm.embed_tokens.weight, *self.action_in_proj.parameters(), *self.time_mlp_in.parameters(),
                                 *self.time_mlp_out.parameters()]))  # fmt: skip
        units += [(f"joint.{l}", [*lp.parameters(), *le.parameters()]) for l, (lp, le) in enumerate(zip(lm.layers, ex.layers))]
        taken = {id(p) for _, ps in units for p in ps}
        dead = pe.gemma_expert.lm_head.weight
        units.append(("head", [p for p in self.parameters() if id(p) not in taken and p is not dead]))
        return units

    def set_unit_hooks(self, hooks):
        self.paligemma_with_expert.unit_hooks = hooks if hooks is not None else _NoUnitHooks()

    def gradient_checkpointing_enable(self):
        """pi0_pytorch.py:126-133. Activations fit in 288 GB HBM, so this is optional here; when enabled each
        layer is rematerialised in backward exactly like the reference's per-layer checkpoint."""
        self.gradient_checkpointing_enabled = True
        self.paligemma_with_expert.remat = True
        pe = self.paligemma_with_expert
        pe.paligemma.model.language_model.gradient_checkpointing = True
        pe.paligemma.model.vision_tower.gradient_checkpointing = True
        pe.gemma_expert.model.gradient_checkpointing = True
        logger.info("Enabled gradient checkpointing for PI0Pytorch model")

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing_enabled = False
        self.paligemma_with_expert.remat = False
        pe = self.paligemma_with_expert
        pe.paligemma.model.language_model.gradient_checkpointing = False
        pe.paligemma.model.vision_tower.gradient_checkpointing = False
        pe.gemma_expert.model.gradient_checkpointing = False

    def is_gradient_checkpointing_enabled(self):
        return self.gradient_checkpointing_enabled

    def sample_noise(self, shape, device):
        return torch.normal(mean=0.0, std=1.0, size=shape, dtype=F32, device=device)

    def sample_time(self, bsize, device):
        a = torch.as_tensor(1.5, dtype=F32, device=device)
        b = torch.as_tensor(1.0, dtype=F32, device=device)
        t = torch.distributions.Beta(a, b).sample((bsize,))
        return (t * 0.999 + 0.001).to(dtype=F32, device=device)

    def _preprocess_observation(self, observation, *, train=True):
        res = self.paligemma_with_expert.siglip_cfg.image_size
        obs = preprocess_observation(observation, train=train, image_resolution=(res, res))
        return (list(obs.images.values()), list(obs.image_masks.values()), obs.tokenized_prompt,
                obs.tokenized_prompt_mask, obs.state)  # fmt: skip

    @staticmethod
    def _trim_prompt(lang_tokens, lang_masks, granule: int = 8):
        """Opt-in (`model.trim_prompt_padding = True` for the training forward, `trim_prompt_padding_infer` for sample_actions): cut
        the prompt to the longest valid prompt of the batch, rounded up to 8 tokens.  The tokenizer pads at the end (tokenizer.py:22-47), padded tokens are invisible as keys
        (`make_att_2d_masks`) and their rows are read by nobody, so the loss and the gradients do not change beyond summation
        order — but every prefix-row GEMM shrinks with the rows (200 slots for 64-128 valid tokens in the bench's synthetic
        prompts: 968 -> 896 prefix rows).  Costs one scalar device-to-host read per step (the host waits for the stream once).
        `granule`: the kept length is rounded up to a multiple of it.  Training uses 8; the serve path 64, because an inference
        engine (stacked weight copies + a captured hipGraph) is built per prompt length and pi0.5's discretised state puts a
        request-dependent number of tokens into the prompt (tokenizer.py:24-29): with 64-token buckets a 200-slot prompt has at
        most four lengths {64, 128, 192, 200} = `_ENGINE_SLOTS`, so no request rebuilds an engine after warm-up."""
        m = lang_masks.to(torch.bool)
        T = m.shape[1]
        idx = torch.arange(1, T + 1, device=m.device, dtype=torch.int32)
        last = int((m * idx).max().item()) if m.numel() else 0  # exclusive end of the last valid token over the batch
        keep = min(T, max(granule, (last + granule - 1) // granule * granule))
        if keep == T:
            return lang_tokens, lang_masks
        return lang_tokens[:, :keep].contiguous(), lang_masks[:, :keep].contiguous()

    # ---- embeddings ---------------------------------------------------------------------------------------
    def embed_prefix(self, images, img_masks, lang_tokens, lang_masks):
        """pi0_pytorch.py:186-235 -> (embs bf16 [B, P, D], pad_masks bool [B, P], att_masks bool [B, P])."""
        pe = self.paligemma_with_expert
        B = lang_tokens.shape[0]
        ncam = len(images)
        feats = pe.embed_image(torch.cat(images, dim=0))  # one SigLIP pass over ncam*B images
        n_img = feats.shape[1]
        D = feats.shape[2]
        lang = ops.embed(pe.paligemma.model.language_model.embed_tokens.weight, lang_tokens, ops.sqrt_scale(D))
        T = lang_tokens.shape[1]
        P = ncam * n_img + T
        embs = PrefixAssembleFn.apply(feats.reshape(ncam * B * n_img, D), lang, B, ncam, n_img, T)
        pad = torch.cat([m[:, None].expand(B, n_img) for m in img_masks] + [lang_masks.to(torch.bool)], dim=1)
        att = torch.zeros((B, P), dtype=torch.bool, device=pad.device)
        return embs.view(B, P, D), pad, att

    def embed_suffix(self, state, noisy_actions, timestep):
        """pi0_pytorch.py:237-314 (pi0.5): -> (action embs f32 [B, H, De], pad, att, adarms_cond f32 [B, De])."""
        B, Hs, A = noisy_actions.shape
        De = self.action_in_proj.out_features
        te = torch.empty((B, De), dtype=F32, device=noisy_actions.device)
        _lib.call("kai0_time_sincos", timestep.contiguous().data_ptr(), te.data_ptr(), B, De, 4e-3, 4.0, ops._stream())
        x = ops.silu_f32(ops.linear_f32(te, self.time_mlp_in.weight, self.time_mlp_in.bias))
        cond = ops.silu_f32(ops.linear_f32(x, self.time_mlp_out.weight, self.time_mlp_out.bias))
        a = ops.linear_f32(noisy_actions.reshape(B * Hs, A).contiguous(), self.action_in_proj.weight, self.action_in_proj.bias)
        pad = torch.ones((B, Hs), dtype=torch.bool, device=a.device)
        att = torch.zeros((B, Hs), dtype=torch.bool, device=a.device)
        att[:, 0] = True
        return a.view(B, Hs, De), pad, att, cond

    # ---- training forward ---------------------------------------------------------------------------------
    def _trunk(self, images, img_masks, lang_tokens, lang_masks, state, actions, noise, time, x_t=None):
        """Everything of the training forward up to the expert's output: -> (u_t f32 [B*H, A], suffix_out f32 [B*H, De],
        v_t f32 [B*H, A]) with suffix_out already through the final adaRMS norm (pi0_pytorch.py:326-368).  With `x_t`
        given the flow-matching mix is skipped (u_t = None): the suffix is embedded from x_t as is."""
        u_t = None
        if x_t is None:
            actions = actions.to(F32).contiguous()
            x_t, u_t = ops.flow_mix(noise.to(F32).contiguous(), actions, time.to(F32).contiguous())
        else:
            x_t = x_t.to(F32).contiguous()
        if self.trim_prompt_padding:
            lang_tokens, lang_masks = self._trim_prompt(lang_tokens, lang_masks)
        prefix, ppad, patt = self.embed_prefix(images, img_masks, lang_tokens, lang_masks)
        suffix, spad, satt, cond = self.embed_suffix(state, x_t, time)
        prefix, suffix, cond = self.paligemma_with_expert.unit_hooks.post_forward("prefix", prefix, suffix, cond)
        B, P, Dp = prefix.shape
        Hs, De = suffix.shape[1], suffix.shape[2]
        qcode, kcode, pos = build_mask_codes(torch.cat([ppad, spad], dim=1), torch.cat([patt, satt], dim=1))
        suffix_bf = ops.cast_ag(suffix.reshape(B * Hs, De), BF16)
        out = self.paligemma_with_expert.forward_joint(prefix.reshape(B * P, Dp), suffix_bf, qcode, kcode, pos, cond, B, P, Hs)
        out32 = ops.cast_ag(out, F32)
        v_t = ops.linear_f32(out32, self.action_out_proj.weight, self.action_out_proj.bias)
        return (u_t.view(B * Hs, -1) if u_t is not None else None), out32, v_t

    def forward(self, observation, actions, noise=None, time=None) -> torch.Tensor:
        """pi0_pytorch.py:316-373 -> un-reduced flow-matching loss f32 [B, H, A]."""
        images, img_masks, lang_tokens, lang_masks, state = self._preprocess_observation(
            observation, train=self.train_augmentation
        )
        if noise is None:
            noise = self.sample_noise(actions.shape, actions.device)
        if time is None:
            time = self.sample_time(actions.shape[0], actions.device)
        B, Hs = actions.shape[0], actions.shape[1]
        u_t, _, v_t = self._trunk(images, img_masks, lang_tokens, lang_masks, state, actions, noise, time)
        return ops.mse_loss(u_t, v_t).view(B, Hs, -1)

    # ---- inference ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_actions(self, device, observation, noise=None, num_steps=10) -> torch.Tensor:
        """pi0_pytorch.py:375-419: prefix pass into a static KV cache, then `num_steps` Euler steps."""
        from .infer import InferenceEngine

        bsize = observation.state.shape[0]
        if noise is None:
            noise = self.sample_noise((bsize, self.config.action_horizon, self.config.action_dim), device)
        images, img_masks, lang_tokens, lang_masks, state = self._preprocess_observation(observation, train=False)
        wait = getattr(self.paligemma_with_expert.unit_hooks, "wait_params", None)
        if wait is not None:  # a sharded trainer owns the parameters: its in-flight all-gathers must have landed
            wait()
        if self.trim_prompt_padding_infer:
            lang_tokens, lang_masks = self._trim_prompt(lang_tokens, lang_masks, granule=self.trim_prompt_granule_infer)
        key = (bsize, lang_tokens.shape[1], len(images))
        eng = self._engine
        if eng is not None and eng.stale():  # weights edited in place behind autograd's back: every cached engine holds old copies
            self._engine = eng = None
            # ... the other prompt-length buckets' engines too: their own stamps still equal their reference (they have not run since
            # the edit), so `compatible()` would accept them and the next request in their bucket would compute from old weights
            self.__dict__.pop("_engine_lru", None)
        if eng is not None and eng.shape_matches(*key):
            # the common serving case: same request shape as the last call.  The captured chunk is queued BEFORE the weights are checked
            # (0.13 ms of host work that would otherwise stand in front of every chunk); a failed check falls through to the rebuild
            out = eng.replay_then_verify(images, img_masks, lang_tokens, lang_masks, noise.to(F32), num_steps)
            if out is not None:
                return out
        if eng is None or not eng.compatible(*key):
            lru = self.__dict__.setdefault("_engine_lru", {})
            eng = lru.pop(key, None)
            if eng is None or not eng.compatible(*key):
                eng = InferenceEngine(self, *key)
            lru[key] = eng  # (re-)inserted last: most recently used
            while len(lru) > self._ENGINE_SLOTS:
                lru.pop(next(iter(lru)))
            self.__dict__["_engine_cur"] = eng
        return eng.sample_actions(images, img_masks, lang_tokens, lang_masks, noise.to(F32), num_steps)


class AdvantageEstimator(PI0Pytorch):
    """The advantage / progress estimator of Stage-Advantage (pi0_pytorch.py:464-644): the pi0.5 trunk plus a value head
    (Linear-SiLU-Linear-SiLU-Linear-Tanh, f32) on the first suffix token's final representation.
      forward      -> loss [B, H] = loss_action_weight * mean_d mse(u_t, v_t) + loss_value_weight * mse(value, clamp(progress, +-1))
      sample_values-> value [B, 1] from one joint forward with sampled (or given) noise / time
    State-dict keys follow nn.Sequential numbering: value_head.{0,2,4}.{weight,bias}.  Observations may carry up to six
    images (two timesteps x three cameras, ordered (timestep, base < left_wrist < right_wrist)); no augmentation
    (preprocess_observation_pytorch_custom(..., apply_aug=False))."""

    def __init__(self, config):
        super().__init__(config)
        self.loss_value_weight = getattr(config, "loss_value_weight", 0.0)
        self.loss_action_weight = getattr(config, "loss_action_weight", 1.0)
        w = self.action_in_proj.out_features
        self.value_head = nn.Sequential(Linear(w, w), nn.Identity(), Linear(w, w), nn.Identity(), Linear(w, 1), nn.Identity())

    def _preprocess_observation(self, observation, *, train=True, return_full_obs=False):
        from .preprocessing import preprocess_observation_custom

        res = self.paligemma_with_expert.siglip_cfg.image_size
        obs = preprocess_observation_custom(observation, train=train, image_resolution=(res, res), apply_aug=False)
        full = (list(obs.images.values()), list(obs.image_masks.values()), obs.tokenized_prompt, obs.tokenized_prompt_mask,
                obs.state, obs)  # fmt: skip
        return full if return_full_obs else full[:-1]

    def _value(self, out32, B: int, Hs: int):
        vh = self.value_head
        deep = out32.view(B, Hs, -1)[:, 0, :].contiguous()  # first suffix token, f32
        x = ops.silu_f32(ops.linear_f32(deep, vh[0].weight, vh[0].bias))
        x = ops.silu_f32(ops.linear_f32(x, vh[2].weight, vh[2].bias))
        return torch.tanh(ops.linear_f32(x, vh[4].weight, vh[4].bias))  # [B, 1]

    def forward(self, observation, actions, noise=None, time=None, return_loss_dict=False):
        images, img_masks, lang_tokens, lang_masks, state, obs_full = self._preprocess_observation(
            observation, train=self.training, return_full_obs=True
        )
        if noise is None:
            noise = self.sample_noise(actions.shape, actions.device)
        if time is None:
            time = self.sample_time(actions.shape[0], actions.device)
        B, Hs = actions.shape[0], actions.shape[1]
        u_t, out32, v_t = self._trunk(images, img_masks, lang_tokens, lang_masks, state, actions, noise, time)
        loss_action = ops.mse_loss(u_t, v_t).view(B, Hs, -1).mean(dim=-1)  # [B, H]
        loss = loss_action * self.loss_action_weight
        value = self._value(out32, B, Hs)
        target = torch.clamp(obs_full.progress.float(), -1.0, 1.0).unsqueeze(1)
        value_loss = (value - target) ** 2 * self.loss_value_weight  # [B, 1]
        aux = {"loss_action": loss_action.detach().mean(), "loss_value": value_loss.detach().mean()}
        loss = loss + value_loss
        return (loss, aux) if return_loss_dict else loss

    @torch.no_grad()
    def sample_values(self, device, observation, noise=None, time=None) -> torch.Tensor:
        """pi0_pytorch.py:596-644.  `noise` / `time` may be injected (the reference samples them)."""
        images, img_masks, lang_tokens, lang_masks, state = self._preprocess_observation(observation, train=False)
        B = state.shape[0]
        Hs = self.config.action_horizon
        if noise is None:
            noise = self.sample_noise((B, Hs, self.config.action_dim), device)
        if time is None:
            time = self.sample_time(B, device)
        _, out32, _ = self._trunk(images, img_masks, lang_tokens, lang_masks, state, None, None, time, x_t=noise)
        return self._value(out32, B, Hs)


class PrefixAssembleFn(torch.autograd.Function):
    """Concatenate [cam0 | cam1 | ... | language] along the sequence axis into the flat prefix [B*P, D]
    (pi0_pytorch.py:228).  feats are laid out image-major: image index = cam*B + b."""

    @staticmethod
    def forward(ctx, feats, lang, B: int, ncam: int, n_img: int, T: int):
        D = feats.shape[1]
        if lang.shape[-1] != D or feats.shape[0] != ncam * B * n_img or lang.numel() != B * T * D:
            raise ValueError(f"prefix assembly: image tokens {tuple(feats.shape)} and prompt tokens {tuple(lang.shape)} do not form "
                             f"one [B={B}, {ncam} x {n_img} + {T}, D] sequence (projector width must equal the PaliGemma width)")
        P = ncam * n_img + T
        out = torch.empty((B * P, D), dtype=BF16, device=feats.device)
        for c in range(ncam):
            src = feats[c * B * n_img :]
            ops._copy_rows(src, out, B, n_img, D, n_img * D, 0, D, P * D, c * n_img, D)
        ops._copy_rows(lang, out, B, T, D, T * D, 0, D, P * D, ncam * n_img, D)
        ctx.dims = (B, ncam, n_img, T, D, P)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, ncam, n_img, T, D, P = ctx.dims
        dout = dout.contiguous()
        dfeats = torch.empty((ncam * B * n_img, D), dtype=BF16, device=dout.device)
        dlang = torch.empty((B * T, D), dtype=BF16, device=dout.device)
        for c in range(ncam):
            dst = dfeats[c * B * n_img :]
            ops._copy_rows(dout, dst, B, n_img, D, P * D, c * n_img, D, n_img * D, 0, D)
        ops._copy_rows(dout, dlang, B, T, D, P * D, ncam * n_img, D, T * D, 0, D)
        return dfeats, dlang, None, None, None, None
