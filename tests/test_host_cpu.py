"""CPU-side checks: the oracle against known answers and its golden fixture, the host logic of the HIP path
(mask codes, Euler schedule, LR schedule, state-dict contract) and the C-ABI surface (symbols only — no compute
without a GPU)."""

import copy
import os
import re

import pytest
import torch
from safetensors.torch import load_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ------------------------------------------------------------------------------------------------- oracle
def test_make_att_2d_masks_known_answers():
    """The three worked examples in the reference docstring (pi0_pytorch.py:52-81)."""
    from oracle.pi0_oracle import make_att_2d_masks

    pad = torch.ones(1, 6, dtype=torch.bool)
    causal = make_att_2d_masks(pad, torch.tensor([[1, 1, 1, 1, 1, 1]]))[0]
    assert torch.equal(causal, torch.tril(torch.ones(6, 6, dtype=torch.bool)))
    prefix_lm = make_att_2d_masks(pad, torch.tensor([[0, 0, 0, 1, 1, 1]]))[0]
    want = torch.tril(torch.ones(6, 6, dtype=torch.bool))
    want[:3, :3] = True
    assert torch.equal(prefix_lm, want)
    blocks = make_att_2d_masks(torch.ones(1, 10, dtype=torch.bool), torch.tensor([[1, 0, 1, 0, 1, 0, 0, 1, 0, 0]]))[0]
    blk = torch.tensor([0, 0, 1, 1, 2, 2, 2, 3, 3, 3])
    assert torch.equal(blocks, blk[None, :] <= blk[:, None])
    pad2 = torch.tensor([[True, True, False, True]])
    m = make_att_2d_masks(pad2, torch.tensor([[0, 0, 0, 0]]))[0]
    assert not m[2].any() and not m[:, 2].any() and m[0, 3] and m[3, 0]


def test_sinusoidal_embedding_known_answers():
    from oracle.pi0_oracle import create_sinusoidal_pos_embedding

    e = create_sinusoidal_pos_embedding(torch.tensor([0.0, 1.0], dtype=torch.float32), 8, 4e-3, 4.0)
    assert e.dtype == torch.float64 and e.shape == (2, 8)
    assert torch.allclose(e[0], torch.tensor([0, 0, 0, 0, 1, 1, 1, 1], dtype=torch.float64))
    # period of the last frequency is max_period = 4 -> sin(2*pi*1/4) = 1
    assert abs(float(e[1, 3]) - 1.0) < 1e-12
    with pytest.raises(ValueError):
        create_sinusoidal_pos_embedding(torch.tensor([0.5]), 7, 4e-3, 4.0)


def test_oracle_reproduces_golden_fixture():
    from tiny import build_pair

    from oracle.pi0_oracle import synthetic_batch

    gold = load_file(os.path.join(HERE, "golden", "tiny_pi05.safetensors"))
    _, oracle, _, ocfg = build_pair("cpu", seed=0, std=0.08)
    obs, actions, noise, time = synthetic_batch(ocfg, 2, seed=0)
    assert torch.equal(noise, gold["noise"]) and torch.equal(time, gold["time"])
    assert torch.equal(obs.tokenized_prompt, gold["tokens"])
    with torch.no_grad():
        loss = oracle(obs, actions, noise, time)
    chunk = oracle.sample_actions(obs, noise, num_steps=10)
    # bf16 CPU GEMM blocking may differ between hosts: compare to tolerance, not bitwise
    assert float((loss - gold["loss"]).norm() / gold["loss"].norm()) < 5e-3
    assert float((chunk - gold["actions"]).norm() / gold["actions"].norm()) < 2e-3


def test_oracle_dtype_policy_and_state_dict_contract():
    """gemma_pytorch.py:63-83 storage dtypes and the key list of SURVEY.md §8a16."""
    from tiny import build_pair

    model, oracle, _, _ = build_pair("cpu")
    sd = oracle.state_dict()
    pw = "paligemma_with_expert."
    vt = pw + "paligemma.model.vision_tower.vision_model."
    lm = pw + "paligemma.model.language_model."
    ex = pw + "gemma_expert.model."
    must = [
        vt + "embeddings.patch_embedding.weight", vt + "embeddings.patch_embedding.bias",
        vt + "embeddings.position_embedding.weight", vt + "encoder.layers.0.layer_norm1.weight",
        vt + "encoder.layers.1.self_attn.out_proj.bias", vt + "encoder.layers.0.mlp.fc1.weight", vt + "post_layernorm.bias",
        pw + "paligemma.model.multi_modal_projector.linear.weight", lm + "embed_tokens.weight",
        lm + "layers.0.self_attn.q_proj.weight", lm + "layers.3.mlp.down_proj.weight", lm + "layers.0.input_layernorm.weight",
        lm + "layers.0.post_attention_layernorm.weight", lm + "norm.weight", pw + "paligemma.lm_head.weight",
        ex + "layers.0.input_layernorm.dense.weight", ex + "layers.0.post_attention_layernorm.dense.bias",
        ex + "norm.dense.weight", ex + "layers.2.mlp.gate_proj.weight", pw + "gemma_expert.lm_head.weight",
        "action_in_proj.weight", "action_out_proj.bias", "time_mlp_in.weight", "time_mlp_out.bias",
    ]  # fmt: skip
    for k in must:
        assert k in sd, k
    assert not any("embed_tokens" in k for k in sd if k.startswith(ex))  # gemma_pytorch.py:59
    assert not any(k.endswith("layernorm.weight") for k in sd if k.startswith(ex))  # adaRMS layers have only `dense`
    assert sd[pw + "paligemma.lm_head.weight"].data_ptr() == sd[lm + "embed_tokens.weight"].data_ptr()  # tied
    f32 = torch.float32
    assert sd[vt + "embeddings.patch_embedding.weight"].dtype == f32
    assert sd[vt + "embeddings.position_embedding.weight"].dtype == f32
    assert sd[lm + "layers.0.input_layernorm.weight"].dtype == f32 and sd[lm + "norm.weight"].dtype == f32
    assert sd[ex + "layers.0.input_layernorm.dense.weight"].dtype == f32 and sd[ex + "norm.dense.bias"].dtype == f32
    assert sd[vt + "encoder.layers.0.layer_norm1.weight"].dtype == torch.bfloat16
    assert sd[vt + "post_layernorm.weight"].dtype == torch.bfloat16
    assert sd[lm + "layers.0.self_attn.q_proj.weight"].dtype == torch.bfloat16
    assert sd["action_in_proj.weight"].dtype == f32 and sd["time_mlp_in.weight"].dtype == f32
    # the HIP model exposes exactly the same keys / shapes / dtypes
    msd = model.state_dict()
    assert list(msd.keys()) == list(sd.keys())
    for k in sd:
        assert msd[k].shape == sd[k].shape and msd[k].dtype == sd[k].dtype, k
        assert torch.equal(msd[k], sd[k]), k


def test_full_size_state_dict_shapes():
    """Full pi0.5 parameter shapes on the meta device (no memory): 3.617 B stored, 3.353 B used."""
    from kai0_amd.config import Pi0Config
    from kai0_amd.model import PI0Pytorch

    with torch.device("meta"):
        m = PI0Pytorch(Pi0Config())
    sd = m.state_dict()
    pw = "paligemma_with_expert."
    assert sd[pw + "paligemma.model.language_model.embed_tokens.weight"].shape == (257152, 2048)
    assert sd[pw + "gemma_expert.lm_head.weight"].shape == (257152, 1024)
    assert sd[pw + "paligemma.model.language_model.layers.17.mlp.gate_proj.weight"].shape == (16384, 2048)
    assert sd[pw + "gemma_expert.model.layers.0.input_layernorm.dense.weight"].shape == (3072, 1024)
    assert sd[pw + "paligemma.model.vision_tower.vision_model.encoder.layers.26.mlp.fc1.weight"].shape == (4304, 1152)
    assert sd[pw + "paligemma.model.vision_tower.vision_model.embeddings.patch_embedding.weight"].shape == (1152, 3, 14, 14)
    seen, total = set(), 0
    for p in m.parameters():
        if id(p) not in seen:
            seen.add(id(p))
            total += p.numel()
    assert abs(total - 3.617e9) < 0.01e9, total
    dead = sd[pw + "gemma_expert.lm_head.weight"].numel()
    assert abs((total - dead) - 3.353e9) < 0.01e9


# --------------------------------------------------------------------------------------------- host logic
def test_mask_codes_are_bit_exact_with_make_att_2d_masks():
    from kai0_amd.model import build_mask_codes
    from oracle.pi0_oracle import make_att_2d_masks

    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        B, S = 3, 37
        pad = torch.rand(B, S, generator=g) > 0.3
        att = (torch.rand(B, S, generator=g) > 0.8).to(torch.int32)
        qcode, kcode, pos = build_mask_codes(pad, att)
        allowed = kcode[:, None, :] <= qcode[:, :, None]
        assert torch.equal(allowed, make_att_2d_masks(pad, att))
        assert torch.equal(pos.to(torch.int64), torch.cumsum(pad, dim=1) - 1)
    # the pi0.5 layout: bidirectional prefix, suffix sees everything valid, prefix never sees the suffix
    P, H = 12, 5
    pad = torch.ones(1, P + H, dtype=torch.bool)
    pad[0, 8:10] = False
    att = torch.zeros(1, P + H, dtype=torch.int32)
    att[0, P] = 1
    qcode, kcode, _ = build_mask_codes(pad, att)
    allowed = (kcode[:, None, :] <= qcode[:, :, None])[0]
    assert not allowed[:P, P:].any() and allowed[P:, :8].all() and not allowed[:, 8:10].any() and allowed[P:, P:].all()


def test_euler_times_matches_reference_loop():
    from kai0_amd.infer import euler_times

    for n in (1, 5, 10, 16):
        dt = torch.tensor(-1.0 / n, dtype=torch.float32)
        t = torch.tensor(1.0, dtype=torch.float32)
        ref = []
        while t >= -dt / 2:  # pi0_pytorch.py:406
            ref.append(float(t))
            t = t + dt
        assert euler_times(n) == ref
    assert len(euler_times(10)) == 10


def test_lr_schedule_matches_reference_formula():
    from kai0_amd.optim import lr_schedule

    kw = dict(warmup_steps=1000, peak_lr=2.5e-5, decay_steps=30000, end_lr=2.5e-6)
    assert abs(lr_schedule(0, **kw) - 2.5e-5 / 1001) < 1e-15
    assert abs(lr_schedule(1000, **kw) - 2.5e-5) < 1e-12
    assert abs(lr_schedule(30000, **kw) - 2.5e-6) < 1e-12
    assert abs(lr_schedule(50000, **kw) - 2.5e-6) < 1e-12
    mid = lr_schedule(15500, **kw)
    assert abs(mid - (2.5e-6 + (2.5e-5 - 2.5e-6) * 0.5)) < 1e-9


def test_observation_from_dict_and_resize_known_answers():
    """Mirrors image_tools_test.py:6-37 (zeros in -> zeros out, shapes) and model.py:128-133."""
    from kai0_amd.preprocessing import Observation, preprocess_observation, resize_with_pad_torch

    u8 = torch.zeros(2, 20, 30, 3, dtype=torch.uint8)
    out = resize_with_pad_torch(u8, 10, 10)
    assert out.shape == (2, 10, 10, 3) and out.dtype == torch.uint8 and int(out.max()) == 0
    f = torch.zeros(2, 3, 20, 30) - 1.0
    out = resize_with_pad_torch(f, 16, 16)
    assert out.shape == (2, 3, 16, 16) and torch.all(out == -1.0)
    data = {
        "image": {k: torch.full((2, 8, 8, 3), 255, dtype=torch.uint8) for k in ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb")},
        "image_mask": {k: torch.ones(2, dtype=torch.bool) for k in ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb")},
        "state": torch.zeros(2, 32), "tokenized_prompt": torch.zeros(2, 5, dtype=torch.int64),
        "tokenized_prompt_mask": torch.ones(2, 5, dtype=torch.bool),
    }  # fmt: skip
    obs = Observation.from_dict(data)
    assert obs.images["base_0_rgb"].shape == (2, 3, 8, 8) and torch.all(obs.images["base_0_rgb"] == 1.0)
    with pytest.raises(ValueError):
        Observation.from_dict({**data, "tokenized_prompt_mask": None} | {"x": 0} if False else {k: v for k, v in data.items() if k != "tokenized_prompt_mask"})
    torch.manual_seed(0)
    pp = preprocess_observation(obs, train=True, image_resolution=(8, 8))
    for k, im in pp.images.items():
        assert im.shape == (2, 3, 8, 8) and float(im.min()) >= -1.0 and float(im.max()) <= 1.0


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected before any kernel is reached."""
    from kai0_amd import ops
    from kai0_amd._lib import Kai0HipError

    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(Kai0HipError):
        ops.linear(x, x)
    with pytest.raises(Kai0HipError):
        ops.rmsnorm(x, torch.zeros(8))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kai0_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"


# ------------------------------------------------------------------------------------------------ C-ABI
def test_cabi_exports_every_declared_symbol():
    """libkai0hip.so loads and exports exactly what include/kai0hip.h declares (no compute call here)."""
    import ctypes

    from kai0_amd import _lib

    header = open(os.path.join(ROOT, "include", "kai0hip.h")).read()
    declared = set(re.findall(r"\b(kai0_[a-z0-9_]+)\s*\(", header)) - {"kai0_gemm_desc"}
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in kai0hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.kai0_abi_version() == 1
    assert ctypes.sizeof(_lib.GemmDesc) == lib.kai0_gemm_desc_size()


def test_pack_skinny_weight_layout():
    """ops.pack_skinny_weight == the fragment-major layout kai0hip.h documents for `kai0_skinny_desc.w_packed`: for the 16-row
    tile t and the 32-wide contraction step s one contiguous block of 512 elements at (t * (K / 32) + s) * 512 holding
    W[16 t + i][32 s + 8 g + e] at (i + 16 g) * 8 + e (pure index arithmetic: runs without a GPU)."""
    import torch

    from kai0_amd import ops

    N, K = 48, 96
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K)
    p = ops.pack_skinny_weight(w).reshape(-1)
    assert p.numel() == N * K and sorted(p.tolist()) == sorted(w.reshape(-1).tolist())
    for t in range(N // 16):
        for s in range(K // 32):
            blk = p[(t * (K // 32) + s) * 512 : (t * (K // 32) + s + 1) * 512]
            for i in (0, 7, 15):
                for g in range(4):
                    for e in (0, 3, 7):
                        assert blk[(i + 16 * g) * 8 + e] == w[16 * t + i, 32 * s + 8 * g + e]
    import pytest

    with pytest.raises(ValueError):
        ops.pack_skinny_weight(torch.zeros(40, 96))


def test_trim_prompt_keeps_every_valid_token():
    """PI0Pytorch._trim_prompt (opt-in training switch): the prompt is cut to the longest valid prompt of the batch, rounded up to 8
    slots; nothing valid is dropped, a batch that uses all slots is returned as is."""
    import torch

    from kai0_amd.model import PI0Pytorch

    tok = torch.arange(2 * 200).reshape(2, 200)
    mask = torch.zeros(2, 200, dtype=torch.bool)
    mask[0, :70] = True
    mask[1, :101] = True
    t, m = PI0Pytorch._trim_prompt(tok, mask)
    assert t.shape == (2, 104) and m.shape == (2, 104) and t.is_contiguous()
    assert torch.equal(t, tok[:, :104]) and int(m.sum()) == int(mask.sum())
    mask[1, :] = True
    t, m = PI0Pytorch._trim_prompt(tok, mask)
    assert t is tok and m is mask
    t, m = PI0Pytorch._trim_prompt(tok, torch.zeros(2, 200, dtype=torch.bool))
    assert t.shape == (2, 8)
