"""Observation preprocessing against vectors produced by executing the reference's own functions
(tests/golden/make_reference_preproc_golden.py: image_tools.resize_with_pad_torch and preprocess_observation_pytorch,
lifted with `ast`).  Deterministic paths must be bit-exact; the augmented path (same seed, same draw order) to 1e-5 —
the reference keeps the rotation angle as an f32 tensor, we take cos/sin on the host in f64."""

import os
import types

import torch
from safetensors.torch import load_file

from kai0_amd import preprocessing as P

G = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_preproc.safetensors"))
KEYS = ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb")


def test_resize_with_pad_bit_exact():
    assert torch.equal(P.resize_with_pad_torch(G["rz.a"], 32, 32), G["rz.a_out"])
    assert torch.equal(P.resize_with_pad_torch(G["rz.b"], 28, 28), G["rz.b_out"])
    assert torch.equal(P.resize_with_pad_torch(G["rz.c"], 40, 40), G["rz.c_out"])


def _obs(prefix):
    return types.SimpleNamespace(images={k: G[f"{prefix}.in.{k}"] for k in KEYS}, image_masks={"base_0_rgb": torch.tensor([True, False])},
                                 state=torch.zeros(2, 32), tokenized_prompt=torch.zeros(2, 4, dtype=torch.long),
                                 tokenized_prompt_mask=torch.ones(2, 4, dtype=torch.bool), token_ar_mask=None, token_loss_mask=None)  # fmt: skip


def test_eval_preprocessing_bit_exact():
    r = P.preprocess_observation(_obs("eval"), train=False, image_resolution=(48, 48))
    for k in KEYS:
        assert torch.equal(r.images[k], G[f"eval.out.{k}"]), k
        assert torch.equal(r.image_masks[k], G[f"eval.mask.{k}"].bool()), k


def test_train_augmentation_same_seed_same_draw_order():
    torch.manual_seed(1234)
    r = P.preprocess_observation(_obs("train"), train=True, image_resolution=(48, 48))
    for k in KEYS:
        ref = G[f"train.out.{k}"]
        assert r.images[k].shape == ref.shape
        assert float((r.images[k] - ref).abs().max()) < 1e-5, k
