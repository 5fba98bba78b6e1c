"""Golden vectors for observation preprocessing, produced by executing the reference's own functions
(models_pytorch/preprocessing_pytorch.py:20-173, shared/image_tools.py:55-126; lifted with `ast`, see
make_reference_blocks_golden.py).  train=True is stochastic: the vectors are tied to torch.manual_seed(1234) on CPU and to
the ORDER in which the reference draws (crop offsets, angle, brightness, contrast, saturation, per camera).

    python tests/golden/make_reference_preproc_golden.py      # build container only; needs /root/reference
"""
import logging
import os
import sys
import types

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_blocks_golden as B  # noqa: E402

SRC = "/root/reference/src/openpi"
ins = B.base_ns()
B.lift(f"{SRC}/shared/image_tools.py", ["resize_with_pad_torch"], ins)
pns = B.base_ns()
pns.update({"image_tools": types.SimpleNamespace(resize_with_pad_torch=ins["resize_with_pad_torch"]), "logger": logging.getLogger("ref"),
            "Sequence": __import__("collections.abc").abc.Sequence,
            "IMAGE_KEYS": ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb"), "IMAGE_RESOLUTION": (224, 224)})  # fmt: skip
B.lift(f"{SRC}/models_pytorch/preprocessing_pytorch.py", ["preprocess_observation_pytorch"], pns)
resize, preprocess = ins["resize_with_pad_torch"], pns["preprocess_observation_pytorch"]

g = torch.Generator().manual_seed(77)
out = {}
# ---- resize_with_pad_torch: f32 NHWC landscape -> square, uint8 NHWC portrait, channels-first f32 up-scaling --------------
a = torch.rand(2, 48, 64, 3, generator=g) * 2 - 1
b = torch.randint(0, 256, (2, 30, 20, 3), generator=g, dtype=torch.uint8)
c = torch.rand(1, 3, 20, 36, generator=g) * 2 - 1
out.update({"rz.a": a, "rz.a_out": resize(a, 32, 32), "rz.b": b, "rz.b_out": resize(b, 28, 28), "rz.c": c, "rz.c_out": resize(c, 40, 40)})


def make_obs(h, w):
    ims = {k: torch.rand(2, 3, h, w, generator=g) * 2 - 1 for k in pns["IMAGE_KEYS"]}
    return types.SimpleNamespace(images=ims, image_masks={"base_0_rgb": torch.tensor([True, False])}, state=torch.zeros(2, 32),
                                 tokenized_prompt=torch.zeros(2, 4, dtype=torch.long), tokenized_prompt_mask=torch.ones(2, 4, dtype=torch.bool),
                                 token_ar_mask=None, token_loss_mask=None)  # fmt: skip


# ---- preprocess, train=False: non-native resolution (resize + pad), default masks ----------------------------------------
o1 = make_obs(40, 56)
r1 = preprocess(o1, train=False, image_resolution=(48, 48))
for k in pns["IMAGE_KEYS"]:
    out[f"eval.in.{k}"] = o1.images[k]
    out[f"eval.out.{k}"] = r1.images[k].contiguous()
    out[f"eval.mask.{k}"] = r1.image_masks[k].to(torch.uint8)
# ---- preprocess, train=True at native resolution: crop/resize/rotate (base camera), colour jitter (all) -------------------
o2 = make_obs(48, 48)
torch.manual_seed(1234)
r2 = preprocess(o2, train=True, image_resolution=(48, 48))
for k in pns["IMAGE_KEYS"]:
    out[f"train.in.{k}"] = o2.images[k]
    out[f"train.out.{k}"] = r2.images[k].contiguous()
save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "reference_preproc.safetensors"), metadata={"train_seed": "1234"})
print("wrote reference_preproc.safetensors", len(out), "tensors")
