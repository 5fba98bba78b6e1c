"""Observation container and device-side preprocessing (host/torch glue at the edge of the hot path).

Mirrors `openpi.models.model.Observation` (models/model.py:85-140) and
`preprocess_observation_pytorch` (models_pytorch/preprocessing_pytorch.py:20-173) including the train-time
augmentation the reference hard-enables in `PI0Pytorch.forward` (pi0_pytorch.py:318).  SURVEY.md §2.1 #3 marks
this as boundary code that stays in torch: it is memory-bound image plumbing, not a kernel target.
"""

from __future__ import annotations

import dataclasses
import logging
from collections.abc import Sequence

import numpy as np
import torch
import torch.nn.functional as F  # noqa: N812

logger = logging.getLogger("kai0_amd")

IMAGE_KEYS = ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb")
IMAGE_RESOLUTION = (224, 224)


@dataclasses.dataclass
class Observation:
    """Inputs of the model (models/model.py:85-124). Images are f32 in [-1, 1], [B, 3, H, W] for torch."""

    images: dict
    image_masks: dict
    state: torch.Tensor
    tokenized_prompt: torch.Tensor | None = None
    tokenized_prompt_mask: torch.Tensor | None = None
    token_ar_mask: torch.Tensor | None = None
    token_loss_mask: torch.Tensor | None = None
    episode_index: torch.Tensor | None = None
    frame_index: torch.Tensor | None = None
    progress: torch.Tensor | None = None
    episode_length: torch.Tensor | None = None
    image_original: dict | None = None

    @classmethod
    def from_dict(cls, data: dict) -> "Observation":
        """models/model.py:126-160: uint8 HWC images become f32 in [-1, 1] (NCHW for torch tensors)."""
        if ("tokenized_prompt" in data) != ("tokenized_prompt_mask" in data):
            raise ValueError("tokenized_prompt and tokenized_prompt_mask must be provided together.")
        images = dict(data["image"])
        for key, img in images.items():
            if isinstance(img, np.ndarray) and img.dtype == np.uint8:
                images[key] = img.astype(np.float32) / 255.0 * 2.0 - 1.0
            elif isinstance(img, torch.Tensor) and img.dtype == torch.uint8:
                images[key] = img.to(torch.float32).permute(0, 3, 1, 2) / 255.0 * 2.0 - 1.0
        return cls(
            images=images,
            image_masks=data["image_mask"],
            state=data["state"],
            tokenized_prompt=data.get("tokenized_prompt"),
            tokenized_prompt_mask=data.get("tokenized_prompt_mask"),
            token_ar_mask=data.get("token_ar_mask"),
            token_loss_mask=data.get("token_loss_mask"),
            episode_index=data.get("episode_index"),
            frame_index=data.get("frame_index"),
            progress=data.get("progress"),
            episode_length=data.get("episode_length"),
            image_original=data.get("image_original"),
        )


def resize_with_pad_torch(images: torch.Tensor, height: int, width: int, mode: str = "bilinear") -> torch.Tensor:
    """shared/image_tools.py:55-126: aspect-preserving resize, pad with black (-1 for f32, 0 for uint8)."""
    channels_last = images.shape[-1] <= 4
    if images.dim() == 3:
        images = images.unsqueeze(0)
    if channels_last:
        images = images.permute(0, 3, 1, 2)
    _, _, cur_h, cur_w = images.shape
    ratio = max(cur_w / width, cur_h / height)
    rh, rw = int(cur_h / ratio), int(cur_w / ratio)
    resized = F.interpolate(images, size=(rh, rw), mode=mode, align_corners=False if mode == "bilinear" else None)
    if images.dtype == torch.uint8:
        resized = torch.round(resized).clamp(0, 255).to(torch.uint8)
    elif images.dtype == torch.float32:
        resized = resized.clamp(-1.0, 1.0)
    else:
        raise ValueError(f"Unsupported image dtype: {images.dtype}")
    ph0, rem_h = divmod(height - rh, 2)
    pw0, rem_w = divmod(width - rw, 2)
    value = 0 if images.dtype == torch.uint8 else -1.0
    padded = F.pad(resized, (pw0, pw0 + rem_w, ph0, ph0 + rem_h), mode="constant", value=value)
    if channels_last:
        padded = padded.permute(0, 2, 3, 1)
    return padded


def _augment(image: torch.Tensor, geometric: bool) -> torch.Tensor:
    """preprocessing_pytorch.py:52-142 on a [B, H, W, C] image in [-1, 1]."""
    image = image / 2.0 + 0.5
    dev = image.device
    if geometric:
        height, width = image.shape[1:3]
        ch, cw = int(height * 0.95), int(width * 0.95)
        max_h, max_w = height - ch, width - cw
        if max_h > 0 and max_w > 0:
            sh = int(torch.randint(0, max_h + 1, (1,), device="cpu"))
            sw = int(torch.randint(0, max_w + 1, (1,), device="cpu"))
            image = image[:, sh : sh + ch, sw : sw + cw, :]
        image = F.interpolate(image.permute(0, 3, 1, 2), size=(height, width), mode="bilinear", align_corners=False)
        angle = float(torch.rand(1) * 10 - 5)
        if abs(angle) > 0.1:
            rad = angle * torch.pi / 180.0
            cos_a, sin_a = float(np.cos(rad)), float(np.sin(rad))
            gx = torch.linspace(-1, 1, width, device=dev)
            gy = torch.linspace(-1, 1, height, device=dev)
            gy, gx = torch.meshgrid(gy, gx, indexing="ij")
            gx = gx.unsqueeze(0).expand(image.shape[0], -1, -1)
            gy = gy.unsqueeze(0).expand(image.shape[0], -1, -1)
            grid = torch.stack([gx * cos_a - gy * sin_a, gx * sin_a + gy * cos_a], dim=-1)
            image = F.grid_sample(image, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        image = image.permute(0, 2, 3, 1)
    image = image * (0.7 + float(torch.rand(1)) * 0.6)  # brightness
    mean = image.mean(dim=[1, 2, 3], keepdim=True)
    image = (image - mean) * (0.6 + float(torch.rand(1)) * 0.8) + mean  # contrast
    gray = image.mean(dim=-1, keepdim=True)
    image = gray + (image - gray) * (0.5 + float(torch.rand(1)) * 1.0)  # saturation
    return torch.clamp(image, 0, 1) * 2.0 - 1.0


class ProcessedObservation:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def preprocess_observation(observation, *, train: bool = False, image_keys: Sequence[str] = IMAGE_KEYS,
                           image_resolution: tuple[int, int] | None = None):  # fmt: skip
    """preprocessing_pytorch.py:20-173. `image_resolution=None` keeps whatever square resolution the images
    already have when it is not 224 (small test towers); the default model resizes to 224x224 with padding."""
    if not set(image_keys).issubset(observation.images):
        raise ValueError(f"images dict missing keys: expected {image_keys}, got {list(observation.images)}")
    batch_shape = observation.state.shape[:-1]
    out_images = {}
    for key in image_keys:
        image = observation.images[key]
        channels_first = image.shape[1] == 3
        if channels_first:
            image = image.permute(0, 2, 3, 1)
        if image_resolution is not None and tuple(image.shape[1:3]) != tuple(image_resolution):
            logger.info(f"Resizing image {key} from {image.shape[1:3]} to {image_resolution}")
            image = resize_with_pad_torch(image, *image_resolution)
        if train:
            image = _augment(image, geometric="wrist" not in key)
        if channels_first:
            image = image.permute(0, 3, 1, 2)
        out_images[key] = image
    out_masks = {}
    for key in out_images:
        if key not in observation.image_masks:
            out_masks[key] = torch.ones(batch_shape, dtype=torch.bool, device=observation.state.device)
        else:
            out_masks[key] = observation.image_masks[key]
    return ProcessedObservation(
        images=out_images,
        image_masks=out_masks,
        state=observation.state,
        tokenized_prompt=observation.tokenized_prompt,
        tokenized_prompt_mask=observation.tokenized_prompt_mask,
        token_ar_mask=getattr(observation, "token_ar_mask", None),
        token_loss_mask=getattr(observation, "token_loss_mask", None),
    )


def preprocess_observation_custom(observation, *, train: bool = False, image_resolution: tuple[int, int] | None = None,
                                  apply_aug: bool = True):  # fmt: skip
    """preprocessing_pytorch.py:175-330: like preprocess_observation, but over WHATEVER images the observation carries
    (keys `<part>_<timestep>_rgb`), ordered by (timestep, base < left_wrist < right_wrist), augmentation only if
    `train and apply_aug`; keeps the extra fields (progress, ...) of the observation."""
    order = {"base": 0, "left_wrist": 1, "right_wrist": 2}

    def sort_key(k):
        part, timestep, _ = k.rsplit("_", 2)
        return int(timestep), order[part]

    keys = sorted(observation.images.keys(), key=sort_key)
    out = preprocess_observation(observation, train=train and apply_aug, image_keys=keys, image_resolution=image_resolution)
    for name in ("progress", "episode_index", "frame_index", "episode_length", "image_original"):
        setattr(out, name, getattr(observation, name, None))
    return out
