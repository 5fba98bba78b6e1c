"""Client / host side image helpers: aspect-preserving resize with zero padding on uint8 HWC images, as the
serve path does before the model sees a frame (`packages/openpi-client/src/openpi_client/image_tools.py:7-58`; the torch
NCHW / float variant used inside the model path lives in kai0_amd/preprocessing.py).

PIL does the resampling (bilinear by default), so results are bit-identical to the reference's for the same Pillow:
tests/test_host_pipeline_cpu.py compares against vectors produced by the reference module itself
(tests/golden/make_host_pipeline_golden.py)."""

from __future__ import annotations

import numpy as np
from PIL import Image


def convert_to_uint8(img: np.ndarray) -> np.ndarray:
    """Float images in [0, 1] become uint8 (truncating, like the reference: 255 * x then astype); others pass through."""
    return (255 * img).astype(np.uint8) if np.issubdtype(img.dtype, np.floating) else img


def _fit(image: Image.Image, height: int, width: int, method) -> Image.Image:
    w0, h0 = image.size
    if (w0, h0) == (width, height):
        return image
    scale = max(w0 / width, h0 / height)          # shrink (or grow) until the larger relative side fits
    w1, h1 = int(w0 / scale), int(h0 / scale)     # truncation, not rounding: tf.image.resize_with_pad's sizes
    canvas = Image.new(image.mode, (width, height), 0)
    canvas.paste(image.resize((w1, h1), resample=method), (max(0, int((width - w1) / 2)), max(0, int((height - h1) / 2))))
    return canvas


def resize_with_pad(images: np.ndarray, height: int, width: int, method=Image.BILINEAR) -> np.ndarray:
    """[..., h, w, c] uint8 -> [..., height, width, c]; the content is centred, the border is zero.  Already-sized input
    is returned as is (same object)."""
    if images.shape[-3:-1] == (height, width):
        return images
    lead = images.shape[:-3]
    frames = images.reshape(-1, *images.shape[-3:])
    out = np.stack([np.asarray(_fit(Image.fromarray(f), height, width, method)) for f in frames])
    return out.reshape(*lead, *out.shape[-3:])
