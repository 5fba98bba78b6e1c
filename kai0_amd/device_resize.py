"""Aspect-preserving bilinear resize with zero padding of uint8 HWC camera frames ON THE DEVICE, bit-identical to the Pillow
resampling the reference's serve path runs on the host (`packages/openpi-client/src/openpi_client/image_tools.py:7-58` ->
`PIL.Image.resize(..., BILINEAR)`; kai0_amd/image_tools.py is the host version).

Why: at ~16 ms of model time per action chunk the three 480 x 640 -> 224 x 224 Pillow resizes of a request cost ~3 ms of host time
(profiles/r04_policy_latency.json: 16 % of the request's wall time).  Here the raw frames cross PCIe once (2.7 MB instead of 0.45 MB:
~50 us) and are resampled by a handful of torch gather / multiply / shift kernels.

Pillow's algorithm for 8-bit channels (src/libImaging/Resample.c), restated:
  * per axis, output sample xx looks at the input interval centred at (xx + 0.5) * scale with half-width support = max(scale, 1)
    (the triangle filter stretched by the downscale factor: antialiasing), taps xmin .. xmin + n - 1 clipped to the image, weights
    normalised to sum 1 in double precision (`precompute_coeffs`);
  * the weights become 22-bit fixed point, rounded half away from zero (`normalize_coeffs_8bpc`, PRECISION_BITS = 32 - 8 - 2);
  * a pass computes clip8((2^21 + sum_k pixel_k * w_k) >> 22) in int32; the horizontal pass runs first and its uint8 result feeds the
    vertical pass.
Everything after the coefficient tables (built on the host in float64, a few hundred numbers per size pair, cached) is integer
arithmetic, so the device result is bit-identical by construction; tests/test_host_pipeline_cpu.py checks it against Pillow."""

from __future__ import annotations

import functools
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


@functools.lru_cache(maxsize=64)
def _pil_bilinear_taps(in_size: int, out_size: int):
    """(index [out, K] int64, weight [out, K] int32) of Pillow's BILINEAR resampling of `in_size` samples to `out_size`."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size  # (double)(in1 - in0) / outSize with float box edges
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    idx = np.zeros((out_size, ksize), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.empty(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS))).astype(np.int64)  # (int): toward zero
        kk[xx, :xmax] = fixed.astype(np.int32)
        idx[xx, :xmax] = np.arange(xmin, xmin + xmax)
        idx[xx, xmax:] = xmin  # unused taps: weight 0, any valid index
    return idx, kk


@functools.lru_cache(maxsize=64)
def _taps_on(in_size: int, out_size: int, device: str):
    """The tap tables on `device`, copied once per (size pair, device): per request they were 4 small pageable host-to-device copies
    per frame, each a synchronisation (round 6)."""
    idx_np, kk_np = _pil_bilinear_taps(in_size, out_size)
    return torch.from_numpy(idx_np).to(device), torch.from_numpy(kk_np).to(device)


def _resample_axis(img: torch.Tensor, out_size: int, axis: int) -> torch.Tensor:
    """One Pillow pass over `axis` of an int32 tensor [N, H, W, C] holding uint8 values."""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    idx, kk = _taps_on(in_size, out_size, str(img.device))
    K = idx.shape[1]
    g = img.index_select(axis, idx.reshape(-1))  # [.., out * K, ..]
    shape = list(img.shape)
    shape[axis : axis + 1] = [out_size, K]
    g = g.reshape(shape)
    wshape = [1] * len(shape)
    wshape[axis], wshape[axis + 1] = out_size, K
    acc = (g * kk.reshape(wshape)).sum(axis + 1, dtype=torch.int32) + (1 << (PRECISION_BITS - 1))
    return torch.bitwise_right_shift(acc, PRECISION_BITS).clamp_(0, 255)


def fit_size(h0: int, w0: int, height: int, width: int) -> tuple[int, int]:
    """(h1, w1) of the resized content inside a height x width canvas: image_tools._fit's truncating sizes."""
    scale = max(w0 / width, h0 / height)
    return int(h0 / scale), int(w0 / scale)


def resize_with_pad_u8(images: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """uint8 [..., h, w, c] -> uint8 [..., height, width, c] on the tensor's device: the content resized with Pillow's BILINEAR
    arithmetic and centred, the border zero — `image_tools.resize_with_pad` bit for bit.  Already-sized input is returned as is."""
    if images.dtype != torch.uint8:
        raise TypeError(f"resize_with_pad_u8: uint8 frames expected, got {images.dtype}")
    h0, w0 = images.shape[-3], images.shape[-2]
    if (h0, w0) == (height, width):
        return images
    lead = images.shape[:-3]
    x = images.reshape(-1, h0, w0, images.shape[-1]).to(torch.int32)
    h1, w1 = fit_size(h0, w0, height, width)
    x = _resample_axis(x, w1, 2)  # horizontal pass first; its clipped 8-bit result feeds the vertical pass
    x = _resample_axis(x, h1, 1)
    out = torch.zeros((x.shape[0], height, width, x.shape[-1]), dtype=torch.uint8, device=images.device)
    top, left = max(0, int((height - h1) / 2)), max(0, int((width - w1) / 2))
    out[:, top : top + h1, left : left + w1] = x.to(torch.uint8)
    return out.reshape(*lead, height, width, out.shape[-1])
