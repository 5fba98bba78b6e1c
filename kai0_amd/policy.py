"""`Policy.infer` — the request handler around `sample_actions` (SURVEY.md §8b "Policy", §8 f2;
`src/openpi/policies/policy.py:25-122`, factory `policy_config.py:16-94`, interface
`packages/openpi-client/src/openpi_client/base_policy.py:5-12`).

One call = copy the observation dict -> input transforms (repack, default prompt, robot transform, normalise, resize,
tokenise, pad) -> add the batch axis and move to the device -> `Observation.from_dict` -> `model.sample_actions(device,
observation, noise=...)` -> first batch entry back to numpy -> output transforms (unnormalise, robot transform) ->
`policy_timing.infer_ms` (model time only, as the reference reports it).  Single in-flight request per Policy: the model's
static buffers and hipGraph are not re-entrant (SURVEY.md §8b "Threading")."""

from __future__ import annotations

import abc
import os
import time
from collections.abc import Mapping, Sequence
from typing import Any

import numpy as np
import torch

from . import transforms as _transforms
from .preprocessing import Observation


class BasePolicy(abc.ABC):
    @abc.abstractmethod
    def infer(self, obs: dict) -> dict:
        """Infer actions from observations."""

    def reset(self) -> None:
        pass


def _map_leaves(fn, tree):
    if isinstance(tree, Mapping):
        return {k: _map_leaves(fn, v) for k, v in tree.items()}
    return fn(tree)


class Policy(BasePolicy):
    def __init__(self, model, *, transforms: Sequence = (), output_transforms: Sequence = (),
                 sample_kwargs: dict[str, Any] | None = None, metadata: dict[str, Any] | None = None,
                 pytorch_device: str = "cuda:0", is_pytorch: bool = True, rng=None, device_resize: bool = True):  # fmt: skip
        if not is_pytorch:
            raise ValueError("kai0_amd.policy.Policy serves the torch-protocol model only (JAX checkpoints: convert first)")
        self._model = model.to(pytorch_device)
        self._model.eval()
        # Round 5: the camera frames' resize (transforms.ResizeImages -> Pillow, ~1 ms per 480 x 640 frame on the host: 16 % of a
        # request's wall time) moves behind the host-to-device copy when the model sits on a GPU: the raw uint8 frames cross PCIe and
        # kai0_amd.device_resize resamples them with Pillow's own fixed-point arithmetic (bit-identical).  Only when nothing after
        # ResizeImages in the stack looks at the images (the pi0.5 stack: TokenizePrompt, PadStatesAndActions).
        transforms = list(transforms)
        self._device_resize = None
        image_blind = (_transforms.TokenizePrompt, _transforms.PadStatesAndActions, _transforms.InjectDefaultPrompt)
        if device_resize == "force" or (device_resize and torch.device(pytorch_device).type == "cuda"):  # ("force": tests on the CPU)
            for i, t in enumerate(transforms):
                if isinstance(t, _transforms.ResizeImages) and all(isinstance(u, image_blind) for u in transforms[i + 1 :]):
                    self._device_resize = t
                    transforms = transforms[:i] + transforms[i + 1 :]
                    break
        self._input_transform = _transforms.compose(transforms)
        self._output_transform = _transforms.compose(output_transforms)
        self._sample_kwargs = sample_kwargs or {}
        self._metadata = metadata or {}
        self._pytorch_device = pytorch_device
        self._sample_actions = self._model.sample_actions

    def infer(self, obs: dict, *, noise: np.ndarray | None = None) -> dict:
        dev = self._pytorch_device
        inputs = _map_leaves(lambda x: x, obs)  # shallow structural copy: transforms may rebind entries
        inputs = self._input_transform(inputs)
        rs = self._device_resize
        if rs is not None and not all(isinstance(v, np.ndarray) and v.dtype == np.uint8 for v in inputs["image"].values()):
            inputs, rs = rs(inputs), None  # float frames: the host path, as before
        inputs = self._to_device(inputs, dev)
        if rs is not None:
            from .device_resize import resize_with_pad_u8

            frames = list(inputs["image"].values())
            if len({(tuple(f.shape), f.dtype) for f in frames}) == 1:  # the cameras of one robot: one batched resize instead of one per frame
                out = resize_with_pad_u8(torch.cat(frames, dim=0), rs.height, rs.width)
                inputs["image"] = {k: out[i : i + 1] for i, k in enumerate(inputs["image"])}
            else:
                inputs["image"] = {k: resize_with_pad_u8(v, rs.height, rs.width) for k, v in inputs["image"].items()}
        kwargs = dict(self._sample_kwargs)
        if noise is not None:
            n = torch.from_numpy(noise).to(dev)
            kwargs["noise"] = n[None, ...] if n.ndim == 2 else n
        observation = Observation.from_dict(inputs)
        t0 = time.monotonic()
        outputs = {"state": inputs["state"], "actions": self._sample_actions(dev, observation, **kwargs)}
        if torch.device(dev).type == "cuda":
            torch.cuda.synchronize()  # sample_actions returns as soon as the graph is launched
            stale = getattr(self._model, "inference_is_stale", None)
            if stale is not None and stale():  # the chunk was computed from weights edited in place: rebuilt and computed again
                outputs["actions"] = self._sample_actions(dev, observation, **kwargs)
                torch.cuda.synchronize()
        model_time = time.monotonic() - t0
        outputs = _map_leaves(lambda x: np.asarray(x[0, ...].detach().cpu()), outputs)
        outputs = self._output_transform(outputs)
        outputs["policy_timing"] = {"infer_ms": model_time * 1000}
        return outputs

    def _to_device(self, tree, dev):
        """Batch axis + host-to-device move of every leaf: plain (pageable, blocking) copies.  Round 6 measured the obvious alternative —
        per-leaf pinned staging buffers + `non_blocking=True` — and it is WORSE on this stack: every third request's closing
        `torch.cuda.synchronize()` then takes ~65 ms instead of ~15 (p50 unchanged, p90 4x; with the blocking copies 24 consecutive
        requests are all within 16.0-16.6 ms wall).  The ~10 leaves of a request cost 0.3 ms this way."""
        return _map_leaves(lambda x: torch.from_numpy(np.array(x)).to(dev)[None, ...], tree)

    @property
    def metadata(self) -> dict[str, Any]:
        return self._metadata


def create_policy(model, *, norm_stats: Mapping | None, tokenizer, action_dim: int = 32, use_quantile_norm: bool = True,
                  discrete_state_input: bool = True, image_size: int = 224, robot_inputs: Sequence = (),
                  robot_outputs: Sequence = (), repack_transforms: _transforms.Group | None = None,
                  default_prompt: str | None = None, sample_kwargs: dict | None = None, metadata: dict | None = None,
                  pytorch_device: str = "cuda:0", device_resize=True) -> Policy:  # fmt: skip
    """The transform stack of `create_trained_policy` (policy_config.py:75-94) with the pi0.5 model transforms of
    `ModelTransformFactory` (training/config.py:129-141), assembled from explicit pieces instead of a TrainConfig:
      inputs : repack -> default prompt -> robot inputs -> Normalize -> [default prompt, ResizeImages, TokenizePrompt,
               PadStatesAndActions]
      outputs: Unnormalize -> robot outputs -> repack outputs."""
    repack = repack_transforms or _transforms.Group()
    return Policy(
        model,
        transforms=[*repack.inputs, _transforms.InjectDefaultPrompt(default_prompt), *robot_inputs,
                    _transforms.Normalize(norm_stats, use_quantiles=use_quantile_norm),
                    _transforms.InjectDefaultPrompt(default_prompt), _transforms.ResizeImages(image_size, image_size),
                    _transforms.TokenizePrompt(tokenizer, discrete_state_input=discrete_state_input),
                    _transforms.PadStatesAndActions(action_dim)],
        output_transforms=[_transforms.Unnormalize(norm_stats, use_quantiles=use_quantile_norm), *robot_outputs, *repack.outputs],
        sample_kwargs=sample_kwargs, metadata=metadata, pytorch_device=pytorch_device, device_resize=device_resize)  # fmt: skip


def create_trained_policy(train_config, checkpoint_dir, *, repack_transforms: _transforms.Group | None = None,
                          sample_kwargs: dict[str, Any] | None = None, default_prompt: str | None = None,
                          norm_stats: Mapping | None = None, pytorch_device: str | None = None) -> Policy:  # fmt: skip
    """`policy_config.create_trained_policy` (policy_config.py:16-94) for torch checkpoints: `model.safetensors` in
    `checkpoint_dir` -> `train_config.model.load_pytorch` -> bf16 storage -> the transform stack of the reference:
      inputs : repack -> InjectDefaultPrompt -> data transforms -> Normalize -> model transforms
      outputs: model outputs -> Unnormalize -> data outputs -> repack outputs
    Norm stats come from `<checkpoint_dir>/assets/<asset_id>` (the stats the run was trained with) unless given.  A
    checkpoint directory without `model.safetensors` is a JAX checkpoint: convert it first (kai0_amd.convert)."""
    import os
    import pathlib

    from . import normalize as _normalize

    repack_transforms = repack_transforms or _transforms.Group()
    checkpoint_dir = pathlib.Path(checkpoint_dir)
    weight_path = os.path.join(checkpoint_dir, "model.safetensors")
    if not os.path.exists(weight_path):
        raise FileNotFoundError(f"{weight_path} not found: kai0_amd serves torch checkpoints (model.safetensors); a JAX `params` "
                                "checkpoint has to go through kai0_amd.convert first")  # fmt: skip
    model = train_config.model.load_pytorch(train_config, weight_path)
    model.paligemma_with_expert.to_bfloat16_for_selected_params("bfloat16")
    if hasattr(model, "trim_prompt_padding_infer") and os.environ.get("KAI0_TRIM_PROMPT", "1") != "0":
        # serve path: the prompt slots the request's prompt does not fill are not computed (model.py _trim_prompt; a task prompt
        # fills 10-40 of pi0.5's 200 slots: 18 % fewer prefix rows).  Same action chunk up to summation order.
        model.trim_prompt_padding_infer = True
    data_config = train_config.data.create(train_config.assets_dirs, train_config.model)
    if norm_stats is None:
        if data_config.asset_id is None:
            raise ValueError("Asset id is required to load norm stats.")
        norm_stats = _normalize.load(checkpoint_dir / "assets" / data_config.asset_id)  # checkpoints.load_norm_stats
    if pytorch_device is None:
        pytorch_device = "cuda" if torch.cuda.is_available() else "cpu"
    return Policy(
        model,
        transforms=[*repack_transforms.inputs, _transforms.InjectDefaultPrompt(default_prompt), *data_config.data_transforms.inputs,
                    _transforms.Normalize(norm_stats, use_quantiles=data_config.use_quantile_norm),
                    *data_config.model_transforms.inputs],
        output_transforms=[*data_config.model_transforms.outputs,
                           _transforms.Unnormalize(norm_stats, use_quantiles=data_config.use_quantile_norm),
                           *data_config.data_transforms.outputs, *repack_transforms.outputs],
        sample_kwargs=sample_kwargs, metadata=train_config.policy_metadata, is_pytorch=True, pytorch_device=pytorch_device)  # fmt: skip
