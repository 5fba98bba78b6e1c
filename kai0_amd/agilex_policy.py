"""Robot-side <-> model-side dictionaries for the Agilex dual-arm platform kai0 ships:
`AgilexInputs` / `AgilexOutputs` of `src/openpi/policies/agilex_policy.py:13-147`.

Input  : {"images": {camera: CHW or HWC, float [0, 1] or uint8}, "state": [14], optional "actions" [H, 14], "prompt",
          and the advantage-estimator extras}.
Output : {"image": {model key: HWC uint8}, "image_mask": {model key: True}, "state": [action_dim] zero-padded, ...}.
Rules kept from the reference: the three current-frame cameras are mandatory, the three "his_-100_*" history cameras are
optional (advantage estimator); unknown cameras are an error; joint values outside [-pi, pi] are sensor glitches and become
0; pi0 (not pi0.5) additionally emits an `action_mask` over the padded action dimensions."""

from __future__ import annotations

import dataclasses
from typing import ClassVar

import numpy as np
import torch

from . import transforms

PI0_TYPES = ("pi0", "pi0_rtc")


def _model_type_name(model_type) -> str:
    return str(getattr(model_type, "value", model_type)).lower()


def _clip_glitches(x: np.ndarray) -> np.ndarray:
    return np.where(np.abs(x) > np.pi, 0, x)


@dataclasses.dataclass(frozen=True)
class AgilexInputs:
    action_dim: int
    model_type: object = "pi0"
    mask_state: bool = False  # feed an all-zero state
    filter_state_glitches: ClassVar[bool] = True

    required_rename_map: ClassVar[dict] = {"top_head": "base_0_rgb", "hand_left": "left_wrist_0_rgb",
                                           "hand_right": "right_wrist_0_rgb"}  # fmt: skip
    optional_rename_map: ClassVar[dict] = {"his_-100_top_head": "base_-100_rgb", "his_-100_hand_left": "left_wrist_-100_rgb",
                                           "his_-100_hand_right": "right_wrist_-100_rgb"}  # fmt: skip
    all_rename_map: ClassVar[dict] = {**required_rename_map, **optional_rename_map}
    EXPECTED_CAMERAS: ClassVar[tuple] = tuple(required_rename_map)
    EXTRA_CAMERAS: ClassVar[tuple] = tuple(optional_rename_map)
    PASSTHROUGH: ClassVar[tuple] = ("frame_index", "episode_length", "progress", "image_original", "episode_index")

    @staticmethod
    def _to_hwc_uint8(img):
        if isinstance(img, torch.Tensor):
            img = img.cpu().numpy()
        if np.issubdtype(img.dtype, np.floating):
            img = (255 * img).astype(np.uint8)
        return np.transpose(img, (1, 2, 0)) if img.shape[0] == 3 else img

    def __call__(self, data: dict) -> dict:
        cams = data["images"]
        if set(cams) - set(self.all_rename_map):
            raise ValueError(f"Expected images to contain {self.EXPECTED_CAMERAS}, got {tuple(cams)}")
        images, masks = {}, {}
        for cam in self.EXPECTED_CAMERAS + self.EXTRA_CAMERAS:
            if cam in cams:
                images[self.all_rename_map[cam]] = self._to_hwc_uint8(cams[cam])
                masks[self.all_rename_map[cam]] = np.True_
            elif cam in self.EXPECTED_CAMERAS:
                raise ValueError(f"Camera {cam} not found in data")
        state = transforms.pad_to_dim(data["state"], self.action_dim).squeeze()
        if self.filter_state_glitches:
            state = _clip_glitches(state)
        out = {"image": images, "image_mask": masks, "state": np.zeros_like(state) if self.mask_state else state}
        if "actions" in data:
            actions = _clip_glitches(transforms.pad_to_dim(data["actions"], self.action_dim))
            if _model_type_name(self.model_type) in PI0_TYPES:
                mask = np.ones_like(actions, dtype=bool)
                mask[:, self.action_dim :] = False
                out["action_mask"] = mask
            out["actions"] = actions.squeeze()
        if "prompt" in data:
            out["prompt"] = data["prompt"]
        for key in self.PASSTHROUGH:
            if key in data:
                out[key] = data[key]
        for key in ("action_advantage", "action_advantage_original"):
            if key in data:
                v = data[key]
                if v is None and key == "action_advantage":
                    out[key] = torch.tensor(1.0)
                elif isinstance(v, np.ndarray):
                    out[key] = torch.from_numpy(v)
                elif isinstance(v, torch.Tensor):
                    out[key] = v.detach().clone()
                else:
                    raise NotImplementedError(f"Unsupported type: {type(v)}")
        return out


@dataclasses.dataclass(frozen=True)
class AgilexOutputs:
    """The robot takes the first 14 of the model's 32 action dimensions (2 x (6 joints + gripper))."""

    def __call__(self, data: dict) -> dict:
        return {"actions": np.asarray(data["actions"][:, :14])}


@dataclasses.dataclass(frozen=True)
class ARXInputs(AgilexInputs):
    """`src/openpi/policies/arx_policy.py`: the ARX platform (HangCloth) — the Agilex mapping without the state glitch filter
    (the action filter stays, arx_policy.py:100-103)."""

    filter_state_glitches: ClassVar[bool] = False


@dataclasses.dataclass(frozen=True)
class ARXOutputs(AgilexOutputs):
    pass
