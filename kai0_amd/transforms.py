"""Per-sample host transforms on either side of `sample_actions` / `forward`: what `Policy.infer` and the
data loader run between a robot observation dict and the model's `Observation`.

Same names, fields, error behaviour and in-place conventions as `src/openpi/transforms.py` (class -> reference lines):
  Group :39-60, CompositeTransform / compose :63-78, RepackTransform :81-103, InjectDefaultPrompt :106-113,
  InsertAdvantageIntoPrompt :115-123, Normalize :126-157, Unnormalize :160-191, ResizeImages :194-201,
  SubsampleActions :204-210, DeltaActions :213-233, AbsoluteActions :236-256, TokenizePrompt :282-301,
  PromptFromLeRobotTask :342-356, PadStatesAndActions :359-369, flatten_dict / unflatten_dict :372-379 (the reference
  defers to flax.traverse_util, un-vendored; restated here for nested dicts with '/'-joined keys), transform_dict :382-437,
  apply_tree :440-457, pad_to_dim :460-468, make_bool_mask :471-491.
The pi0-FAST token transforms (:304-339) belong to a model family outside this path and are not provided.

Pure numpy; nothing here touches the GPU.  Parity: tests/test_transforms_cpu.py runs the reference's own known-answer tests
(transforms_test.py) against this module and compares with the reference's classes executed from source on random inputs.
"""

from __future__ import annotations

import dataclasses
import re
from collections.abc import Callable, Mapping, Sequence

import numpy as np

from . import image_tools
from . import normalize as _normalize
from .normalize import NormStats  # noqa: F401  (re-exported: `transforms.NormStats` as in the reference)


@dataclasses.dataclass(frozen=True)
class Group:
    inputs: Sequence[Callable] = ()
    outputs: Sequence[Callable] = ()

    def push(self, *, inputs: Sequence[Callable] = (), outputs: Sequence[Callable] = ()) -> "Group":
        """inputs go to the END of the input chain, outputs to the FRONT of the output chain."""
        return Group(inputs=(*self.inputs, *inputs), outputs=(*outputs, *self.outputs))


@dataclasses.dataclass(frozen=True)
class CompositeTransform:
    transforms: Sequence[Callable]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


def compose(transforms: Sequence[Callable]) -> CompositeTransform:
    return CompositeTransform(transforms)


# ------------------------------------------------------------------------------------------------------------- tree helpers
def flatten_dict(tree, sep: str = "/") -> dict:
    """{'a': {'b': 1}} -> {'a/b': 1}.  Only dicts are containers; an empty dict is kept as a leaf (as flax does)."""
    out = {}

    def walk(node, prefix):
        if isinstance(node, Mapping) and len(node) > 0:
            for k, v in node.items():
                walk(v, (*prefix, k))
        else:
            out[sep.join(map(str, prefix))] = node

    if not isinstance(tree, Mapping):
        raise TypeError(f"expected a mapping, got {type(tree)}")
    for k, v in tree.items():
        walk(v, (k,))
    return out


def unflatten_dict(flat: Mapping, sep: str = "/") -> dict:
    out: dict = {}
    for path, v in flat.items():
        keys = path.split(sep)
        node = out
        for k in keys[:-1]:
            node = node.setdefault(k, {})
        node[keys[-1]] = v
    return out


def _tree_map(fn, tree):
    if isinstance(tree, Mapping):
        return {k: _tree_map(fn, v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_tree_map(fn, v) for v in tree)
    return fn(tree)


def transform_dict(patterns: Mapping[str, str | None], tree) -> dict:
    """Rename / drop leaves by regular expressions over the flattened keys: the first pattern that FULLY matches a key
    wins; value None drops the leaf; values may use back-references."""
    data = flatten_dict(tree)
    compiled = {re.compile(k): v for k, v in patterns.items()}
    output = {}
    for k in data:
        for pattern, repl in compiled.items():
            if pattern.fullmatch(k):
                new_k = pattern.sub(repl, k, count=1) if repl is not None else None
                break
        else:
            new_k = k
        if new_k is not None:
            if new_k in output:
                raise ValueError(f"Key '{new_k}' already exists in output")
            output[new_k] = data[k]
    names = sorted(output)
    for name, next_name in zip(names, names[1:]):
        if next_name.startswith(name + "/"):
            raise ValueError(f"Leaf '{name}' aliases a node of '{next_name}'")
    return unflatten_dict(output)


def apply_tree(tree, selector, fn: Callable, *, strict: bool = False):
    """fn(leaf, selector_leaf) on every leaf of `tree` whose flattened key is in `selector`."""
    tree = flatten_dict(tree)
    selector = flatten_dict(selector)
    if strict:
        for k in selector:
            if k not in tree:
                raise ValueError(f"Selector key {k} not found in tree")
    return unflatten_dict({k: (fn(v, selector[k]) if k in selector else v) for k, v in tree.items()})


def pad_to_dim(x: np.ndarray, target_dim: int, axis: int = -1, value: float = 0.0) -> np.ndarray:
    cur = x.shape[axis]
    if cur < target_dim:
        pad = [(0, 0)] * len(x.shape)
        pad[axis] = (0, target_dim - cur)
        return np.pad(x, pad, constant_values=value)
    return x


def make_bool_mask(*dims: int) -> tuple[bool, ...]:
    """make_bool_mask(2, -2, 2) == (True, True, False, False, True, True); 0 contributes nothing."""
    out: list[bool] = []
    for d in dims:
        out.extend([True] * d if d > 0 else [False] * (-d))
    return tuple(out)


def _assert_quantile_stats(norm_stats) -> None:
    for k, v in flatten_dict(norm_stats).items():
        if v.q01 is None or v.q99 is None:
            raise ValueError(f"quantile stats must be provided if use_quantile_norm is True. Key {k} is missing q01 or q99.")


# --------------------------------------------------------------------------------------------------------------- transforms
@dataclasses.dataclass(frozen=True)
class RepackTransform:
    """structure: new nested keys -> '/'-joined paths into the input, e.g. {"state": "observation.state"}."""

    structure: Mapping

    def __call__(self, data):
        flat = flatten_dict(data)
        return _tree_map(lambda k: flat[k], self.structure)


@dataclasses.dataclass(frozen=True)
class InjectDefaultPrompt:
    prompt: str | None

    def __call__(self, data):
        if self.prompt is not None and "prompt" not in data:
            data["prompt"] = np.asarray(self.prompt)
        return data


@dataclasses.dataclass(frozen=True)
class InsertAdvantageIntoPrompt:
    def __call__(self, data):
        assert "advantage" in data, f"advantage is not in data, data_keys: {data.keys()}"
        assert "prompt" in data, f"prompt is not in data, data_keys: {data.keys()}"
        data["prompt"] = data["prompt"] + f", Advantage: {data['advantage']:.4f}"
        return data


class _StatsMap:
    """Shared by Normalize / Unnormalize: apply one of the two affine maps of kai0_amd.normalize to every leaf that has
    statistics under the same flattened key."""

    norm_stats: Mapping | None
    use_quantiles: bool

    def __post_init__(self):
        if self.norm_stats is not None and self.use_quantiles:
            _assert_quantile_stats(self.norm_stats)

    def _run(self, data, fn, strict):
        if self.norm_stats is None:
            return data
        q = self.use_quantiles
        return apply_tree(data, self.norm_stats, lambda x, st: fn(x, st, use_quantiles=q), strict=strict)


@dataclasses.dataclass(frozen=True)
class Normalize(_StatsMap):
    """z-score `(x - mean) / (std + 1e-6)` or quantile `(x - q01) / (q99 - q01 + 1e-6) * 2 - 1`; statistics longer than the
    vector are cut to it.  strict: every statistics key must be present in the data."""

    norm_stats: Mapping | None
    use_quantiles: bool = False
    strict: bool = False

    def __call__(self, data):
        return self._run(data, _normalize.normalize, self.strict)


@dataclasses.dataclass(frozen=True)
class Unnormalize(_StatsMap):
    """Inverse maps; always strict.  z-score statistics shorter than the vector are padded with mean 0 / std 1, quantile
    statistics leave the extra trailing dimensions untouched."""

    norm_stats: Mapping | None
    use_quantiles: bool = False

    def __call__(self, data):
        return self._run(data, _normalize.unnormalize, True)


@dataclasses.dataclass(frozen=True)
class ResizeImages:
    height: int
    width: int

    def __call__(self, data):
        data["image"] = {k: image_tools.resize_with_pad(v, self.height, self.width) for k, v in data["image"].items()}
        return data


@dataclasses.dataclass(frozen=True)
class SubsampleActions:
    stride: int

    def __call__(self, data):
        data["actions"] = data["actions"][:: self.stride]
        return data


def _shift_actions(data, mask, sign: int):
    """actions[..., :len(mask)] += sign * state[..., :len(mask)] on the masked dimensions, in place (callers rely on the
    array identity); the input dict itself is returned untouched when there is nothing to do."""
    if mask is None or "actions" not in data:
        return data
    m = np.asarray(mask)
    n = m.shape[-1]
    offset = np.where(m, data["state"][..., :n], 0)[..., None, :]
    acts = data["actions"]
    if sign > 0:
        acts[..., :n] += offset
    else:
        acts[..., :n] -= offset
    data["actions"] = acts
    return data


@dataclasses.dataclass(frozen=True)
class DeltaActions:
    """Absolute -> delta action space for the dimensions selected by `mask` (see make_bool_mask)."""

    mask: Sequence[bool] | None

    def __call__(self, data):
        return _shift_actions(data, self.mask, -1)


@dataclasses.dataclass(frozen=True)
class AbsoluteActions:
    """Delta -> absolute action space."""

    mask: Sequence[bool] | None

    def __call__(self, data):
        return _shift_actions(data, self.mask, +1)


@dataclasses.dataclass(frozen=True)
class TokenizePrompt:
    """prompt (+ the discretised state for pi0.5) -> tokenized_prompt / tokenized_prompt_mask."""

    tokenizer: object
    discrete_state_input: bool = False

    def __call__(self, data):
        if (prompt := data.pop("prompt", None)) is None:
            raise ValueError("Prompt is required")
        if self.discrete_state_input:
            if (state := data.get("state", None)) is None:
                raise ValueError("State is required.")
        else:
            state = None
        if not isinstance(prompt, str):
            prompt = prompt.item()
        tokens, token_masks = self.tokenizer.tokenize(prompt, state)
        return {**data, "tokenized_prompt": tokens, "tokenized_prompt_mask": token_masks}


@dataclasses.dataclass(frozen=True)
class PromptFromLeRobotTask:
    tasks: dict

    def __call__(self, data):
        if "task_index" not in data:
            raise ValueError('Cannot extract prompt without "task_index"')
        task_index = int(data["task_index"])
        if (prompt := self.tasks.get(task_index)) is None:
            raise ValueError(f"{task_index=} not found in task mapping: {self.tasks}")
        return {**data, "prompt": prompt}


@dataclasses.dataclass(frozen=True)
class PadStatesAndActions:
    model_action_dim: int

    def __call__(self, data):
        data["state"] = pad_to_dim(data["state"], self.model_action_dim, axis=-1)
        if "actions" in data:
            data["actions"] = pad_to_dim(data["actions"], self.model_action_dim, axis=-1)
        return data
