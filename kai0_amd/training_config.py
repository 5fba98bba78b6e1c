"""`TrainConfig` / `get_config(name)` / `cli()` — the configuration surface of `src/openpi/training/config.py` that the pi0.5 path
reads (SURVEY.md §8b "Config"): `TrainConfig` fields of config.py:656-757, `DataConfig` / `DataConfigFactory` and the Agilex /
ARX factories (config.py:67-102,175-232,361-545), `ModelTransformFactory` (config.py:108-171, pi0.5 branch), the schedule and
optimizer configs (training/optimizer.py:15-85), and the registry with kai0's pi0.5 entries (config.py:1137-1376).

Same names, field meanings and errors as the reference, so `policy_config.create_trained_policy(get_config(name), ckpt_dir)`,
`scripts/train_pytorch.py`-style drivers and kai0's `model_arithmetic` / `stage_advantage` modules read it unchanged.  Not carried
over (outside the hot path): tyro's CLI machinery (`cli()` is a small argparse front end with the same
`<config-name> --field value` shape), the JAX-side `weight_loader.load` / `freeze_filter` / `ema_decay` behaviour (kept as plain
data), RLDS / DROID / LIBERO / ALOHA factories, wandb."""

from __future__ import annotations

import abc
import argparse
import dataclasses
import difflib
import logging
import pathlib
from collections.abc import Sequence
from typing import Any, Literal

from . import agilex_policy
from . import normalize as _normalize
from . import transforms as _transforms
from .config import AdvantageEstimatorConfig, Pi0Config
from .optim import lr_schedule as _lr_schedule


# ------------------------------------------------------------------------------- training/optimizer.py:15-85
@dataclasses.dataclass(frozen=True)
class CosineDecaySchedule:
    """Warm-up + cosine decay (optimizer.py:15-32); `create()` returns step -> lr (train_pytorch.py:483-491)."""

    warmup_steps: int = 1_000
    peak_lr: float = 2.5e-5
    decay_steps: int = 30_000
    decay_lr: float = 2.5e-6

    def create(self):
        return lambda step: _lr_schedule(int(step), warmup_steps=self.warmup_steps, peak_lr=self.peak_lr,
                                         decay_steps=self.decay_steps, end_lr=self.decay_lr)  # fmt: skip


@dataclasses.dataclass(frozen=True)
class AdamW:
    """optimizer.py:66-85 — clip by global norm, then AdamW."""

    b1: float = 0.9
    b2: float = 0.95
    eps: float = 1e-8
    weight_decay: float = 1e-10
    clip_gradient_norm: float = 1.0


# -------------------------------------------------------------------------- training/weight_loaders.py:31-55
@dataclasses.dataclass(frozen=True)
class NoOpWeightLoader:
    def load(self, params):
        return params


@dataclasses.dataclass(frozen=True)
class CheckpointWeightLoader:
    """Names an Orbax `params` directory (JAX checkpoints).  The torch path takes converted weights: `kai0_amd.convert`
    turns the restored numpy tree into this model's state dict; `pytorch_weight_path` names the converted checkpoint."""

    params_path: str

    def load(self, params):
        raise NotImplementedError("CheckpointWeightLoader.load restores a JAX param tree (orbax); on the torch path convert the "
                                  "checkpoint with kai0_amd.convert and set `pytorch_weight_path`")  # fmt: skip


# ------------------------------------------------------------------------------------ config.py:48-102
@dataclasses.dataclass(frozen=True)
class AssetsConfig:
    assets_dir: str | None = None
    asset_id: str | None = None


@dataclasses.dataclass(frozen=True)
class DataConfig:
    repo_id: str | None = None
    asset_id: str | None = None
    norm_stats: dict[str, _normalize.NormStats] | None = None
    repack_transforms: _transforms.Group = dataclasses.field(default_factory=_transforms.Group)
    data_transforms: _transforms.Group = dataclasses.field(default_factory=_transforms.Group)
    model_transforms: _transforms.Group = dataclasses.field(default_factory=_transforms.Group)
    use_quantile_norm: bool = False
    action_sequence_keys: Sequence[str] = ("actions",)
    prompt_from_task: bool = False
    episodes: list[int] | None = None


@dataclasses.dataclass(frozen=True)
class ModelTransformFactory:
    """config.py:108-171, pi0 / pi0.5 branches.  `tokenizer_model`: the PaliGemma sentencepiece model (path, bytes or
    processor); None -> $KAI0_PALIGEMMA_TOKENIZER (the reference downloads it from GCS; nothing is downloaded here)."""

    default_prompt: str | None = None
    tokenizer_model: Any = None

    def __call__(self, model_config) -> _transforms.Group:
        from .tokenizer import PaligemmaTokenizer

        mt = model_config.model_type
        if mt not in ("pi0", "pi05"):
            raise NotImplementedError(f"model transforms for model_type {mt!r} are outside the pi0.5 hot path")
        tok = PaligemmaTokenizer(model_config.max_token_len, model=self.tokenizer_model)
        kw = {"discrete_state_input": model_config.discrete_state_input} if mt == "pi05" else {}
        return _transforms.Group(inputs=[
            _transforms.InjectDefaultPrompt(self.default_prompt),
            _transforms.ResizeImages(224, 224),
            _transforms.TokenizePrompt(tok, **kw),
            _transforms.PadStatesAndActions(model_config.action_dim),
        ])  # fmt: skip


_MISSING = "<missing>"  # tyro.MISSING


@dataclasses.dataclass(frozen=True)
class DataConfigFactory(abc.ABC):
    repo_id: str = _MISSING
    assets: AssetsConfig = dataclasses.field(default_factory=AssetsConfig)
    base_config: DataConfig | None = None

    @abc.abstractmethod
    def create(self, assets_dirs: pathlib.Path, model_config) -> DataConfig: ...

    def create_base_config(self, assets_dirs: pathlib.Path, model_config) -> DataConfig:
        repo_id = self.repo_id if self.repo_id != _MISSING else None
        asset_id = self.assets.asset_id or repo_id
        return dataclasses.replace(
            self.base_config or DataConfig(), repo_id=repo_id, asset_id=asset_id,
            norm_stats=self._load_norm_stats(pathlib.Path(self.assets.assets_dir or assets_dirs), asset_id),
            use_quantile_norm=model_config.model_type not in ("pi0", "pi0_rtc"))  # fmt: skip

    def _load_norm_stats(self, assets_dir: pathlib.Path, asset_id: str | None):
        if asset_id is None:
            return None
        data_assets_dir = assets_dir / asset_id
        try:
            stats = _normalize.load(data_assets_dir)
            logging.info(f"Loaded norm stats from {data_assets_dir}")
            return stats
        except FileNotFoundError:
            logging.info(f"Norm stats not found in {data_assets_dir}, skipping.")
        return None


@dataclasses.dataclass(frozen=True)
class FakeDataConfig(DataConfigFactory):
    repo_id: str = "fake"

    def create(self, assets_dirs, model_config) -> DataConfig:
        return DataConfig(repo_id=self.repo_id)


def _agilex_repack() -> _transforms.Group:
    return _transforms.Group(inputs=[_transforms.RepackTransform({
        "images": {"top_head": "observation.images.top_head", "hand_left": "observation.images.hand_left",
                   "hand_right": "observation.images.hand_right"},
        "state": "observation.state", "actions": "action"})])  # fmt: skip


@dataclasses.dataclass(frozen=True)
class LerobotAgilexDataConfig(DataConfigFactory):
    """config.py:361-453: the Agilex dual-arm dataset (three cameras, 14-DoF state / action)."""

    use_delta_joint_actions: bool = True
    default_prompt: str | None = None
    episodes: list[int] | None = None
    repack_transforms: _transforms.Group = dataclasses.field(default_factory=_agilex_repack)
    action_sequence_keys: Sequence[str] = ("action",)
    mask_state: bool = False
    insert_advantage_into_prompt: bool = False
    tokenizer_model: Any = None  # see ModelTransformFactory

    _inputs_cls = agilex_policy.AgilexInputs
    _outputs_cls = agilex_policy.AgilexOutputs

    def create(self, assets_dirs, model_config) -> DataConfig:
        repack = self.repack_transforms
        if self.base_config and self.base_config.prompt_from_task:  # the dataset's task string becomes the prompt
            structure = dict(repack.inputs[0].structure)
            structure["prompt"] = "prompt"
            repack = _transforms.Group(inputs=[_transforms.RepackTransform(structure)])
        data = _transforms.Group(
            inputs=[self._inputs_cls(action_dim=model_config.action_dim, model_type=model_config.model_type, mask_state=self.mask_state)],
            outputs=[self._outputs_cls()])  # fmt: skip
        if self.insert_advantage_into_prompt:
            data = _transforms.Group(inputs=[_transforms.InsertAdvantageIntoPrompt(), *data.inputs], outputs=data.outputs)
        if self.use_delta_joint_actions:
            mask = _transforms.make_bool_mask(6, -1, 6, -1)  # joints as deltas, the two grippers (6, 13) absolute
            data = data.push(inputs=[_transforms.DeltaActions(mask)], outputs=[_transforms.AbsoluteActions(mask)])
        # the reference passes self.default_prompt here even when prompt_from_task cleared the local copy (config.py:445)
        model = ModelTransformFactory(default_prompt=self.default_prompt, tokenizer_model=self.tokenizer_model)(model_config)
        return dataclasses.replace(self.create_base_config(assets_dirs, model_config), repack_transforms=repack,
                                   data_transforms=data, model_transforms=model,
                                   action_sequence_keys=self.action_sequence_keys, episodes=self.episodes)  # fmt: skip


@dataclasses.dataclass(frozen=True)
class LerobotARXDataConfig(LerobotAgilexDataConfig):
    """config.py:456-545: the ARX platform (HangCloth) — same layout, no state glitch filter (arx_policy.py)."""

    _inputs_cls = agilex_policy.ARXInputs
    _outputs_cls = agilex_policy.ARXOutputs


# ------------------------------------------------------------------------------------ config.py:656-757
@dataclasses.dataclass(frozen=True)
class TrainConfig:
    name: str
    project_name: str = "openpi"
    exp_name: str = _MISSING
    model: Pi0Config = dataclasses.field(default_factory=Pi0Config)
    weight_loader: Any = dataclasses.field(default_factory=NoOpWeightLoader)
    pytorch_weight_path: str | None = None
    pytorch_training_precision: Literal["bfloat16", "float32"] = "bfloat16"
    lr_schedule: CosineDecaySchedule = dataclasses.field(default_factory=CosineDecaySchedule)
    optimizer: AdamW = dataclasses.field(default_factory=AdamW)
    ema_decay: float | None = 0.99
    freeze_filter: Any = None
    data: DataConfigFactory = dataclasses.field(default_factory=FakeDataConfig)
    assets_base_dir: str = "./assets"
    checkpoint_base_dir: str = "./checkpoints"
    seed: int = 42
    batch_size: int = 32
    num_workers: int = 2
    num_train_steps: int = 30_000
    log_interval: int = 100
    save_interval: int = 1000
    advantage_estimator: bool = False
    is_train: bool = True
    split: str = "all"
    drop_last: bool = True
    skip_norm_stats: bool = False
    keep_period: int | None = 5000
    overwrite: bool = False
    resume: bool = False
    wandb_enabled: bool = True
    policy_metadata: dict[str, Any] | None = None
    fsdp_devices: int = 1

    @property
    def assets_dirs(self) -> pathlib.Path:
        return (pathlib.Path(self.assets_base_dir) / self.name).resolve()

    @property
    def checkpoint_dir(self) -> pathlib.Path:
        if not self.exp_name or self.exp_name == _MISSING:
            raise ValueError("--exp_name must be set")
        return (pathlib.Path(self.checkpoint_base_dir) / self.name / self.exp_name).resolve()

    def __post_init__(self) -> None:
        if self.resume and self.overwrite:
            raise ValueError("Cannot resume and overwrite at the same time.")


# ------------------------------------------------------------------------------------ config.py:761-1394 (pi0.5 / kai0 entries)
def _advantage_repack() -> _transforms.Group:
    return _transforms.Group(inputs=[_transforms.RepackTransform({
        "images": {"top_head": "observation.images.top_head", "hand_left": "observation.images.hand_left",
                   "hand_right": "observation.images.hand_right",
                   "his_-100_top_head": "his_-100_observation.images.top_head",
                   "his_-100_hand_left": "his_-100_observation.images.hand_left",
                   "his_-100_hand_right": "his_-100_observation.images.hand_right"},
        "state": "observation.state", "actions": "action", "episode_length": "episode_length", "frame_index": "frame_index",
        "episode_index": "episode_index", "progress_gt": "progress_gt", "stage_progress_gt": "stage_progress_gt",
        "progress": "progress"})])  # fmt: skip


def _kai0_task(name: str, factory, task_dir: str, prompt: str, *, awbc: bool) -> TrainConfig:
    sub = "advantage" if awbc else "base"
    return TrainConfig(
        name=name, model=Pi0Config(pi05=True),
        data=factory(repo_id=f"<path_to_repo_root>/data/{task_dir}/{sub}", default_prompt=prompt, use_delta_joint_actions=False,
                     base_config=DataConfig(prompt_from_task=True) if awbc else None),
        weight_loader=CheckpointWeightLoader("<path/to/pi05_base/checkpoint>"),
        num_train_steps=100_000, keep_period=5000, num_workers=8, batch_size=256)  # fmt: skip


_CONFIGS = [
    TrainConfig(name="debug", data=FakeDataConfig(), batch_size=2,
                model=Pi0Config(pi05=False, paligemma_variant="dummy", action_expert_variant="dummy"), save_interval=100,
                overwrite=True, exp_name="debug", num_train_steps=10, wandb_enabled=False),  # fmt: skip
    TrainConfig(name="debug_pi05", model=Pi0Config(pi05=True, paligemma_variant="dummy", action_expert_variant="dummy"),
                data=FakeDataConfig(), batch_size=2, num_train_steps=10, overwrite=True, exp_name="debug_pi05",
                wandb_enabled=False),  # fmt: skip
    # normal pi0.5 full fine-tuning (config.py:1176-1218)
    _kai0_task("pi05_flatten_fold_normal", LerobotAgilexDataConfig, "FlattenFold", "Flatten and fold the cloth.", awbc=False),
    _kai0_task("pi05_tee_shirt_sort_normal", LerobotAgilexDataConfig, "TeeShirtSort",
               "Fetch the clothes, fold the tee shirts and hand-over the collared shirts.", awbc=False),
    _kai0_task("pi05_hang_cloth_normal", LerobotARXDataConfig, "HangCloth", "Fetch and hang the cloth.", awbc=False),
    # Stage-Advantage estimator (config.py:1220-1272)
    TrainConfig(
        name="ADVANTAGE_TORCH_KAI0_FLATTEN_FOLD", advantage_estimator=True,
        model=AdvantageEstimatorConfig(pi05=True, loss_value_weight=1.0, loss_action_weight=0.0, discrete_state_input=False),
        data=LerobotAgilexDataConfig(repo_id="Path/to/your/advantage/dataset",
                                     assets=AssetsConfig(assets_dir="Path/to/your/advantage/dataset/assets",
                                                         asset_id="Your_advantage_dataset_name"),
                                     default_prompt="Flatten and fold the cloth.", repack_transforms=_advantage_repack()),
        pytorch_weight_path="Path/to/your/pi05_base/checkpoint", num_train_steps=100_000, keep_period=10000,
        save_interval=10000, num_workers=8, batch_size=16, skip_norm_stats=True),  # fmt: skip
    # advantage-weighted behaviour cloning (config.py:1327-1374)
    _kai0_task("pi05_flatten_fold_awbc", LerobotAgilexDataConfig, "FlattenFold", "Flatten and fold the cloth.", awbc=True),
    _kai0_task("pi05_tee_shirt_sort_awbc", LerobotAgilexDataConfig, "TeeShirtSort",
               "Fetch the clothes, fold the tee shirts and hand-over the collared shirts.", awbc=True),
    _kai0_task("pi05_hang_cloth_awbc", LerobotARXDataConfig, "HangCloth", "Fetch and hang the cloth.", awbc=True),
]

if len({c.name for c in _CONFIGS}) != len(_CONFIGS):
    raise ValueError("Config names must be unique.")
_CONFIGS_DICT = {c.name: c for c in _CONFIGS}


def get_config(config_name: str) -> TrainConfig:
    """config.py:1402-1409."""
    if config_name not in _CONFIGS_DICT:
        closest = difflib.get_close_matches(config_name, _CONFIGS_DICT.keys(), n=1, cutoff=0.0)
        closest_str = f" Did you mean '{closest[0]}'? " if closest else ""
        raise ValueError(f"Config '{config_name}' not found.{closest_str}")
    return _CONFIGS_DICT[config_name]


def cli(argv: Sequence[str] | None = None) -> TrainConfig:
    """`<config-name> [--field value ...]` -> TrainConfig (config.py:1398-1399; tyro's overridable_config_cli reduced to the
    scalar top-level fields, which is what the training scripts pass: --exp_name, --batch_size, --overwrite, --resume, ...)."""
    ap = argparse.ArgumentParser(description="kai0_amd training configuration")
    sub = ap.add_subparsers(dest="_name", required=True)
    scalar = {}
    for f in dataclasses.fields(TrainConfig):
        if f.name != "name" and f.type in ("str", "int", "bool", "float", "str | None", "int | None", "float | None"):
            scalar[f.name] = f.type
    for name in _CONFIGS_DICT:
        sp = sub.add_parser(name)
        for fname, ftype in scalar.items():
            if ftype == "bool":
                sp.add_argument(f"--{fname}", f"--{fname.replace('_', '-')}", dest=fname, action=argparse.BooleanOptionalAction, default=None)
            else:
                conv = {"int": int, "float": float}.get(ftype.split(" ")[0], str)
                sp.add_argument(f"--{fname}", f"--{fname.replace('_', '-')}", dest=fname, type=conv, default=None)
    ns = vars(ap.parse_args(argv))
    base = _CONFIGS_DICT[ns.pop("_name")]
    return dataclasses.replace(base, **{k: v for k, v in ns.items() if v is not None})
