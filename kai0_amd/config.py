"""Model configuration with openpi's field names (src/openpi/models/pi0_config.py:19-40,
src/openpi/models/gemma.py:43-109) so callers written against `Pi0Config` / `gemma.get_config` keep working."""

from __future__ import annotations

import dataclasses
from typing import Literal

PALIGEMMA_VOCAB_SIZE = 257_152  # models/gemma.py:40

Variant = Literal["dummy", "gemma_300m", "gemma_2b"]


@dataclasses.dataclass
class GemmaConfig:
    """`openpi.models.gemma.Config` (models/gemma.py:43-52). LoRA variants are out of scope."""

    width: int
    depth: int
    mlp_dim: int
    num_heads: int
    num_kv_heads: int
    head_dim: int


def get_config(variant: str) -> GemmaConfig:
    """`openpi.models.gemma.get_config` (models/gemma.py:58-109)."""
    if variant == "dummy":
        return GemmaConfig(width=64, depth=4, mlp_dim=128, num_heads=8, num_kv_heads=1, head_dim=16)
    if variant == "gemma_300m":
        return GemmaConfig(width=1024, depth=18, mlp_dim=4096, num_heads=8, num_kv_heads=1, head_dim=256)
    if variant == "gemma_2b":
        return GemmaConfig(width=2048, depth=18, mlp_dim=16_384, num_heads=8, num_kv_heads=1, head_dim=256)
    if variant.endswith("_lora"):
        raise NotImplementedError(f"LoRA variant {variant!r} is outside the pi0.5 full fine-tune hot path")
    raise ValueError(f"Unknown variant: {variant}")


@dataclasses.dataclass
class SiglipConfig:
    """SigLIP So400m/14 as the reference's PyTorch path always builds it (gemma_pytorch.py:24-41; HF
    SiglipVisionConfig defaults; models/siglip.py:318-363).  Only tests shrink it."""

    hidden_size: int = 1152
    num_layers: int = 27
    num_heads: int = 16
    intermediate_size: int = 4304
    patch_size: int = 14
    image_size: int = 224
    projection_dim: int = 2048
    layer_norm_eps: float = 1e-6


@dataclasses.dataclass
class Pi0Config:
    """`openpi.models.pi0_config.Pi0Config` fields (pi0_config.py:19-40). Only pi05=True is implemented."""

    dtype: str = "bfloat16"
    paligemma_variant: str = "gemma_2b"
    action_expert_variant: str = "gemma_300m"
    action_dim: int = 32
    action_horizon: int = 50
    max_token_len: int | None = None
    pi05: bool = True
    discrete_state_input: bool | None = None
    # not in the reference: lets tests build a small vision tower / vocabulary
    siglip: SiglipConfig = dataclasses.field(default_factory=SiglipConfig)
    vocab_size: int = PALIGEMMA_VOCAB_SIZE

    def __post_init__(self):
        if self.max_token_len is None:
            self.max_token_len = 200 if self.pi05 else 48
        if self.discrete_state_input is None:
            self.discrete_state_input = self.pi05

    @property
    def model_type(self) -> str:
        """`ModelType` value (models/model.py:27-35; pi0_config.py:43-47)."""
        return "pi05" if self.pi05 else "pi0"

    def _model_class(self):
        from .model import PI0Pytorch

        return PI0Pytorch

    def load_pytorch(self, train_config, weight_path: str):
        """`BaseModelConfig.load_pytorch` (models/model.py:276-280): build the torch-protocol model from `train_config.model`
        and load `model.safetensors` (strict, tied `lm_head` forgiven as `safetensors.torch.load_model` does).  The weights
        are read before any trainer / inference engine exists, as both require (kai0_amd.sharded / kai0_amd.infer)."""
        import logging

        from .checkpoint import load_model_safetensors

        logging.getLogger("kai0_amd").info(f"train_config: {train_config}")
        model = train_config.model._model_class()(train_config.model)
        load_model_safetensors(model, weight_path)
        return model


@dataclasses.dataclass
class AdvantageEstimatorConfig(Pi0Config):
    """`openpi.models.pi0_config.AdvantageEstimatorConfig` (pi0_config.py:138-142): the two loss weights of the
    Stage-Advantage estimator (training/config.py:1225-1226 trains it with value 1 / action 0)."""

    loss_action_weight: float = 1.0
    loss_value_weight: float = 1.0

    def _model_class(self):
        from .model import AdvantageEstimator

        return AdvantageEstimator
