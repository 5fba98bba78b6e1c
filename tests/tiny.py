"""Shared tiny configurations for parity tests (oracle on CPU vs HIP model on the GPU)."""

import torch


def tiny_cfgs(action_horizon=10, max_token_len=24):
    from kai0_amd.config import Pi0Config, SiglipConfig
    from oracle.pi0_oracle import OracleConfig, SiglipCfg

    sk = dict(hidden_size=64, num_layers=2, num_heads=4, intermediate_size=136, patch_size=14, image_size=56,
              projection_dim=64, layer_norm_eps=1e-6)
    common = dict(dtype="bfloat16", paligemma_variant="dummy", action_expert_variant="dummy", action_dim=32,
                  action_horizon=action_horizon, max_token_len=max_token_len, pi05=True, vocab_size=304)
    return Pi0Config(siglip=SiglipConfig(**sk), **common), OracleConfig(siglip=SiglipCfg(**sk), **common)


def build_pair(device, seed=0, std=None, **kw):
    """(hip_model on `device`, oracle on CPU) with identical synthetic weights."""
    from kai0_amd.model import PI0Pytorch
    from oracle.pi0_oracle import OraclePI0, synthetic_weights_

    pcfg, ocfg = tiny_cfgs(**kw)
    oracle = OraclePI0(ocfg)
    synthetic_weights_(oracle, seed=seed)
    if std is not None:
        with torch.no_grad():
            for n, p in oracle.named_parameters():
                if p.dim() >= 2:
                    p.mul_(std / 0.02)
    model = PI0Pytorch(pcfg)
    missing, unexpected = model.load_state_dict(oracle.state_dict(), strict=True)
    model.train_augmentation = False
    return model.to(device), oracle, pcfg, ocfg


def obs_to(obs, device):
    from oracle.pi0_oracle import SimpleObs

    return SimpleObs(
        images={k: v.to(device) for k, v in obs.images.items()},
        image_masks={k: v.to(device) for k, v in obs.image_masks.items()},
        state=obs.state.to(device), tokenized_prompt=obs.tokenized_prompt.to(device),
        tokenized_prompt_mask=obs.tokenized_prompt_mask.to(device), token_ar_mask=None, token_loss_mask=None,
    )


def estimator_case(E):
    """The AdvantageEstimator case of tests/golden/reference_e2e.safetensors (make_reference_e2e_golden.py): the tiny model's
    weights + a seeded value head, six images (two timesteps x three cameras) inserted out of (timestep, camera) order,
    progress targets (one outside [-1, 1]).  Returns (oracle_estimator, observation, actions, noise, time)."""
    from oracle.pi0_oracle import OracleAdvantageEstimator, OraclePI0, SimpleObs, synthetic_batch, synthetic_weights_

    _, ocfg = tiny_cfgs()
    base = OraclePI0(ocfg)
    synthetic_weights_(base, seed=0)
    with torch.no_grad():
        for n, p in base.named_parameters():
            if p.dim() >= 2:
                p.mul_(0.08 / 0.02)
    est = OracleAdvantageEstimator(ocfg, loss_value_weight=0.7, loss_action_weight=1.3)
    est.load_state_dict({**base.state_dict(), **{k[3:]: v for k, v in E.items() if k.startswith("ae.value_head.")}}, strict=True)
    obs, actions, noise, time = synthetic_batch(ocfg, 2, seed=0)
    extra = {k[7:]: v for k, v in E.items() if k.startswith("ae.img.")}
    images = {"left_wrist_0_rgb": obs.images["left_wrist_0_rgb"], "right_wrist_-1_rgb": extra["right_wrist_-1_rgb"],
              "base_-1_rgb": extra["base_-1_rgb"], "left_wrist_-1_rgb": extra["left_wrist_-1_rgb"],
              "base_0_rgb": obs.images["base_0_rgb"], "right_wrist_0_rgb": obs.images["right_wrist_0_rgb"]}  # fmt: skip
    obs6 = SimpleObs(images=images, image_masks={k: torch.ones(2, dtype=torch.bool) for k in images}, state=obs.state,
                     tokenized_prompt=obs.tokenized_prompt, tokenized_prompt_mask=obs.tokenized_prompt_mask, token_ar_mask=None,
                     token_loss_mask=None, progress=E["ae.progress"])  # fmt: skip
    return est, obs6, actions, noise, time
