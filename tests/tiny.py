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
