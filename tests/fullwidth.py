"""BASELINE.json's configuration at FULL WIDTH with reduced depth: Gemma-2B (2048 / 16384 / 8 x 256) + Gemma-300M expert
(1024 / 4096) joint layers, SigLIP so400m/14 layers (1152 / 4304 / 16 heads), three 224^2 cameras, 200 prompt tokens,
50 x 32 actions — S = 1018, P = 968 — with `n_joint` of the 18 joint layers, `n_sig` of the 27 SigLIP layers and a
2048-entry vocabulary (the synthetic prompts only draw ids below 2048).  Every kernel runs in the tile / split / schedule
configuration of the real model (those depend on the widths and the sequence, not on the depth), while the CPU oracle
(fp32 and bf16 choreography) still finishes in seconds, so the two can be compared number by number."""

import contextlib
import copy

import torch


@contextlib.contextmanager
def _depth(n_joint: int):
    """Temporarily make `gemma_2b` / `gemma_300m` resolve to `n_joint`-layer configs, for the oracle and for the HIP model."""
    import kai0_amd.config as kc
    import kai0_amd.model as km
    from oracle import pi0_oracle as O

    o_orig, k_orig = O.get_gemma_config, km.get_config

    def o_get(v):
        c = copy.copy(o_orig(v))
        c.depth = n_joint
        return c

    def k_get(v):
        c = copy.copy(kc.get_config(v))
        c.depth = n_joint
        return c

    O.get_gemma_config, km.get_config = o_get, k_get
    try:
        yield
    finally:
        O.get_gemma_config, km.get_config = o_orig, k_orig


def build_oracle(n_joint: int = 2, n_sig: int = 2, dtype: str = "bfloat16", seed: int = 0):
    from oracle import pi0_oracle as O

    with _depth(n_joint):
        cfg = O.OracleConfig(dtype=dtype, vocab_size=2048, siglip=O.SiglipCfg(num_layers=n_sig))
        oracle = O.OraclePI0(cfg)
    O.synthetic_weights_(oracle, seed=seed)
    return oracle, cfg


def build_hip(oracle, n_joint: int = 2, n_sig: int = 2, device="cuda:0"):
    """The HIP model with the oracle's weights (state dict moved over unchanged)."""
    from kai0_amd.config import Pi0Config, SiglipConfig
    from kai0_amd.model import PI0Pytorch

    with _depth(n_joint):
        model = PI0Pytorch(Pi0Config(vocab_size=2048, siglip=SiglipConfig(num_layers=n_sig)))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.train_augmentation = False
    return model.to(device)
