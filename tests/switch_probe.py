"""Run by tests/test_switches_gpu.py in a subprocess (the library reads its KAI0_* switches once per process): the full-width,
one-joint-layer / one-SigLIP-layer pi0.5 (tests/fullwidth.py: every kernel in the launch configuration of the real model) with
seeded weights and inputs — training loss, three parameter gradients and the 10-step B = 1 action chunk — saved to argv[1]."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main(out_path: str) -> None:
    from fullwidth import build_hip, build_oracle
    from test_fullsize_gpu import _take
    from tiny import obs_to

    from oracle.pi0_oracle import synthetic_batch

    dev = torch.device("cuda:0")
    oracle, ocfg = build_oracle(1, 1)  # (the oracle only provides the seeded weights and the synthetic batch here)
    model = build_hip(oracle, 1, 1, dev)
    obs, actions, noise, time = synthetic_batch(ocfg, 2, seed=3)
    gobs = obs_to(obs, dev)
    model.train()
    loss = model(gobs, actions.to(dev), noise=noise.to(dev), time=time.to(dev))
    loss.mean().backward()
    pe = model.paligemma_with_expert
    grads = {
        "prefix.q_proj": pe.paligemma.model.language_model.layers[0].self_attn.q_proj.weight.grad,
        "expert.down_proj": pe.gemma_expert.model.layers[0].mlp.down_proj.weight.grad,
        "siglip.fc1": pe.paligemma.model.vision_tower.vision_model.encoder.layers[0].mlp.fc1.weight.grad,
        "action_in_proj": model.action_in_proj.weight.grad,
    }
    model.eval()
    chunk = model.sample_actions(dev, _take(gobs, 1), noise=noise[1:2].to(dev), num_steps=10)
    torch.cuda.synchronize()
    torch.save({"loss": loss.detach().float().cpu(), "chunk": chunk.float().cpu(),
                **{"grad." + k: v.detach().float().cpu() for k, v in grads.items()}}, out_path)  # fmt: skip


if __name__ == "__main__":
    main(sys.argv[1])
