"""Generate tests/golden/tiny_pi05.safetensors from the CPU oracle (oracle/pi0_oracle.py).

The reference ships no golden vectors for the model and cannot be imported in this image, so
these fixtures pin the *oracle restatement* (seeded weights + seeded inputs -> loss and action chunk) so that
any later drift of the oracle, or of the HIP path against it, is caught.  Re-run only when the oracle changes
on purpose:  python tests/golden/make_golden.py
"""

import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from tiny import build_pair  # noqa: E402

from oracle.pi0_oracle import synthetic_batch  # noqa: E402


def main():
    torch.manual_seed(0)
    _, oracle, _, ocfg = build_pair("cpu", seed=0, std=0.08)
    obs, actions, noise, time = synthetic_batch(ocfg, 2, seed=0)
    loss = oracle(obs, actions, noise, time)
    chunk = oracle.sample_actions(obs, noise, num_steps=10)
    out = {"loss": loss.detach().contiguous(), "actions": chunk.contiguous(), "noise": noise, "time": time,
           "in_actions": actions, "tokens": obs.tokenized_prompt, "token_mask": obs.tokenized_prompt_mask.to(torch.uint8)}
    save_file(out, os.path.join(HERE, "tiny_pi05.safetensors"))
    print("wrote tiny_pi05.safetensors: loss mean %.6f, |actions| mean %.6f" % (float(loss.detach().mean()), float(chunk.abs().mean())))


if __name__ == "__main__":
    main()
