"""Fused AdamW over one 256 Mi-element bf16 bucket (f32 master / moments: 3 GiB of state + 0.5 GiB grad + 0.5 GiB parameters):
ms per launch and HBM bytes per second (28 B per element).  Round 3: a four-elements-per-thread version (16-B accesses to the state
streams) measured 1.606 ms against 1.562 ms for the one-element loop that is in the library: the update is bound by the mixed
read / write HBM stream (4.8 TB/s), not by its instruction count."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kai0_amd.optim import adamw_step_  # noqa: E402

dev = torch.device("cuda:0")
n = 256 << 20
master = torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
grad = torch.randn(n, device=dev).bfloat16()
param = master.bfloat16()
coef = torch.ones(1, device=dev)
for _ in range(3):
    adamw_step_(master, m, v, grad, param, lr=1e-4, beta1=0.9, beta2=0.95, eps=1e-8, wd=1e-10, step=1, clip_coef=coef)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(10):
    adamw_step_(master, m, v, grad, param, lr=1e-4, beta1=0.9, beta2=0.95, eps=1e-8, wd=1e-10, step=2 + i, clip_coef=coef)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"adamw: {ms:.3f} ms per {n >> 20} Mi elements = {28 * n / ms / 1e6:.0f} GB/s")
