"""Joint prefix-LM MQA attention forward at the training shape (B = 32, S = 1018, 8 heads of 256): ms per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, S, H, HD = 32, 1018, 8, 256
S_ld = 1024
q = torch.randn(B, S_ld, H * HD, device=dev).to(torch.bfloat16)
k = torch.randn(B, S_ld, HD, device=dev).to(torch.bfloat16)
v = torch.randn(B, S_ld, HD, device=dev).to(torch.bfloat16)
code = torch.zeros(B, S, dtype=torch.int32, device=dev)
code[:, 968:] = 1
qcode, kcode = code.clone(), code.clone()
kcode[:, 900:968] = 2**31 - 1  # padded prompt tokens
qcode[:, 900:968] = -1


def run():
    return ops.mqa_attention_fwd(q, k, v, qcode, kcode, B, S, 0, S, S_ld, H, HD, HD**-0.5)


for _ in range(3):
    att, probs = run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"attention fwd: {ms:.3f} ms  {4 * B * H * S * S * HD / ms / 1e9:.1f} TF/s algorithmic; checksum {float(att.float().abs().sum()):.6e} {float(probs.float().sum()):.6e}")
