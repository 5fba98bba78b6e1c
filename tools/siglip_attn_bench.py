"""SigLIP attention forward at the training shape (96 images x 16 heads x 256 tokens x 72): us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 96
S, NH, HD = 256, 16, 72
q, k, v = (torch.randn(n_img * S, NH * HD, device=dev).to(torch.bfloat16) for _ in range(3))
for _ in range(3):
    out = ops.siglip_attention(q, k, v, n_img, S, NH, HD)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.siglip_attention(q, k, v, n_img, S, NH, HD)
e.record()
torch.cuda.synchronize()
print(f"siglip attention fwd n_img={n_img}: {s.elapsed_time(e) / 20 * 1e3:.1f} us; checksum {float(out.float().abs().sum()):.6e}")
