"""One-pass attention forward at the training shape: ms per call (KAI0_ATTN_ABLATE bit mask removes phases; timing only).
Needs a library built with KAI0_HIPCC_FLAGS=-DKAI0_ABLATE (the shipped kernels carry no ablation branch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import ops
dev = torch.device("cuda:0")
B, S, H, HD, S_ld = 32, 1018, 8, 256, 1024
q = torch.randn(B, S_ld, H * HD, device=dev).to(torch.bfloat16)
k = torch.randn(B, S_ld, HD, device=dev).to(torch.bfloat16)
v = torch.randn(B, S_ld, HD, device=dev).to(torch.bfloat16)
code = torch.zeros(B, S, dtype=torch.int32, device=dev); code[:, 968:] = 1
qcode, kcode = code.clone(), code.clone()
kcode[:, 900:968] = 2**31 - 1; qcode[:, 900:968] = -1
run = lambda: ops.mqa_attention_fwd(q, k, v, qcode, kcode, B, S, 0, S, S_ld, H, HD, HD**-0.5, want_lse=True)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print(f"ablate={os.environ.get('KAI0_ATTN_ABLATE','0'):>3s} one-pass attention fwd: {s.elapsed_time(e) / 10:.3f} ms")
