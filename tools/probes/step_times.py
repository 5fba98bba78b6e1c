"""Per-step wall times of the benchmarked training step (B = 32, synthetic batch resident in HBM), each closed by a device
synchronisation: is the mean of bench.py's timed region made of equal steps, or of fast steps and periodic outliers?
usage: python tools/probes/step_times.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from kai0_amd.config import Pi0Config
from kai0_amd.train import Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0"); cfg = Pi0Config()
model = bench.build_model(cfg, dev, 0); model.train()
tr = Trainer(model, world_size=1, rank=0, peak_lr=2.5e-5, warmup_steps=1000, decay_steps=30000, end_lr=2.5e-6, weight_decay=1e-10, clip_norm=1.0)
obs, actions = bench.synthetic_batch(cfg, 32, seed=1000, device=dev)
for _ in range(3): tr.train_step(obs, actions)
torch.cuda.synchronize()
ts = []
for _ in range(n):
    t0 = time.perf_counter(); tr.train_step(obs, actions); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("synchronised per step:", " ".join(f"{t:.0f}" for t in ts))
s = sorted(ts); print(f"min {s[0]:.1f} p50 {s[len(s)//2]:.1f} max {s[-1]:.1f} mean {sum(ts)/len(ts):.1f}")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): tr.train_step(obs, actions)
torch.cuda.synchronize(); print(f"unsynchronised mean {(time.perf_counter() - t0) * 1e3 / n:.1f} ms/step")
