"""B=1 action-chunk inference a few times (for rocprofv3 / latency breakdown). usage: infer_once.py [iters] [graph 0|1]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kai0_amd.config import Pi0Config  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
os.environ["KAI0_INFER_GRAPH"] = sys.argv[2] if len(sys.argv) > 2 else "1"
dev = torch.device("cuda:0")
cfg = Pi0Config()
model = bench.build_model(cfg, dev, 0).eval()
obs, _ = bench.synthetic_batch(cfg, 1, seed=123, device=dev)
noise = torch.randn(1, cfg.action_horizon, cfg.action_dim, device=dev)
for _ in range(2):
    model.sample_actions(dev, obs, noise=noise)
torch.cuda.synchronize()
ts = []
for _ in range(iters):
    t0 = time.perf_counter()
    model.sample_actions(dev, obs, noise=noise).cpu()
    ts.append((time.perf_counter() - t0) * 1e3)
print("chunk ms:", [round(t, 2) for t in ts])
