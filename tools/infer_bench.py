"""p50 of one B = 1 action chunk + stage split, without the training bench (quick iteration on the inference path).
usage: python tools/infer_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kai0_amd.config import Pi0Config  # noqa: E402

dev = torch.device("cuda:0")
cfg = Pi0Config()
model = bench.build_model(cfg, dev, 0)
res = bench.measure_latency(model, cfg, dev, iters=40)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
