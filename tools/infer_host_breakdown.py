"""Where the host-side time of one B = 1 `sample_actions` call goes (p50 over calls): preprocessing, engine lookup (weight
fingerprint), input copies, graph replay (device time), output clone, D2H.  usage: python tools/infer_host_breakdown.py [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kai0_amd.config import Pi0Config  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
cfg = Pi0Config()
model = bench.build_model(cfg, dev, 0).eval()
obs, _ = bench.synthetic_batch(cfg, 1, seed=123, device=dev)
noise = torch.randn(1, cfg.action_horizon, cfg.action_dim, device=dev)
for _ in range(3):
    model.sample_actions(dev, obs, noise=noise)
torch.cuda.synchronize()
eng = model._engine
rows = {k: [] for k in ("total", "preprocess", "compatible", "copies", "replay", "clone", "d2h")}


def tick():
    torch.cuda.synchronize()
    return time.perf_counter()


with torch.no_grad():
    for _ in range(iters):
        t0 = time.perf_counter()
        model.sample_actions(dev, obs, noise=noise).cpu()
        rows["total"].append(time.perf_counter() - t0)
        t = tick()
        images, img_masks, lang_tokens, lang_masks, state = model._preprocess_observation(obs, train=False)
        t1 = tick(); rows["preprocess"].append(t1 - t)
        eng.compatible(1, lang_tokens.shape[1], len(images))
        t2 = tick(); rows["compatible"].append(t2 - t1)
        si = eng._static_in
        for dst, src in zip(si["images"], images):
            dst.copy_(src)
        for dst, src in zip(si["img_masks"], img_masks):
            dst.copy_(src)
        si["lang_tokens"].copy_(lang_tokens); si["lang_masks"].copy_(lang_masks); si["noise"].copy_(noise)
        t3 = tick(); rows["copies"].append(t3 - t2)
        eng._graph.replay()
        t4 = tick(); rows["replay"].append(t4 - t3)
        out = eng._static_out.clone()
        t5 = tick(); rows["clone"].append(t5 - t4)
        out.cpu()
        t6 = tick(); rows["d2h"].append(t6 - t5)
for k, v in rows.items():
    v.sort()
    print(f"{k:12s} p50 {v[len(v) // 2] * 1e3:8.3f} ms   min {v[0] * 1e3:8.3f}")
