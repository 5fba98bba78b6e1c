"""Serve-path latency at full size: `Policy.infer` on a raw Agilex observation (three 480x640 uint8 cameras,
14-DoF state, prompt) through the HIP model with random weights — wall time per request vs the model time the policy reports
(`policy_timing.infer_ms`), i.e. how much of a request the numpy / PIL / tokeniser / H2D host pipeline costs.
usage: python tools/policy_latency.py [requests]   (writes gpurun_out/policy_latency.json)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kai0_amd import agilex_policy, policy, tokenizer  # noqa: E402
from kai0_amd.config import Pi0Config  # noqa: E402
from kai0_amd.normalize import NormStats  # noqa: E402

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
cfg = Pi0Config()
model = bench.build_model(cfg, dev, 0).eval()
G = np.load(os.path.join(ROOT, "tests", "golden", "host_pipeline.npz"))
tok = tokenizer.PaligemmaTokenizer(max_len=cfg.max_token_len, model=G["tok.model"].tobytes())  # tiny vocabulary, real code path
rng = np.random.default_rng(0)
stats = {k: NormStats(mean=np.zeros(32), std=np.ones(32), q01=-np.ones(32) * 2, q99=np.ones(32) * 2) for k in ("state", "actions")}
pol = policy.create_policy(model, norm_stats=stats, tokenizer=tok, action_dim=32, use_quantile_norm=True, image_size=224,
                           robot_inputs=[agilex_policy.AgilexInputs(action_dim=32, model_type="pi05")],
                           robot_outputs=[agilex_policy.AgilexOutputs()], sample_kwargs={"num_steps": 10}, pytorch_device="cuda:0")  # fmt: skip


def request():
    return {"images": {k: rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for k in ("top_head", "hand_left", "hand_right")},
            "state": rng.uniform(-1, 1, size=14), "prompt": "Flatten and fold the cloth."}  # fmt: skip


def run(trim: bool):
    model.trim_prompt_padding_infer = trim  # (what policy.create_trained_policy sets for the serve path)
    for _ in range(3):
        pol.infer(request())
    wall, modelms, stages = [], [], []
    for _ in range(n_req):
        req = request()
        t0 = time.perf_counter()
        pol._input_transform(dict(req))
        t1 = time.perf_counter()
        stages.append((t1 - t0) * 1e3)
        t0 = time.perf_counter()
        out = pol.infer(req)
        wall.append((time.perf_counter() - t0) * 1e3)
        modelms.append(out["policy_timing"]["infer_ms"])
    wall.sort(), modelms.sort(), stages.sort()
    return {"wall_p50_ms": wall[n_req // 2], "wall_p90_ms": wall[int(n_req * 0.9)], "wall_max_ms": wall[-1],  # (the tail: a p50 hides periodic stalls)
            "model_p50_ms": modelms[n_req // 2], "host_p50_ms": wall[n_req // 2] - modelms[n_req // 2],
            "input_transform_p50_ms": stages[n_req // 2], "prompt_slots": int(model._engine.T)}  # fmt: skip


res = {"requests": n_req, **run(False), "trimmed_prompt": run(True),
       "note": "three 480x640 uint8 cameras -> tokenise -> H2D (raw frames) -> resize_with_pad 224 on the device (kai0_amd.device_resize, Pillow-bit-identical) -> sample_actions (graph) -> D2H -> unnormalise; "
               "top level: all max_token_len prompt slots computed; trimmed_prompt: the serve default (slots the prompt does not fill dropped)"}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "policy_latency.json"), "w"), indent=1)
print(json.dumps(res))
