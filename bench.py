"""bench.py — pi0.5 full fine-tune throughput (train samples/s) on N MI355X, plus p50 action-chunk latency.

Contract (see the round prompt): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it
under torch.distributed.run, one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): pi0.5 full fine-tune, bf16 compute, batch 32 per GPU, 3 cameras 224x224 +
200 prompt tokens + 50x32 action chunk, synthetic data resident in HBM, random-init weights of the real
architecture (3.617 B stored / 3.353 B used parameters).  A step = augmentation + forward + backward + global-norm
clip + fused AdamW over every parameter: nothing is skipped inside the timed region.

Extra objects on the same line:
  roofline     — the dominant kernel (gemm_bf16_kernel): achieved = algorithmic FLOPs (14.04 TFLOP per
                 sample per train step) / summed GEMM launch durations, measured with HIP events on the launch stream
                 inside the timed region; peak = 2.5 PFLOP/s dense bf16 MFMA.
  cpu_baseline — the CPU oracle (a port: the reference cannot be imported here) timed on the host cores on a bounded
                 sample of the same workload (see `sample`).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRAIN_TFLOP_PER_SAMPLE = 14.04  # 3 x 4.68 TFLOP forward, no remat, no unused lm_head
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def synthetic_batch(cfg, batch: int, seed: int, device):
    """SURVEY.md §8d synthetic inputs, generated on the device."""
    from kai0_amd.preprocessing import IMAGE_KEYS, Observation

    g = torch.Generator(device=device).manual_seed(seed)
    hw = cfg.siglip.image_size
    images, masks = {}, {}
    for k in IMAGE_KEYS:
        u8 = torch.randint(0, 256, (batch, hw, hw, 3), generator=g, device=device, dtype=torch.uint8)
        images[k] = (u8.to(torch.float32) / 255.0 * 2.0 - 1.0).permute(0, 3, 1, 2).contiguous()
        masks[k] = torch.ones(batch, dtype=torch.bool, device=device)
    L = cfg.max_token_len
    tokens = torch.randint(0, 2048, (batch, L), generator=g, device=device, dtype=torch.int64)
    n_valid = 64 + torch.randint(0, 65, (batch,), generator=g, device=device)
    tmask = torch.arange(L, device=device)[None, :] < n_valid[:, None]
    state = torch.zeros(batch, cfg.action_dim, device=device)
    state[:, :14] = torch.rand(batch, 14, generator=g, device=device) * 2 - 1
    actions = torch.zeros(batch, cfg.action_horizon, cfg.action_dim, device=device)
    actions[..., :14] = torch.randn(batch, cfg.action_horizon, 14, generator=g, device=device)
    obs = Observation(images=images, image_masks=masks, state=state, tokenized_prompt=tokens, tokenized_prompt_mask=tmask)
    return obs, actions


def build_model(cfg, device, seed: int):
    from kai0_amd.model import PI0Pytorch

    torch.manual_seed(seed)
    with torch.device(device):
        model = PI0Pytorch(cfg)
    # explicit weights — plain RMSNorm weights 0 (checkpoint-like), adaRMS dense non-trivial
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "dense.weight" in name:
                p.normal_(0.0, 0.02)
    return model


class GemmTimer:
    """Brackets every kai0_gemm_bf16 launch with HIP events on the launch stream (torch's current stream)."""

    def __init__(self):
        self.events = []

    def install(self):
        from kai0_amd import ops

        self._orig = ops.gemm
        timer = self

        def timed_gemm(A, B, out, **kw):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = timer._orig(A, B, out, **kw)
            e.record()
            timer.events.append((s, e, 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("batch", 1),
                                 (int(kw.get("a_kc", True)), int(kw.get("b_kc", True)), kw["M"], kw["N"], kw["K"], kw.get("batch", 1))))
            return r

        ops.gemm = timed_gemm

    def uninstall(self):
        from kai0_amd import ops

        ops.gemm = self._orig

    def summarize(self):
        ms = sum(s.elapsed_time(e) for s, e, _, _ in self.events)
        fl = sum(f for _, _, f, _ in self.events)
        return ms, fl, len(self.events)

    def breakdown(self):
        agg = {}
        for s, e, f, key in self.events:
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return [{"a_kc": k[0], "b_kc": k[1], "M": k[2], "N": k[3], "K": k[4], "batch": k[5], "calls": v[0], "ms": v[1],
                 "tflops": v[2] / v[1] / 1e9} for k, v in rows]


def cpu_baseline(budget_s: float = 20.0):
    """The CPU oracle (a port — the reference cannot be imported here) on the host cores: one sample, forward +
    backward through a full-width slice of the network (2 of 27 SigLIP layers x 3 cameras, 1 of 18 joint
    Gemma-2B/expert layers, vocab 2048), fp32 (bf16 matmuls are emulated on hosts without AMX and would understate the
    CPU); the time is scaled to the full depth by FLOP share.  The full 3.6 B-parameter oracle needs minutes just to
    initialise on CPU, which would not fit the default run."""
    import copy

    from oracle import pi0_oracle as O

    n_sig, n_joint = 2, 1
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    vlm, exp = copy.copy(O.get_gemma_config("gemma_2b")), copy.copy(O.get_gemma_config("gemma_300m"))
    vlm.depth = exp.depth = n_joint
    orig = O.get_gemma_config
    O.get_gemma_config = lambda v: vlm if v == "gemma_2b" else exp
    try:
        cfg = O.OracleConfig(dtype="float32", vocab_size=2048, siglip=O.SiglipCfg(num_layers=n_sig))
        model = O.OraclePI0(cfg)
    finally:
        O.get_gemma_config = orig
    O.synthetic_weights_(model, seed=0)
    obs, actions, noise, t = O.synthetic_batch(cfg, 1, seed=0)
    times = []
    t_start = time.time()
    while len(times) < 4 and (not times or time.time() - t_start + times[-1] < budget_s):
        model.zero_grad(set_to_none=True)
        t0 = time.time()
        model(obs, actions, noise, t).mean().backward()
        times.append(time.time() - t0)
    per_slice = min(times)
    # FLOP share of the slice (per-sample forward TFLOP): SigLIP 0.661 over 27 layers (3 cameras included),
    # joint layers (3.837 + 0.153 + 0.031) over 18
    slice_tf = 0.661 * n_sig / 27 + (3.837 + 0.153 + 0.031) * n_joint / 18
    full_tf = 4.68
    est_step_s = per_slice * full_tf / slice_tf
    return {
        "value": 1.0 / est_step_s,
        "unit": "samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"1 sample fwd+bwd (no optimizer), fp32 oracle, full-width slice: {n_sig}/27 SigLIP x3 cams + "
        f"{n_joint}/18 joint layers = {per_slice:.2f} s (best of {len(times)}), scaled by FLOP share "
        f"{slice_tf / full_tf:.3f} to {est_step_s:.1f} s/sample",
    }


def measure_latency(model, cfg, device, iters: int = 30):
    """p50 of Policy-level model time for one action chunk at B=1 (prefix pass + 10 denoise steps, hipGraph)."""
    model.eval()
    obs, _ = synthetic_batch(cfg, 1, seed=123, device=device)
    noise = torch.randn(1, cfg.action_horizon, cfg.action_dim, device=device)
    for _ in range(3):
        model.sample_actions(device, obs, noise=noise, num_steps=10)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        out = model.sample_actions(device, obs, noise=noise, num_steps=10)
        out.cpu()  # Policy.infer moves the chunk to the host (policy.py:114-118)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    model.train()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-gemm-timing", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist

    from kai0_amd.config import Pi0Config
    from kai0_amd.train import Trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("KAI0_FORCE_COLLECTIVES") == "1":  # the latter: 1-rank RCCL dry run of the sharded path
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # NCCL_DEBUG=VERSION (set on the bench boxes) makes RCCL print a banner through C stdio, which a pipe flushes at
        # process exit — i.e. AFTER the JSON line this script owes its caller as the last line of stdout.  Banner off;
        # anything else a user asked for (WARN, INFO, TRACE) stays, and every rank flushes C stdio before rank 0 prints.
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        dist.init_process_group(backend="nccl", init_method="env://", device_id=device)

    cfg = Pi0Config()
    B = args.batch_per_gpu
    model = build_model(cfg, device, seed=0)  # same weights on every rank
    model.train()
    trainer = Trainer(model, world_size=world, rank=rank, peak_lr=2.5e-5, warmup_steps=1000, decay_steps=30000,
                      end_lr=2.5e-6, weight_decay=1e-10, clip_norm=1.0)  # fmt: skip
    obs, actions = synthetic_batch(cfg, B, seed=1000 + rank, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(obs, actions)
    # Timed region: exactly `steps` training steps between barriers, nothing else on the stream.
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(obs, actions)
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline of the dominant kernel: HIP events around every bf16 GEMM launch, over `timer_steps` further identical steps
    # right after the timed region (every rank steps, rank 0 measures).  Inside the timed region the 2 x 1417 event
    # records per step cost ~10 ms (1.7 %) and, at N > 1, would make rank 0 the straggler the max-over-ranks reports.
    timer = None
    timer_steps = 2
    if not args.no_gemm_timing:
        if rank == 0:
            timer = GemmTimer()
            timer.install()
        for _ in range(timer_steps):
            trainer.train_step(obs, actions)
        barrier()
        if timer is not None:
            timer.uninstall()
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el)
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    if world > 1:  # every rank empties its C stdio buffers (RCCL warnings) now, so nothing of theirs can follow rank 0's line
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "train samples/sec pi0.5 full FT",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic (3x224x224 RGB + 200 prompt tokens + 50x32 actions, resident in HBM; random-init weights)",
            "config": {
                "workload": "pi0.5 full fine-tune bf16, batch 32 per MI355X, 3-cam 224x224 (BASELINE.json configs[1])",
                "global_batch": B * world,
                "seq_len": 968 + 50,
                "parallelism": f"dp{world}" + ("" if world == 1 else " (sharded optimizer/grads, RCCL reduce-scatter + all-gather)"),
                "params_stored": 3.617e9,
                "final_loss": float(loss),
            },
        }
        if timer is not None:
            gemm_ms, gemm_flops, n_launch = timer.summarize()
            # algorithmic flops of a launch = 2 M N K of that launch (no tile padding counted); summed over the launches
            # of the measured steps and divided by their summed HIP-event durations
            achieved = gemm_flops / 1e12 / (gemm_ms / 1e3)
            # PMC counters cannot be read from inside this process: the per-launch HBM-side bytes come from the committed
            # rocprofv3 --pmc passes over this same command (tools/collect_profiles.sh -> profiles/gemm_traffic.json)
            traffic = {}
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "gemm_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath))
            out["roofline"] = {
                "bound": "mfma",
                "kernel": "gemm_bf16_kernel (NT/NN/TN variants; every Linear, attention matmul, dgrad and wgrad)",
                "achieved": achieved,
                "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                "traffic": traffic.get("bytes_per_launch"),
                "traffic_unit": "HBM-side bytes per GEMM launch (fetch + write), rocprofv3 PMC; not re-measured by this run",
                "traffic_source": traffic.get("source"),
                "launches_per_step": n_launch // timer_steps,
                "avg_launch_ms": gemm_ms / n_launch,
                "gemm_ms_per_step": gemm_ms / timer_steps,
                "algorithmic_tflop_per_step": TRAIN_TFLOP_PER_SAMPLE * B,
                "gemm_tflop_per_step": gemm_flops / timer_steps / 1e12,
                "timed": f"HIP events over {timer_steps} further identical steps right after the timed region",
                "step_frac_of_mfma_peak": TRAIN_TFLOP_PER_SAMPLE * B / (ms_per_step / 1e3) / MFMA_BF16_PEAK_TFLOPS,
            }
            if os.environ.get("KAI0_GEMM_BREAKDOWN"):
                os.makedirs("gpurun_out", exist_ok=True)
                json.dump(timer.breakdown(), open("gpurun_out/gemm_breakdown.json", "w"), indent=0)
        if world == 1 and not args.no_latency:
            del trainer
            torch.cuda.empty_cache()
            out["p50_action_chunk_ms"] = measure_latency(model, cfg, device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        try:  # whatever native libraries still hold in C stdio buffers goes out first: the JSON line must be the last one
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
