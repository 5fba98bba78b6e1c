"""Training step driver mirroring scripts/train_pytorch.py:309-633 (the hot loop, not its CLI/wandb shell).

One process per GPU; `Trainer.train_step(observation, actions)` = LR schedule -> forward (flow-matching loss; each unit waits
for its own parameter all-gather) -> backward (gradient reduce-scatter per bucket, issued from inside backward) -> global-norm
clip -> sharded fused AdamW -> parameter all-gathers started (zero2) / shards kept (fsdp).  Checkpoints follow the reference
layout (train_pytorch.py:149-194): `model.safetensors` + ONE world-size independent `optimizer.pt` + `metadata.pt`, written to
`tmp_<step>` then renamed atomically; a checkpoint resumes on any number of GPUs.

Weights must be in the model BEFORE the Trainer is built (it moves every parameter into flat buffers and cuts its f32 master
copies from them); weights written in place afterwards (`load_state_dict`, `p.data.copy_`, model_arithmetic) are adopted
with `Trainer.sync_weights()`.  Rebinding `p.data` afterwards (dtype casts, `.to(device)`) is an error the next step reports.
"""

from __future__ import annotations

import os
import shutil

import torch

from .optim import lr_schedule
from .sharded import ShardedDataParallel


class Trainer:
    def __init__(self, model, *, world_size: int = 1, rank: int = 0, group=None, peak_lr=2.5e-5, warmup_steps=1000,
                 decay_steps=30000, end_lr=2.5e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-10, clip_norm=1.0,
                 shard_ops=None, bucket_bytes: int = 512 << 20, mode: str | None = None, prefetch: int = 1):  # fmt: skip
        self.model = model
        self.world, self.rank = world_size, rank
        self.sched = dict(warmup_steps=warmup_steps, peak_lr=peak_lr, decay_steps=decay_steps, end_lr=end_lr)
        # `gemma_expert.lm_head` is never used by the forward: it can never receive a gradient,
        # so it is left out of the optimizer (torch.optim.AdamW skips grad-less parameters as well).
        dead = model.paligemma_with_expert.gemma_expert.lm_head.weight
        self._param_order = [n for n, _ in model.named_parameters()]  # torch's optimizer indexing (ties once)
        named = [(n, p) for n, p in model.named_parameters() if p is not dead]
        units = model.sharding_units() if hasattr(model, "sharding_units") else None
        mode = mode or os.environ.get("KAI0_SHARD_MODE", "zero2")
        self.engine = ShardedDataParallel(named, world_size=world_size, rank=rank, group=group, ops=shard_ops, betas=betas,
                                          eps=eps, weight_decay=weight_decay, max_grad_norm=clip_norm,
                                          bucket_bytes=bucket_bytes, units=units, mode=mode, prefetch=prefetch)  # fmt: skip
        # the model announces its units (pre_forward / post_forward): parameter gathers are awaited layer by layer
        self._hooked = hasattr(model, "set_unit_hooks")
        if self._hooked:
            model.set_unit_hooks(self.engine)
        self.global_step = 0
        self.last_grad_norm = None

    def lr(self) -> float:
        return lr_schedule(self.global_step, **self.sched)

    def train_step(self, observation, actions, noise=None, time=None) -> torch.Tensor:
        """Returns the (local) mean loss as a 0-d device tensor — no host sync inside the step."""
        lr = self.lr()
        if getattr(self.model, "_engine", None) is not None:
            self.model.invalidate_inference_engine()  # the step below rewrites the weights through raw pointers
        if not self._hooked:
            self.engine.wait_params()
        losses = self.model(observation, actions, noise=noise, time=time)
        loss = losses.mean()
        # gradients are SUMMED across ranks: the 1/N of the global mean goes into the loss
        (loss / self.world if self.world > 1 else loss).backward()
        self.last_grad_norm = self.engine.step(lr)
        self.global_step += 1
        return loss.detach()

    def sync_weights(self):
        """Adopt weights written into the model in place after construction (see the module docstring)."""
        self.engine.sync_master_from_params()
        if getattr(self.model, "_engine", None) is not None:
            self.model.invalidate_inference_engine()

    def params_ready(self):
        """All parameters complete on this GPU (before sample_actions / a state_dict() read between steps)."""
        self.engine.wait_params()

    # ---------------------------------------------------------------------------------------- checkpoints
    def save_checkpoint(self, checkpoint_dir: str) -> str:
        """train_pytorch.py:149-194: rank 0 writes model.safetensors, optimizer.pt (gathered, world-size independent) and
        metadata.pt; every rank takes part in the gathers."""
        from .checkpoint import save_model_safetensors

        step = self.global_step
        final = os.path.join(checkpoint_dir, f"{step}")
        tmp = os.path.join(checkpoint_dir, f"tmp_{step}")
        self.engine.wait_params()
        opt_sd = self.engine.state_dict(self._param_order)
        if self.rank == 0:
            if os.path.exists(tmp):
                shutil.rmtree(tmp)
            os.makedirs(tmp, exist_ok=True)
            save_model_safetensors(self.model, os.path.join(tmp, "model.safetensors"))
            torch.save(opt_sd, os.path.join(tmp, "optimizer.pt"))
            torch.save({"global_step": step, "world_size": self.world}, os.path.join(tmp, "metadata.pt"))
            if os.path.exists(final):
                shutil.rmtree(final)
            os.rename(tmp, final)
        if self.engine.mode == "fsdp":
            self.engine.release_params()
        if self.world > 1:
            torch.distributed.barrier()
        return final

    def load_checkpoint(self, checkpoint_dir: str) -> int:
        """Resume from the highest numeric step directory (train_pytorch.py:197-259); any world size."""
        from .checkpoint import load_model_safetensors

        steps = [int(d) for d in os.listdir(checkpoint_dir) if d.isdigit()]
        if not steps:
            raise FileNotFoundError(f"No checkpoints found in {checkpoint_dir}")
        step = max(steps)
        d = os.path.join(checkpoint_dir, str(step))
        self.engine.wait_params()
        load_model_safetensors(self.model, os.path.join(d, "model.safetensors"))  # in place: the flat buffers keep their views
        opt = torch.load(os.path.join(d, "optimizer.pt"), map_location="cpu", weights_only=False)
        self.engine.load_state_dict(opt, self._param_order)
        self.global_step = torch.load(os.path.join(d, "metadata.pt"), weights_only=False)["global_step"]
        if getattr(self.model, "_engine", None) is not None:
            self.model.invalidate_inference_engine()
        return self.global_step
