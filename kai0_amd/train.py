"""Training step driver mirroring scripts/train_pytorch.py:309-633 (the hot loop, not its CLI/wandb shell).

One process per GPU; `Trainer.train_step(observation, actions)` = LR schedule -> forward (flow-matching loss) ->
backward (gradient reduce-scatter overlapped through autograd hooks) -> global-norm clip -> sharded fused AdamW
-> parameter all-gather.  Checkpoints follow the reference layout (train_pytorch.py:149-194): `model.safetensors`
(+ optimizer shards + metadata) written to `tmp_<step>` then renamed atomically.
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
                 shard_ops=None, bucket_bytes: int = 512 << 20):  # fmt: skip
        self.model = model
        self.world, self.rank = world_size, rank
        self.sched = dict(warmup_steps=warmup_steps, peak_lr=peak_lr, decay_steps=decay_steps, end_lr=end_lr)
        # `gemma_expert.lm_head` is never used by the forward: it can never receive a gradient,
        # so it is left out of the optimizer (torch.optim.AdamW skips grad-less parameters as well).
        dead = model.paligemma_with_expert.gemma_expert.lm_head.weight
        params = [p for p in model.parameters() if p is not dead]
        self.engine = ShardedDataParallel(params, world_size=world_size, rank=rank, group=group, ops=shard_ops, betas=betas,
                                          eps=eps, weight_decay=weight_decay, max_grad_norm=clip_norm,
                                          bucket_bytes=bucket_bytes)  # fmt: skip
        self.global_step = 0
        self.last_grad_norm = None

    def lr(self) -> float:
        return lr_schedule(self.global_step, **self.sched)

    def train_step(self, observation, actions, noise=None, time=None) -> torch.Tensor:
        """Returns the (local) mean loss as a 0-d device tensor — no host sync inside the step."""
        lr = self.lr()
        if getattr(self.model, "_engine", None) is not None:
            self.model.invalidate_inference_engine()  # the step below rewrites the weights through raw pointers
        losses = self.model(observation, actions, noise=noise, time=time)
        loss = losses.mean()
        loss.backward()
        self.last_grad_norm = self.engine.step(lr)
        self.global_step += 1
        return loss.detach()

    # ---------------------------------------------------------------------------------------- checkpoints
    def save_checkpoint(self, checkpoint_dir: str) -> str:
        """train_pytorch.py:149-194. Rank 0 writes the model; every rank writes its optimizer shard."""
        from .checkpoint import save_model_safetensors

        step = self.global_step
        final = os.path.join(checkpoint_dir, f"{step}")
        tmp = os.path.join(checkpoint_dir, f"tmp_{step}")
        if self.rank == 0:
            if os.path.exists(tmp):
                shutil.rmtree(tmp)
            os.makedirs(tmp, exist_ok=True)
            save_model_safetensors(self.model, os.path.join(tmp, "model.safetensors"))
            torch.save({"global_step": step, "world_size": self.world}, os.path.join(tmp, "metadata.pt"))
        if self.world > 1:
            torch.distributed.barrier()
        torch.save(self.engine.state_dict(), os.path.join(tmp, f"optimizer_rank{self.rank}.pt"))
        if self.world > 1:
            torch.distributed.barrier()
        if self.rank == 0:
            if os.path.exists(final):
                shutil.rmtree(final)
            os.rename(tmp, final)
        if self.world > 1:
            torch.distributed.barrier()
        return final

    def load_checkpoint(self, checkpoint_dir: str) -> int:
        """Resume from the highest numeric step directory (train_pytorch.py:197-259)."""
        from .checkpoint import load_model_safetensors

        steps = [int(d) for d in os.listdir(checkpoint_dir) if d.isdigit()]
        if not steps:
            raise FileNotFoundError(f"No checkpoints found in {checkpoint_dir}")
        step = max(steps)
        d = os.path.join(checkpoint_dir, str(step))
        load_model_safetensors(self.model, os.path.join(d, "model.safetensors"))
        self.engine.load_state_dict(torch.load(os.path.join(d, f"optimizer_rank{self.rank}.pt"), map_location=self.engine.device))
        self.global_step = torch.load(os.path.join(d, "metadata.pt"))["global_step"]
        return self.global_step
