"""Sharded data-parallel engine: gradients reduce-scattered, optimizer state (f32 master + Adam moments) and the
parameter update sharded 1/N per GPU, updated bf16 parameters all-gathered — RCCL over xGMI, overlapped with
backward.

What it replaces: DistributedDataParallel's bucketed all-reduce in the reference's PyTorch trainer
(train_pytorch.py:440-447) and the GSPMD FSDP sharding of the JAX trainer (training/sharding.py:48-102).

MI355X-first design:
  * 288 GB of HBM per GPU: the bf16 model copy (7.2 GB) stays fully resident on every GPU, so forward/backward
    never wait on a parameter gather (no all-gather in the critical path of a layer, no re-gather for backward);
    what is sharded is what is big: 12 B/param of f32 master + moments and the gradient reduction.
  * per step and GPU: one reduce-scatter (grads) + one all-gather (updated params) of the flat buffers,
    = 2 x 7/8 x 7.2 GB over xGMI instead of DDP's all-reduce (same bytes) plus a replicated 16 B/param optimizer
    pass; FSDP's third collective (backward re-gather) is not needed at all.
  * few, large collectives: parameters are packed (reverse registration order = gradient-ready order) into flat
    buckets of ~512 MB; a bucket's reduce-scatter is issued from the autograd hook of its last gradient, on RCCL's
    stream, while backward keeps computing earlier layers.
  * the whole optimizer is 3 kernels per bucket on flat shards (sum-of-squares, clip coefficient kept on device,
    fused AdamW) — no per-tensor launches, no host sync.

The arithmetic is pluggable (`ShardOps`) only so the collective/partition logic can be exercised on CPU with the
gloo backend in tests; the product default is the HIP kernels and it raises without them.
"""

from __future__ import annotations

import math

import functools
import os

import torch
import torch.distributed as dist

F32, BF16 = torch.float32, torch.bfloat16


class HipShardOps:
    """sum-of-squares / clip coefficient / fused AdamW on flat shards via libkai0hip.so."""

    def sumsq(self, grad, out):
        from .optim import sumsq_accumulate_

        sumsq_accumulate_(grad, out)

    def clip_coef(self, sumsq, max_norm, coef, norm):
        from .optim import clip_coef_

        clip_coef_(sumsq, max_norm, coef, norm)

    def adamw(self, master, m, v, grad, param, *, lr, beta1, beta2, eps, wd, step, clip_coef):
        from .optim import adamw_step_

        adamw_step_(master, m, v, grad, param, lr=lr, beta1=beta1, beta2=beta2, eps=eps, wd=wd, step=step, clip_coef=clip_coef)


class _Bucket:
    def __init__(self, params, dtype, world, rank, device, align=256, alias_shard=True):
        self.params = params
        self.dtype = dtype
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8  # keep every parameter 16-B aligned inside the flat buffer
        unit = world * align
        self.numel = (off + unit - 1) // unit * unit
        self.shard = self.numel // world
        self.flat_param = torch.zeros(self.numel, dtype=dtype, device=device)
        self.flat_grad = torch.zeros(self.numel, dtype=dtype, device=device)
        for p, o in zip(params, self.offsets):
            self.flat_param[o : o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o : o + p.numel()].view(p.shape)
        lo = rank * self.shard
        self.param_shard = self.flat_param[lo : lo + self.shard]
        # one GPU: the "shard" is the whole buffer — alias it instead of copying
        self.grad_shard = self.flat_grad if (world == 1 and alias_shard) else torch.zeros(self.shard, dtype=dtype, device=device)
        for p, o in zip(params, self.offsets):
            # producers that can write a gradient in place (LinearFn wgrad, EmbedFn) pick this up
            p._kai0_grad_out = self.flat_grad[o : o + p.numel()].view(p.shape)
        self.master = self.param_shard.to(F32).clone()
        self.exp_avg = torch.zeros(self.shard, dtype=F32, device=device)
        self.exp_avg_sq = torch.zeros(self.shard, dtype=F32, device=device)
        self.pending = len(params)
        self.arrived = set()
        self.stale = set()  # parameters whose slice of flat_grad still holds the previous step's gradient
        self.work = None


class ShardedDataParallel:
    def __init__(self, params, *, world_size: int, rank: int, group=None, ops=None, betas=(0.9, 0.95), eps=1e-8,
                 weight_decay=1e-10, max_grad_norm=1.0, bucket_bytes: int = 512 << 20):  # fmt: skip
        self.world, self.rank, self.group = world_size, rank, group
        self.ops = ops or HipShardOps()
        self.betas, self.eps, self.wd, self.max_grad_norm = betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        seen, uniq = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        if not uniq:
            raise ValueError("no trainable parameters")
        self.device = uniq[0].device
        # KAI0_FORCE_COLLECTIVES=1: run the reduce-scatter / all-gather calls even with one rank (validates the RCCL call
        # pattern on a single-GPU box; tools/nccl_same_gpu_probe.py)
        self.collectives = world_size > 1 or (os.environ.get("KAI0_FORCE_COLLECTIVES") == "1" and dist.is_initialized())
        self.backend = dist.get_backend(group) if self.collectives else "none"
        # gradients become ready in (roughly) reverse registration order: pack buckets in that order
        self.buckets: list[_Bucket] = []
        for dtype in (BF16, F32):
            cur, cur_bytes = [], 0
            for p in reversed([q for q in uniq if q.dtype == dtype]):
                cur.append(p)
                cur_bytes += p.numel() * p.element_size()
                if cur_bytes >= bucket_bytes:
                    self.buckets.append(_Bucket(cur, dtype, world_size, rank, self.device, alias_shard=not self.collectives))
                    cur, cur_bytes = [], 0
            if cur:
                self.buckets.append(_Bucket(cur, dtype, world_size, rank, self.device, alias_shard=not self.collectives))
        other = [p for p in uniq if p.dtype not in (BF16, F32)]
        if other:
            raise TypeError(f"unsupported parameter dtype {other[0].dtype}")
        self._where = {}
        for b in self.buckets:
            for p, o in zip(b.params, b.offsets):
                self._where[p] = (b, o)
                p.register_post_accumulate_grad_hook(self._on_grad)
                p._kai0_grad_done = functools.partial(self._on_grad_inplace, p)
        self._sumsq = torch.zeros(1, dtype=F32, device=self.device)
        self._coef = torch.ones(1, dtype=F32, device=self.device)
        self._norm = torch.zeros(1, dtype=F32, device=self.device)

    # ------------------------------------------------------------------------------------------ collectives
    def _reduce_scatter_avg(self, b: _Bucket):
        if not self.collectives:
            return None  # grad_shard aliases flat_grad
        if self.backend == "nccl":  # RCCL
            return dist.reduce_scatter_tensor(b.grad_shard, b.flat_grad, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        # gloo (CPU tests): no reduce_scatter / AVG — all-reduce then keep the local shard
        dist.all_reduce(b.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        lo = self.rank * b.shard
        b.grad_shard.copy_(b.flat_grad[lo : lo + b.shard] / self.world)
        return None

    def _all_gather(self, b: _Bucket):
        if not self.collectives:
            return None
        if self.backend == "nccl":
            return dist.all_gather_into_tensor(b.flat_param, b.param_shard, group=self.group, async_op=True)
        chunks = [torch.empty_like(b.param_shard) for _ in range(self.world)]
        dist.all_gather(chunks, b.param_shard.clone(), group=self.group)
        for r, c in enumerate(chunks):
            b.flat_param[r * b.shard : (r + 1) * b.shard].copy_(c)
        return None

    # ------------------------------------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        """post-accumulate-grad hook: a gradient that autograd materialised in p.grad (ops that do not know about the
        flat buffer, e.g. plain torch modules) is copied into the parameter's slice."""
        if p.grad is None:  # the producer returned None: either _on_grad_inplace already ran, or there is no gradient
            return
        b, o = self._where[p]
        b.flat_grad[o : o + p.numel()].copy_(p.grad.reshape(-1))
        p.grad = None
        self._arrived(p, b)

    def _on_grad_inplace(self, p):
        """called by the backward shims (ops._grad_ret) after they wrote the gradient straight into p._kai0_grad_out;
        they then return None to autograd, so no clone, no accumulate and no hook happen for this parameter."""
        self._arrived(p, self._where[p][0])

    def _arrived(self, p, b):
        if id(p) in b.arrived:
            raise RuntimeError("a parameter received two gradients in one step: the in-place gradient path supports "
                               "one use per parameter per backward")  # fmt: skip
        b.arrived.add(id(p))
        b.pending -= 1
        if b.pending == 0:
            b.work = self._reduce_scatter_avg(b)  # overlaps with the rest of backward

    # ------------------------------------------------------------------------------------------------- step
    @torch.no_grad()
    def step(self, lr: float):
        """Finish the gradient reduction, clip by the global norm, update the local shards, gather the parameters.
        Returns the global (pre-clip) gradient norm as a 1-element device tensor."""
        self.step_count += 1
        for b in self.buckets:
            if b.pending > 0:  # parameters that received no gradient this step contribute zeros
                for p, o in zip(b.params, b.offsets):
                    if id(p) in b.stale and id(p) not in b.arrived:  # ... not what an earlier step left in their slice
                        b.flat_grad[o : o + p.numel()].zero_()
                        b.stale.discard(id(p))
                b.work = self._reduce_scatter_avg(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
        coef = None
        if self.max_grad_norm is not None:
            self._sumsq.zero_()
            for b in self.buckets:
                self.ops.sumsq(b.grad_shard, self._sumsq)
            if self.world > 1:
                dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=self.group)
            self.ops.clip_coef(self._sumsq, self.max_grad_norm, self._coef, self._norm)
            coef = self._coef
        works = []
        for b in self.buckets:
            self.ops.adamw(b.master, b.exp_avg, b.exp_avg_sq, b.grad_shard, b.param_shard, lr=lr, beta1=self.betas[0],
                           beta2=self.betas[1], eps=self.eps, wd=self.wd, step=self.step_count, clip_coef=coef)  # fmt: skip
            works.append(self._all_gather(b))
        for w in works:
            if w is not None:
                w.wait()
        # Every gradient producer overwrites its parameter's whole slice, so the 7 GB of flat gradients are NOT cleared per
        # step: a slice is zeroed only if its producer accumulates into it (the embedding scatter: `_kai0_grad_accumulates`)
        # or, lazily in the next step(), if it holds an old gradient and no new one arrived.
        full = os.environ.get("KAI0_ZERO_GRADS") == "full"  # diagnostics: clear everything, as before
        for b in self.buckets:
            if full:
                b.flat_grad.zero_()
                b.stale.clear()
            for p, o in zip(b.params, b.offsets):
                if id(p) in b.arrived and not full:
                    if getattr(p, "_kai0_grad_accumulates", False):
                        b.flat_grad[o : o + p.numel()].zero_()
                        b.stale.discard(id(p))
                    else:
                        b.stale.add(id(p))
            b.pending = len(b.params)
            b.arrived.clear()
        return self._norm

    # ------------------------------------------------------------------------------------------ checkpointing
    def state_dict(self):
        return {
            "step": self.step_count,
            "shards": [{"master": b.master, "exp_avg": b.exp_avg, "exp_avg_sq": b.exp_avg_sq} for b in self.buckets],
            "world": self.world,
            "rank": self.rank,
        }

    def load_state_dict(self, sd):
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise ValueError("sharded optimizer state was saved with a different world size / rank")
        self.step_count = sd["step"]
        for b, s in zip(self.buckets, sd["shards"], strict=True):
            b.master.copy_(s["master"])
            b.exp_avg.copy_(s["exp_avg"])
            b.exp_avg_sq.copy_(s["exp_avg_sq"])

    def optimizer_state_bytes(self) -> int:
        return sum(3 * 4 * b.shard for b in self.buckets)
