"""Sharded data-parallel engine over RCCL / xGMI: one process per GPU, gradients reduce-scattered as backward produces them,
optimizer state (f32 master + Adam moments) and the parameter update sharded 1/N per GPU, bf16 parameters all-gathered.

What it replaces: DistributedDataParallel's bucketed all-reduce in the reference's PyTorch trainer
(train_pytorch.py:440-447) and the GSPMD FSDP sharding of the JAX trainer (training/sharding.py:48-102).

Two modes over the same buckets, collectives and kernels:

  * "zero2" (default) — 288 GB of HBM per GPU: the bf16 model copy (7.2 GB) stays resident on every GPU, so no layer ever
    waits on a gather inside forward / backward and nothing is re-gathered for backward.  Per step and GPU: one
    reduce-scatter of the gradients + one all-gather of the updated parameters = 2 x (N-1)/N x 7.2 GB over xGMI.  The
    all-gathers are issued per bucket, in forward-use order, right after each bucket's AdamW, and are NOT waited for in
    step(): the next forward waits per unit (`pre_forward`), so the gather of layer k+1.. hides behind the compute of
    layers ..k.
  * "fsdp" — parameters sharded too (training/sharding.py:48-102; north_star "optimizer/grad/param FSDP-style"): the
    persistent bf16 copy is the 1/N shard; a bucket's full parameters exist only around its use — gathered one bucket
    ahead of the forward, freed after it, gathered again one bucket ahead of the backward, freed when the bucket's
    gradients have been reduce-scattered.  3 x (N-1)/N x 7.2 GB per step; saves (N-1)/N x 7.2 GB of HBM.  The full-size
    gradient buffer of a bucket is a staging buffer in this mode: it exists from the start of the bucket's backward until
    its reduce-scatter has finished (one bucket later) — only the 1/N gradient shard persists.

Unit = what the model uses together (one SigLIP layer; one joint Gemma-2B + expert layer; the embeddings; the heads): the
model reports them in forward-use order (`sharding_units()`), consecutive units are packed into buckets of ~512 MB, a
bucket's reduce-scatter is issued from inside backward the moment its last gradient has been written (the backward shims
write straight into the flat gradient buffer: no copy, no autograd accumulation).  SUM collectives only (the loss is
scaled by 1/N), so the gloo tests on CPU drive exactly the call path RCCL runs.

Gradient reduce-scatter, two forms (`rs_algo`, env KAI0_RS_ALGO): "rccl" — the library's `reduce_scatter_tensor` (default); "alltoall"
— the all-pairs form for the fully connected xGMI mesh: one `all_to_all_single` sends slice j of the bucket straight to rank j
(every link of the GPU busy at once, one hop), `ShardOps.sum_chunks` (kai0_sum_chunks) adds the N received copies of the own
slice in f32 with one rounding.  Everything around it (issue from inside backward, staging, waits, bookkeeping) is shared.

Evidence for the first multi-GPU run (`comm_profile = True`, read with `comm_report()`): every place where the compute
stream has to wait for a collective (a parameter gather in pre_forward / pre_backward, the reduce-scatters and the norm
all-reduce in step()) is bracketed by two events on the compute stream, so `comm_exposed_ms` is the time the chip sat in those
waits — what overlap did NOT hide — next to the bytes each rank moved.

The whole optimizer is 3 kernels per bucket on flat shards (sum of squares, clip coefficient kept on device, fused
AdamW) — no per-tensor launches, no host sync.  The arithmetic is pluggable (`ShardOps`) only so the collective /
partition logic can be exercised on CPU with gloo; the product default is the HIP kernels and raises without them.
"""

from __future__ import annotations

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

    def sum_chunks(self, src, chunks, out):
        from .optim import sum_chunks_

        sum_chunks_(src, chunks, out)

    def adamw(self, master, m, v, grad, param, *, lr, beta1, beta2, eps, wd, step, clip_coef):
        from .optim import adamw_step_

        adamw_step_(master, m, v, grad, param, lr=lr, beta1=beta1, beta2=beta2, eps=eps, wd=wd, step=step, clip_coef=clip_coef)


    def adamw_rows(self, master, m, v, grad, param, row_len, row_active, *, lr, beta1, beta2, eps, wd, step, clip_coef):
        from .optim import adamw_rows_step_

        adamw_rows_step_(master, m, v, grad, param, row_len, row_active, lr=lr, beta1=beta1, beta2=beta2, eps=eps, wd=wd, step=step,
                         clip_coef=clip_coef)  # fmt: skip


class _Bucket:
    """Flat buffers of the same-dtype parameters of a run of consecutive units."""

    @staticmethod
    def layout(params, world: int, align: int = 256):
        """(offsets, padded numel) of the flat buffer: every parameter 16-B aligned, the total a multiple of world * align elements
        (so every rank's shard is a whole number of 256-element kernel blocks).  Pure shape arithmetic: works on meta tensors."""
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += (p.numel() + 7) // 8 * 8
        unit = world * align
        return offsets, (off + unit - 1) // unit * unit

    def __init__(self, params, names, dtype, world, rank, device, *, alias_shard: bool, fsdp: bool, align=256):
        self.params, self.names, self.dtype = params, names, dtype
        self.offsets, self.numel = self.layout(params, world, align)
        self.shard = self.numel // world
        self.lo = rank * self.shard
        self.flat_param = torch.zeros(self.numel, dtype=dtype, device=device)
        self.flat_grad = torch.zeros(self.numel, dtype=dtype, device=device)
        for p, o in zip(params, self.offsets):
            self.flat_param[o : o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o : o + p.numel()].view(p.shape)
            # producers that can write a gradient in place (LinearFn wgrad, EmbedFn, ...) pick this up
            p._kai0_grad_out = self.flat_grad[o : o + p.numel()].view(p.shape)
        self.fsdp = fsdp
        self.full_nbytes = self.flat_param.untyped_storage().nbytes()
        self.resident = True  # fsdp: whether flat_param's storage currently exists
        self.grad_nbytes = self.flat_grad.untyped_storage().nbytes()
        self.grad_resident = True  # fsdp: whether flat_grad's storage currently exists (staging buffer, see the module docstring)
        self.grad_freeable = False  # fsdp: set once a step has shown that the bucket's backward announces itself first
        self.announced = False      # this step: _pre_backward ran for the bucket's group before its first gradient arrived
        self.param_shard = None  # set by carve_shards() (after the construction-time broadcast)
        # one GPU: the "shard" is the whole buffer — alias it instead of copying
        self.grad_shard = self.flat_grad if alias_shard else torch.zeros(self.shard, dtype=dtype, device=device)
        self.master = self.exp_avg = self.exp_avg_sq = None
        self.pending = len(params)
        self.arrived = set()
        self.stale = set()  # parameters whose slice of flat_grad still holds the previous step's gradient
        self.rs_work = None  # in-flight reduce-scatter of the gradients
        self.ag_work = None  # in-flight all-gather of the parameters

    def carve_shards(self):
        sl = self.flat_param[self.lo : self.lo + self.shard]
        # fsdp: the shard outlives the full buffer, so it owns its memory; zero2: a slice (in-place all-gather)
        self.param_shard = sl.clone() if self.fsdp else sl
        self.master = sl.to(F32).clone()
        self.exp_avg = torch.zeros(self.shard, dtype=F32, device=sl.device)
        self.exp_avg_sq = torch.zeros(self.shard, dtype=F32, device=sl.device)


class _AllPairsReduce:
    """In-flight all-pairs reduce-scatter of one bucket: the all-to-all that delivers every peer's copy of this rank's
    gradient slice, then (at wait time, on the waiting stream) the f32 sum of the `world` received slices into the shard."""

    def __init__(self, work, recv, bucket, world, ops):
        self.work, self.recv, self.bucket, self.world, self.ops = work, recv, bucket, world, ops

    def wait(self):
        self.work.wait()
        self.ops.sum_chunks(self.recv, self.world, self.bucket.grad_shard)
        if self.recv.is_cuda:  # allocated under the stream that issued the collective, read here under the waiting one
            self.recv.record_stream(torch.cuda.current_stream(self.recv.device))
        self.recv = None


class _BackwardMark(torch.autograd.Function):
    """Identity on a unit's outputs whose backward tells the engine that the unit's backward is about to start
    (fsdp: its parameters are gathered again, the bucket before it is prefetched)."""

    @staticmethod
    def forward(ctx, engine, unit, *tensors):
        ctx.engine, ctx.unit = engine, unit
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        ctx.engine._pre_backward(ctx.unit)
        return (None, None, *grads)


def group_units(ulist, bucket_bytes: int):
    """Consecutive units (forward-use order; a unit is never split) packed into groups of >= bucket_bytes: [(unit names, params)]."""
    groups, cur, cur_units, cur_bytes = [], [], [], 0
    for uname, ps in ulist:
        cur += ps
        cur_units.append(uname)
        cur_bytes += sum(p.numel() * p.element_size() for p in ps)
        if cur_bytes >= bucket_bytes:
            groups.append((cur_units, cur))
            cur, cur_units, cur_bytes = [], [], 0
    if cur:
        groups.append((cur_units, cur))
    return groups


def plan_partition(units, *, world_size: int, bucket_bytes: int, exclude=()):
    """The unit -> group -> bucket partition the engine WOULD build, from shapes alone (meta tensors are fine): what is gathered /
    reduce-scattered together, how many bytes, and what each rank keeps.  `units` as `model.sharding_units()` returns them.
    Returns [{units, buckets: [{dtype, numel, shard, bytes, n_params}]}] plus totals — tests/test_sharded_cpu.py checks the real
    pi0.5 model's plan at world 8 against SURVEY.md section 8e with it, and bench.py prints it into the `comm` object."""
    skip = {id(p) for p in exclude}
    seen, ulist = set(), []
    for uname, ups in units:
        ps = [p for p in ups if id(p) not in seen and id(p) not in skip and p.requires_grad]
        seen.update(id(p) for p in ps)
        if ps:
            ulist.append((uname, ps))
    plan = []
    for unames, ps in group_units(ulist, bucket_bytes):
        bk = []
        for dtype in (BF16, F32):
            dps = [p for p in ps if p.dtype == dtype]
            if dps:
                _, numel = _Bucket.layout(dps, world_size)
                esz = 2 if dtype == BF16 else 4
                bk.append({"dtype": str(dtype).replace("torch.", ""), "numel": numel, "shard": numel // world_size, "bytes": numel * esz,
                           "n_params": len(dps)})
        plan.append({"units": list(unames), "buckets": bk})
    total = sum(b["bytes"] for g in plan for b in g["buckets"])
    return {"groups": plan, "total_bytes": total, "world_size": world_size,
            # per rank and step over the links: zero2 = reduce-scatter + all-gather, fsdp = + the backward's second all-gather
            "bytes_per_rank_per_step": {"zero2": 2 * (world_size - 1) / world_size * total, "fsdp": 3 * (world_size - 1) / world_size * total}}


class ShardedDataParallel:
    def __init__(self, params, *, world_size: int, rank: int, group=None, ops=None, betas=(0.9, 0.95), eps=1e-8,
                 weight_decay=1e-10, max_grad_norm=1.0, bucket_bytes: int = 512 << 20, units=None, mode: str = "zero2",
                 prefetch: int = 1, sync_params: bool = True, rs_algo: str | None = None):  # fmt: skip
        """`params`: parameters, or (name, parameter) pairs (names make the checkpoint world-size independent).
        `units`: [(unit name, [parameters])] in forward-use order; parameters not listed form a last unit "rest"."""
        if mode not in ("zero2", "fsdp"):
            raise ValueError(f"mode must be 'zero2' or 'fsdp', got {mode!r}")
        rs_algo = rs_algo or os.environ.get("KAI0_RS_ALGO", "rccl")
        if rs_algo not in ("rccl", "alltoall"):
            raise ValueError(f"rs_algo must be 'rccl' or 'alltoall', got {rs_algo!r}")
        self.rs_algo = rs_algo
        self.world, self.rank, self.group = world_size, rank, group
        self.ops = ops or HipShardOps()
        self.betas, self.eps, self.wd, self.max_grad_norm = betas, eps, weight_decay, max_grad_norm
        self.prefetch = max(0, int(prefetch))
        self.step_count = 0
        seen, uniq, names = set(), [], {}
        for i, item in enumerate(params):
            name, p = item if isinstance(item, tuple) else (f"param.{i}", item)
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
                names[id(p)] = name
        if not uniq:
            raise ValueError("no trainable parameters")
        other = [p for p in uniq if p.dtype not in (BF16, F32)]
        if other:
            raise TypeError(f"unsupported parameter dtype {other[0].dtype}")
        self.device = uniq[0].device
        # KAI0_FORCE_COLLECTIVES=1: run the collectives even with one rank (validates the RCCL call pattern on a one-GPU box)
        self.collectives = world_size > 1 or (os.environ.get("KAI0_FORCE_COLLECTIVES") == "1" and dist.is_initialized())
        self.mode = mode if self.collectives else "zero2"  # without peers there is nothing to shard
        fsdp = self.mode == "fsdp"

        # ---- units -> buckets (forward-use order; a unit is never split) ----------------------------------------
        ulist, assigned = [], set()
        for uname, ups in units or []:
            ps = [p for p in ups if id(p) in seen and id(p) not in assigned]
            assigned.update(id(p) for p in ps)
            if ps:
                ulist.append((uname, ps))
        rest = [p for p in uniq if id(p) not in assigned]
        if rest:
            if units:
                ulist.append(("rest", rest))
            else:  # no unit information: every parameter its own unit, registration order
                ulist += [(names[id(p)], [p]) for p in rest]
        self.buckets: list[_Bucket] = []
        self.groups: list[list[int]] = []  # group g -> bucket indices (one per dtype present)
        self.unit_group: dict[str, int] = {}
        self._group_last_unit: list[str] = []
        for cur_units, cur in group_units(ulist, bucket_bytes):
            g = len(self.groups)
            ids = []
            for dtype in (BF16, F32):
                ps = [p for p in cur if p.dtype == dtype]
                if ps:
                    ids.append(len(self.buckets))
                    self.buckets.append(_Bucket(ps, [names[id(p)] for p in ps], dtype, world_size, rank, self.device,
                                                alias_shard=not self.collectives, fsdp=fsdp))  # fmt: skip
            self.groups.append(ids)
            for u in cur_units:
                self.unit_group[u] = g
            self._group_last_unit.append(cur_units[-1])

        self._where = {}
        for bi, b in enumerate(self.buckets):
            for p, o in zip(b.params, b.offsets):
                self._where[p] = (b, o)
                p.register_post_accumulate_grad_hook(self._on_grad)
                p._kai0_grad_done = functools.partial(self._on_grad_inplace, p)
                # fsdp: the flat gradient buffer is a staging buffer that may be given back between uses; a producer that is about
                # to write into `_kai0_grad_out` calls this first (ops._grad_dst), so the view never points at a 0-byte storage
                p._kai0_grad_ensure = functools.partial(self._ensure_grad, b)
        self._bucket_group = {}
        for g, ids in enumerate(self.groups):
            for bi in ids:
                self._bucket_group[id(self.buckets[bi])] = g

        # replicas must start identical (DDP broadcasts rank 0's module at construction): a rank that initialised or loaded
        # different weights would otherwise contribute its own slice to the first all-gather, silently
        if self.collectives and sync_params:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            for b in self.buckets:
                dist.broadcast(b.flat_param, src=src, group=group)
        for b in self.buckets:
            b.carve_shards()
        self._sumsq = torch.zeros(1, dtype=F32, device=self.device)
        self._coef = torch.ones(1, dtype=F32, device=self.device)
        self._norm = torch.zeros(1, dtype=F32, device=self.device)
        self._in_backward = False
        # KAI0_SPARSE_EMBED=0: the embedding table's rows all go through the dense update (A/B; the result is bit-identical)
        self._sparse_rows = os.environ.get("KAI0_SPARSE_EMBED", "1") != "0"
        self._rs_inflight: list[_Bucket] = []  # fsdp: buckets whose reduce-scatter runs out of a staging buffer
        self.comm_profile = False
        self._comm_events: list[tuple[str, object, object]] = []
        if fsdp:
            for g in range(len(self.groups)):
                self._release(g)

    # ------------------------------------------------------------------------------------------ collectives
    def _reduce_scatter(self, b: _Bucket):
        """SUM of the ranks' gradients (the loss carries the 1/N), each rank keeping its 1/N slice."""
        if not self.collectives:
            return None  # grad_shard aliases flat_grad
        self._join_streams()  # the bucket's gradients were written under two streams (ops.side_stream): behind both
        if self.rs_algo == "alltoall":
            # all-pairs form for the fully connected xGMI mesh: slice j of the flat gradients goes straight to rank j over the
            # link the two share (N-1 links busy at once, one hop), and the N copies of the own slice are summed locally in f32
            # (kai0_sum_chunks) — the library's ring moves the same (N-1)/N bytes per rank but in N-1 dependent hops, each bound by
            # ONE link and each rounding the running sum to bf16.  Costs a receive buffer of the bucket's size while in flight.
            recv = torch.empty_like(b.flat_grad)
            work = dist.all_to_all_single(recv, b.flat_grad, group=self.group, async_op=True)
            return _AllPairsReduce(work, recv, b, self.world, self.ops)
        return dist.reduce_scatter_tensor(b.grad_shard, b.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _join_streams(self):
        if self.device.type == "cuda":
            from . import ops

            ops.join_streams(self.device)

    def _all_gather(self, b: _Bucket):
        if not self.collectives:
            return None
        return dist.all_gather_into_tensor(b.flat_param, b.param_shard, group=self.group, async_op=True)

    # ---- exposed-communication bookkeeping ----------------------------------------------------------------
    def _wait(self, work, kind: str):
        """`work.wait()` makes the CURRENT stream wait for the collective; with `comm_profile` the wait is bracketed by two
        events on that stream: their distance is the time the compute stream stalled here (0 if the collective was done)."""
        if work is None:
            return
        if self.comm_profile and self.device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            self._comm_events.append((kind, e0, e1))
        else:
            work.wait()

    def comm_bytes_per_step(self) -> dict:
        """Bytes this rank sends (= receives) per training step: ring-equivalent (N-1)/N of every bucket per collective."""
        n = self.world
        if not self.collectives or n <= 1:
            return {"reduce_scatter": 0, "all_gather": 0, "total": 0}
        full = sum(b.numel * b.flat_param.element_size() for b in self.buckets)
        rs = full * (n - 1) // n
        ag = rs * (2 if self.mode == "fsdp" else 1)  # fsdp gathers for the forward and again for the backward
        return {"reduce_scatter": rs, "all_gather": ag, "total": rs + ag}

    def comm_report(self, reset: bool = True) -> dict:
        """Exposed communication since the last report (synchronises): ms the compute stream spent waiting, by kind."""
        out = {"all_gather_wait": 0.0, "reduce_scatter_wait": 0.0, "norm_all_reduce": 0.0}
        if self._comm_events:
            torch.cuda.synchronize(self.device)
            for kind, e0, e1 in self._comm_events:
                out[kind] = out.get(kind, 0.0) + e0.elapsed_time(e1)
        out["comm_exposed_ms"] = sum(out.values())
        out["waits"] = len(self._comm_events)
        if reset:
            self._comm_events = []
        return out

    # ---- parameter residency (fsdp) / gather hand-off (zero2) ---------------------------------------------
    def _issue_gather(self, g: int):
        if g < 0 or g >= len(self.groups):
            return
        for bi in self.groups[g]:
            b = self.buckets[bi]
            if b.fsdp and not b.resident:
                b.flat_param.untyped_storage().resize_(b.full_nbytes)
                b.resident = True
                b.ag_work = self._all_gather(b)

    def _ensure(self, g: int) -> bool:
        """The parameters of group g are complete on this GPU (for every kernel enqueued on the current stream from now on).
        Returns whether the current stream was made to wait for a gather."""
        self._issue_gather(g)
        waited = False
        for bi in self.groups[g]:
            b = self.buckets[bi]
            if b.ag_work is not None:
                self._wait(b.ag_work, "all_gather_wait")
                b.ag_work = None
                waited = True
        return waited

    def _release(self, g: int):
        for bi in self.groups[g]:
            b = self.buckets[bi]
            if b.fsdp and b.resident:
                if b.ag_work is not None:
                    b.ag_work.wait()
                    b.ag_work = None
                b.flat_param.untyped_storage().resize_(0)
                b.resident = False

    # ---- gradient staging buffers (fsdp) ---------------------------------------------------------------------
    def _ensure_grad(self, b: _Bucket):
        if not b.grad_resident:
            b.flat_grad.untyped_storage().resize_(b.grad_nbytes)
            b.flat_grad.zero_()  # parameters without a gradient this step contribute zeros; accumulating producers start at 0
            b.grad_resident = True
            b.stale.clear()

    def _free_grad(self, b: _Bucket):
        """after the bucket's reduce-scatter has been waited for on the current stream (memory reuse is stream-ordered)"""
        if b.fsdp and b.grad_freeable and b.grad_resident and b.grad_shard is not b.flat_grad:
            b.flat_grad.untyped_storage().resize_(0)
            b.grad_resident = False
            b.stale.clear()

    def _retire_reduce_scatters(self, keep: int):
        while len(self._rs_inflight) > keep:
            b = self._rs_inflight.pop(0)
            if b.rs_work is not None:
                self._wait(b.rs_work, "reduce_scatter_wait")
                b.rs_work = None
            self._free_grad(b)

    def begin_step(self):
        """Called at the start of every training step: state that an aborted backward (an exception, an evaluation with
        gradients but no step()) may have left behind does not leak into this step."""
        self._in_backward = False
        # a backward that did not reach step(): drop its collectives and its bucket bookkeeping, or this step would raise "a
        # parameter received two gradients", launch a reduce-scatter on stale counts, or retire a dead work handle
        for b in self._rs_inflight:
            if b.rs_work is not None:
                self._wait(b.rs_work, "reduce_scatter_wait")
                b.rs_work = None
        self._rs_inflight.clear()
        for b in self.buckets:
            if b.rs_work is not None:
                self._wait(b.rs_work, "reduce_scatter_wait")
                b.rs_work = None
            if b.arrived or b.pending != len(b.params):
                if b.grad_resident:  # the aborted step's partial gradients must not be taken for this step's
                    for p, o in zip(b.params, b.offsets):
                        if id(p) not in b.arrived:
                            continue
                        if getattr(p, "_kai0_grad_accumulates", False):
                            # a producer that scatter-ADDS into a slice it assumes zero (the embedding table): "stale" would only
                            # re-zero it when no new gradient arrives — the aborted gradient would be added to the new one
                            b.flat_grad[o : o + p.numel()].zero_()
                        else:
                            b.stale.add(id(p))
                b.arrived.clear()
                b.pending = len(b.params)
            b.announced = False
        if self.device.type == "cuda":
            from . import ops

            ops.reset_backward_state()

    # ---- unit hooks: called by the model around every unit's compute ------------------------------------------
    def pre_forward(self, unit: str) -> bool:
        """Returns True if the current stream had to wait for this unit's parameter gather (other streams that read the
        parameters must then be ordered behind the current one)."""
        g = self.unit_group.get(unit)
        if g is None:
            return False
        waited = self._ensure(g)
        if self.mode == "fsdp" and not self._in_backward:
            for k in range(1, self.prefetch + 1):
                self._issue_gather(g + k)
        return waited

    def post_forward(self, unit: str, *tensors):
        """Marks the end of a unit's forward; returns `tensors` (wrapped so that the unit's backward announces itself)."""
        g = self.unit_group.get(unit)
        if g is None or self.mode != "fsdp":
            return tensors
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            tensors = _BackwardMark.apply(self, unit, *tensors)
        if not self._in_backward and self._group_last_unit[g] == unit and g != len(self.groups) - 1:
            self._release(g)  # (the last group is used again at once by the backward: it stays)
        return tensors

    def _pre_backward(self, unit: str):
        self._in_backward = True
        g = self.unit_group[unit]
        for bi in self.groups[g]:
            b = self.buckets[bi]
            if not b.arrived:
                b.announced = True
            self._ensure_grad(b)
        self._ensure(g)
        for k in range(1, self.prefetch + 1):
            self._issue_gather(g - k)

    def wait_params(self):
        """Every parameter complete on this GPU (callers without unit hooks: plain modules, checkpointing, inference)."""
        for g in range(len(self.groups)):
            self._ensure(g)

    materialize = wait_params

    def release_params(self):
        """fsdp: drop the full parameter buffers again (after `materialize()` for a checkpoint / an inference call)."""
        for g in range(len(self.groups)):
            self._release(g)

    # ------------------------------------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        """post-accumulate-grad hook: a gradient that autograd materialised in p.grad (ops that do not know about the
        flat buffer, e.g. plain torch modules) is copied into the parameter's slice."""
        if p.grad is None:  # the producer returned None: either _on_grad_inplace already ran, or there is no gradient
            return
        b, o = self._where[p]
        self._ensure_grad(b)
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
        self._in_backward = True
        b.arrived.add(id(p))
        b.pending -= 1
        if b.pending == 0:
            b.rs_work = self._reduce_scatter(b)  # overlaps with the rest of backward
            if b.fsdp:
                # the staging buffer of the bucket BEFORE this one is given back once its reduce-scatter is done (it has had
                # a whole bucket of backward compute to finish)
                self._rs_inflight.append(b)
                self._retire_reduce_scatters(keep=1)
                g = self._bucket_group[id(b)]
                if all(self.buckets[bi].pending == 0 for bi in self.groups[g]):
                    self._release(g)  # nothing in backward reads these parameters any more

    def _check_views(self):
        """Every parameter must still be the view into its flat buffer that the optimizer updates."""
        for b in self.buckets:
            if not b.resident:
                continue
            base, es = b.flat_param.data_ptr(), b.flat_param.element_size()
            for p, o, n in zip(b.params, b.offsets, b.names):
                if p.data_ptr() != base + o * es or p.dtype != b.dtype:
                    raise RuntimeError(
                        f"parameter {n} no longer aliases the trainer's flat buffer (its .data was rebound: dtype cast, "
                        ".to(device), to_bfloat16_for_selected_params, ...): the optimizer would update memory the model does "
                        "not read.  Load / cast weights BEFORE constructing the Trainer, or write them in place "
                        "(p.data.copy_ / load_state_dict) and call sync_master_from_params().")  # fmt: skip

    @torch.no_grad()
    def sync_master_from_params(self):
        """Adopt weights written into the model after construction (load_state_dict, p.data.copy_, model_arithmetic): the
        f32 master copies (and the fsdp shards) are re-read from the parameters; Adam moments are kept.  Collective in fsdp
        mode only in the sense that every rank must call it."""
        self.wait_params()
        self._check_views()
        for b in self.buckets:
            sl = b.flat_param[b.lo : b.lo + b.shard]
            b.master.copy_(sl)
            if b.fsdp:
                b.param_shard.copy_(sl)
        if self.mode == "fsdp":
            self.release_params()

    # ------------------------------------------------------------------------------------------------- step
    @torch.no_grad()
    def step(self, lr: float):
        """Finish the gradient reduction, clip by the global norm, update the local shards, start the parameter all-gathers
        (zero2; waited for by the next forward, unit by unit).  Returns the global (pre-clip) gradient norm as a 1-element
        device tensor."""
        self.step_count += 1
        self._check_views()
        self._join_streams()  # gradients written by backward nodes on the second stream
        for b in self.buckets:
            if b.pending > 0:  # parameters that received no gradient this step contribute zeros
                self._ensure_grad(b)
                for p, o in zip(b.params, b.offsets):
                    if id(p) in b.stale and id(p) not in b.arrived:  # ... not what an earlier step left in their slice
                        b.flat_grad[o : o + p.numel()].zero_()
                        b.stale.discard(id(p))
                b.rs_work = self._reduce_scatter(b)
        for b in self.buckets:
            if b.rs_work is not None:
                self._wait(b.rs_work, "reduce_scatter_wait")
                b.rs_work = None
        self._rs_inflight.clear()
        coef = None
        if self.max_grad_norm is not None:
            self._sumsq.zero_()
            for b in self.buckets:
                self.ops.sumsq(b.grad_shard, self._sumsq)
            if self.world > 1:
                self._wait(dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=self.group, async_op=True), "norm_all_reduce")
            self.ops.clip_coef(self._sumsq, self.max_grad_norm, self._coef, self._norm)
            coef = self._coef
        for b in self.buckets:  # forward-use order: the first layers' parameters are complete first
            self._update_bucket(b, lr, coef)
            if not b.fsdp:
                b.ag_work = self._all_gather(b)  # not waited for here: pre_forward / wait_params do
        # Every gradient producer overwrites its parameter's whole slice, so the 7 GB of flat gradients are NOT cleared per
        # step: a slice is zeroed only if its producer accumulates into it (the embedding scatter: `_kai0_grad_accumulates`)
        # or, lazily in the next step(), if it holds an old gradient and no new one arrived.
        for b in self.buckets:
            for p, o in zip(b.params, b.offsets):
                if not b.grad_resident:  # fsdp staging buffer already given back: it comes back zero-filled (_ensure_grad)
                    break
                if id(p) in b.arrived:
                    if getattr(p, "_kai0_grad_accumulates", False):
                        b.flat_grad[o : o + p.numel()].zero_()
                        b.stale.discard(id(p))
                    else:
                        b.stale.add(id(p))
            b.pending = len(b.params)
            b.arrived.clear()
            if b.fsdp:
                # a bucket whose backward announced itself before its first gradient (model unit hooks) may drop its staging
                # buffer between steps from now on; one that did not (plain modules) keeps it: its producers write unannounced
                b.grad_freeable = b.grad_freeable or b.announced
                b.announced = False
                self._free_grad(b)
        self._in_backward = False
        if self.mode == "fsdp":
            self.release_params()
            self._issue_gather(0)  # the next forward starts with group 0
        return self._norm

    # ---- optimizer update of one bucket's shard ---------------------------------------------------------------
    _SPARSE_MIN_ROWS = 1024

    def _sparse_segments(self, b: _Bucket):
        """Row-sparse ranges of this rank's shard of `b`: [(first element in the shard, rows, row length, activity flags)] for every
        large 2-D parameter whose gradient producer scatters into a zeroed slice (`_kai0_grad_accumulates`: the embedding table —
        a step touches <= B x 200 of its 257152 rows).  Whole rows only; what a shard boundary cuts off is updated densely."""
        key = tuple(id(p) for p in b.params if getattr(p, "_kai0_grad_accumulates", False))
        if getattr(b, "_sparse_key", None) == key:
            return b._sparse_segs
        segs = []
        for p, o in zip(b.params, b.offsets):
            if not getattr(p, "_kai0_grad_accumulates", False) or p.dim() != 2 or p.shape[0] < self._SPARSE_MIN_ROWS:
                continue
            rl = int(p.shape[1])
            lo, hi = max(o, b.lo), min(o + p.numel(), b.lo + b.shard)
            r0, r1 = -(-(lo - o) // rl), (hi - o) // rl  # whole rows inside [lo, hi)
            if r1 > r0:
                first = o + r0 * rl - b.lo
                # a row is active once its moments may be nonzero: rebuilt from them, so a loaded checkpoint needs nothing extra
                mm = b.exp_avg[first : first + (r1 - r0) * rl].view(r1 - r0, rl)
                vv = b.exp_avg_sq[first : first + (r1 - r0) * rl].view(r1 - r0, rl)
                active = ((mm != 0).any(1) | (vv != 0).any(1)).to(torch.uint8)
                segs.append((first, r1 - r0, rl, active))
        b._sparse_key, b._sparse_segs = key, segs
        return segs

    def _update_bucket(self, b: _Bucket, lr: float, coef):
        kw = dict(lr=lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, wd=self.wd, step=self.step_count, clip_coef=coef)
        segs = []
        if self._sparse_rows and hasattr(self.ops, "adamw_rows"):
            from .optim import sparse_rows_ok

            if sparse_rows_ok(lr, self.wd):  # an idle row must be a fixed point of the update (1 - lr*wd rounds to 1)
                segs = self._sparse_segments(b)
        if not segs:
            # dense pass over the whole shard: rows first touched here get nonzero moments WITHOUT their activity flag being set
            # (only kai0_adamw_rows sets flags), so the cached flags are void — rebuilt from the moments at the next sparse step
            # (a schedule whose lr * wd crosses the 2^-25 threshold goes sparse -> dense -> sparse)
            b._sparse_key = None
        cur = 0
        for first, rows, rl, active in segs:
            if first > cur:
                self.ops.adamw(b.master[cur:first], b.exp_avg[cur:first], b.exp_avg_sq[cur:first], b.grad_shard[cur:first],
                               b.param_shard[cur:first], **kw)  # fmt: skip
            end = first + rows * rl
            self.ops.adamw_rows(b.master[first:end], b.exp_avg[first:end], b.exp_avg_sq[first:end], b.grad_shard[first:end],
                                b.param_shard[first:end], rl, active, **kw)  # fmt: skip
            cur = end
        if cur < b.shard:
            self.ops.adamw(b.master[cur:], b.exp_avg[cur:], b.exp_avg_sq[cur:], b.grad_shard[cur:], b.param_shard[cur:], **kw)

    # ------------------------------------------------------------------------------------------ checkpointing
    @torch.no_grad()
    def state_dict(self, param_order=None):
        """World-size independent optimizer state, shaped like `torch.optim.AdamW.state_dict()` (the reference's
        `optimizer.pt`, train_pytorch.py:170-180): state[i] = {step, exp_avg, exp_avg_sq} for parameter i of `param_order`
        (names in `model.named_parameters()` order — the order torch indexes them in; default: this engine's bucket order),
        plus this engine's f32 `master` copy and the name list.  Parameters the engine does not train have no entry, as in
        torch.  COLLECTIVE: every rank calls it; rank 0 gets the dictionary (CPU tensors), the others None."""
        by_name = {}
        for b in self.buckets:
            fulls = {}
            for key in ("master", "exp_avg", "exp_avg_sq"):
                sh = getattr(b, key)
                if self.collectives:
                    full = torch.empty(b.numel, dtype=F32, device=self.device)
                    dist.all_gather_into_tensor(full, sh, group=self.group)
                else:
                    full = sh
                fulls[key] = full
            if self.rank == 0:
                for p, o, n in zip(b.params, b.offsets, b.names):
                    by_name[n] = {"step": torch.tensor(float(self.step_count)),
                                  **{k: v[o : o + p.numel()].view(p.shape).cpu().clone() for k, v in fulls.items()}}  # fmt: skip
            del fulls
        if self.rank != 0:
            return None
        order = list(param_order) if param_order is not None else [n for b in self.buckets for n in b.names]
        missing = set(by_name) - set(order)
        if missing:
            raise KeyError(f"param_order lacks trained parameters: {sorted(missing)[:3]} ...")
        return {"state": {i: by_name[n] for i, n in enumerate(order) if n in by_name},
                "param_groups": [{"betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "params": list(range(len(order)))}],
                "param_names": order, "step": self.step_count}  # fmt: skip

    @torch.no_grad()
    def load_state_dict(self, sd, param_order=None):
        """Every rank loads the same dictionary and keeps its slices — any world size, any bucket layout.  Entries are matched
        by parameter name (`param_names` in the file, else `param_order`); a plain torch AdamW state dict (the reference's
        optimizer.pt: index = position in model.parameters(), bf16 moments, no master, no entry for parameters that never
        received a gradient) therefore loads too: moments are widened to f32, master copies missing from the file are taken
        from the current parameters, parameters without an entry start from zero moments.  The model weights must already be in
        place (load them first): the fsdp shards are re-read from them."""
        pnames = sd.get("param_names", param_order)
        if pnames is None:
            raise ValueError("optimizer state without `param_names` needs param_order (names in model.named_parameters() order)")
        state = sd["state"]
        by_name = {pnames[i]: ent for i, ent in state.items()}
        steps = [float(s["step"]) for s in state.values() if "step" in s]
        self.step_count = int(sd.get("step", max(steps) if steps else 0))
        # torch.optim.AdamW only holds state for parameters that have received a gradient: in the reference's optimizer.pt the
        # last layer's prefix o_proj / MLP / post-attention norm and language_model.norm (dead values) have no entry.
        # A missing entry = "never updated": zero moments, master copy = the current parameter.  Master copies are adopted per
        # parameter — an entry without `master` (torch-shaped file) does not discard the f32 masters other entries carry.
        from_params: list[tuple[_Bucket, int, int]] = []  # (bucket, lo, hi) slices whose master comes from the parameters
        absent = []
        for b in self.buckets:
            for p, o, n in zip(b.params, b.offsets, b.names):
                ent = by_name.get(n)
                lo, hi = max(o, b.lo), min(o + p.numel(), b.lo + b.shard)
                if ent is None:
                    absent.append(n)
                    ent = {}
                for key in ("exp_avg", "exp_avg_sq", "master"):
                    if key not in ent:
                        if lo < hi:
                            if key == "master":
                                from_params.append((b, lo, hi))
                            else:
                                getattr(b, key)[lo - b.lo : hi - b.lo].zero_()
                        continue
                    if ent[key].numel() != p.numel():
                        raise ValueError(f"optimizer state of {n}: {tuple(ent[key].shape)} vs parameter {tuple(p.shape)}")
                    if lo < hi:
                        getattr(b, key)[lo - b.lo : hi - b.lo].copy_(ent[key].reshape(-1)[lo - o : hi - o].to(F32))
        if absent:
            import logging

            logging.getLogger("kai0_amd").warning(
                "optimizer state has no entry for %d parameter(s) (never updated when it was written; zero moments assumed): %s%s",
                len(absent), ", ".join(absent[:4]), " ..." if len(absent) > 4 else "")  # fmt: skip
        for b in self.buckets:
            b._sparse_key = None  # the row-activity flags are rebuilt from the loaded moments at the next step
        self.wait_params()
        self._check_views()
        for b, lo, hi in from_params:  # the alignment padding between parameters keeps whatever the master held (zeros)
            b.master[lo - b.lo : hi - b.lo].copy_(b.flat_param[lo:hi])
        for b in self.buckets:
            if b.fsdp:
                b.param_shard.copy_(b.flat_param[b.lo : b.lo + b.shard])
        if self.mode == "fsdp":
            self.release_params()

    def optimizer_state_bytes(self) -> int:
        return sum(3 * 4 * b.shard for b in self.buckets)
