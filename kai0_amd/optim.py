"""Fused AdamW + global-norm clipping on the HIP kernels, and the reference's LR schedule.

Semantics follow scripts/train_pytorch.py:469-491,557-561 (torch.optim.AdamW + clip_grad_norm_) and
src/openpi/training/optimizer.py:15-85 (warmup + cosine).  Difference, on purpose and documented in DESIGN.md:
the reference's PyTorch trainer updates bf16 parameters in bf16 with bf16 Adam moments; here every parameter has
an f32 master copy and f32 moments (the JAX trainer's precision), and the bf16 model copy is re-rounded from the
master each step.  16 B/param of optimizer state is HBM-bound work, fused into one pass per tensor.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from .ops import BF16, F32, _stream


def lr_schedule(step: int, *, warmup_steps: int, peak_lr: float, decay_steps: int, end_lr: float) -> float:
    """train_pytorch.py:483-491 (matches optax warmup_cosine_decay_schedule with init = peak/(warmup+1))."""
    if step < warmup_steps:
        init_lr = peak_lr / (warmup_steps + 1)
        return init_lr + (peak_lr - init_lr) * step / warmup_steps
    progress = min(1.0, (step - warmup_steps) / max(1, decay_steps - warmup_steps))
    cos = 0.5 * (1 + np.cos(np.pi * progress))
    return float(end_lr + (peak_lr - end_lr) * cos)


WEIGHT_UPDATES = [0]  # bumped by every parameter update made through the HIP kernels (see infer.InferenceEngine._fingerprint)


def adamw_step_(master, m, v, grad, param, *, lr, beta1, beta2, eps, wd, step: int, clip_coef=None):
    """One fused AdamW update of a flat f32 shard (kai0_adamw)."""
    WEIGHT_UPDATES[0] += 1
    n = master.numel()
    bc1 = 1.0 - beta1**step
    bc2 = 1.0 - beta2**step
    _lib.call("kai0_adamw", master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), int(grad.dtype == F32),
              param.data_ptr(), int(param.dtype == F32), n, lr, beta1, beta2, eps, wd, bc1, bc2,
              None if clip_coef is None else clip_coef.data_ptr(), _stream())  # fmt: skip


def sparse_rows_ok(lr: float, wd: float) -> bool:
    """True if an idle row (zero moments, zero gradient) is a fixed point of the update: 1 - lr*wd rounds to 1 in f32."""
    return bool(np.float32(1.0) - np.float32(lr) * np.float32(wd) == np.float32(1.0))


def adamw_rows_step_(master, m, v, grad, param, row_len: int, row_active, *, lr, beta1, beta2, eps, wd, step: int, clip_coef=None):
    """kai0_adamw_rows: the same update on a flat [rows * row_len] range whose gradient is zero in most rows (embedding table);
    bit-identical to adamw_step_, idle rows cost their gradient read only.  row_active: uint8 [rows], persistent."""
    WEIGHT_UPDATES[0] += 1
    n = master.numel()
    assert n % row_len == 0 and row_active.numel() == n // row_len and row_active.dtype == torch.uint8
    bc1 = 1.0 - beta1**step
    bc2 = 1.0 - beta2**step
    _lib.call("kai0_adamw_rows", master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), int(grad.dtype == F32),
              param.data_ptr(), int(param.dtype == F32), n // row_len, row_len, row_active.data_ptr(), lr, beta1, beta2, eps, wd,
              bc1, bc2, None if clip_coef is None else clip_coef.data_ptr(), _stream())  # fmt: skip


_SUMSQ_SCRATCH: dict = {}


def sumsq_accumulate_(grad, out):
    """out[0] += sum(grad^2), deterministic (block partials + ordered finish)."""
    key = (grad.device.index, _stream())
    scratch = _SUMSQ_SCRATCH.get(key)
    if scratch is None:
        scratch = _SUMSQ_SCRATCH[key] = torch.empty(4096, dtype=F32, device=grad.device)
    _lib.call("kai0_sumsq", grad.data_ptr(), int(grad.dtype == F32), grad.numel(), out.data_ptr(), scratch.data_ptr(), _stream())


def sum_chunks_(src, chunks: int, out):
    """out[i] = sum_j src[j * out.numel() + i] in f32, one rounding (the local half of the all-pairs reduce-scatter)."""
    n = out.numel()
    assert src.dtype == out.dtype and src.numel() == chunks * n and src.is_contiguous() and out.is_contiguous()
    _lib.call("kai0_sum_chunks", src.data_ptr(), int(src.dtype == F32), int(chunks), n, n, out.data_ptr(), _stream())


def clip_coef_(sumsq, max_norm: float, coef, norm_out):
    _lib.call("kai0_clip_coef", sumsq.data_ptr(), float(max_norm), coef.data_ptr(), norm_out.data_ptr(), _stream())


class FusedAdamW:
    """torch.optim.AdamW-shaped optimizer (param_groups with "lr", step(), zero_grad(), state_dict()).

    step() = [sum of squared grads over all params -> clip coefficient on device] -> fused AdamW per tensor.
    No host synchronisation: the clip coefficient stays in device memory and is read by the update kernel."""

    def __init__(self, params, lr=2.5e-5, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-10, max_grad_norm=1.0):
        self.params = [p for p in params if p.requires_grad]
        # tied parameters appear once
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "params": self.params}]
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.state = {}
        dev = self.params[0].device
        for p in self.params:
            self.state[p] = {
                "master": p.detach().to(F32).clone().contiguous(),
                "exp_avg": torch.zeros(p.shape, dtype=F32, device=p.device),
                "exp_avg_sq": torch.zeros(p.shape, dtype=F32, device=p.device),
            }
        self._sumsq = torch.zeros(1, dtype=F32, device=dev)
        self._coef = torch.ones(1, dtype=F32, device=dev)
        self._norm = torch.zeros(1, dtype=F32, device=dev)

    def master_params(self):
        return [self.state[p]["master"] for p in self.params]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        """Returns the (pre-clip) global grad norm as a 1-element device tensor (no sync)."""
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        self.step_count += 1
        live = [p for p in self.params if p.grad is not None]
        coef = None
        if self.max_grad_norm is not None:
            self._sumsq.zero_()
            for p in live:
                sumsq_accumulate_(p.grad.contiguous(), self._sumsq)
            clip_coef_(self._sumsq, self.max_grad_norm, self._coef, self._norm)
            coef = self._coef
        for p in live:
            st = self.state[p]
            adamw_step_(st["master"], st["exp_avg"], st["exp_avg_sq"], p.grad.contiguous(), p.data, lr=lr, beta1=b1,
                        beta2=b2, eps=eps, wd=wd, step=self.step_count, clip_coef=coef)  # fmt: skip
        return self._norm

    def state_dict(self):
        return {
            "step": self.step_count,
            "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}],
            "state": [{k: v for k, v in self.state[p].items()} for p in self.params],
        }

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v
        for p, st in zip(self.params, sd["state"], strict=True):
            for k in ("master", "exp_avg", "exp_avg_sq"):
                self.state[p][k].copy_(st[k])
