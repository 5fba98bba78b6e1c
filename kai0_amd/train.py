"""Training driver mirroring scripts/train_pytorch.py:309-633: `Trainer` (one step) and `train_loop(config)` (the loop a
`TrainConfig` describes: data loader -> device feed -> steps -> logging -> checkpoints -> resume), with
`python -m kai0_amd.train <config-name> [--field value ...]` as the command line (wandb / tqdm shell left out).

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

import dataclasses
import logging
import os
import shutil
import time

# RCCL on this driver needs dmabuf IPC (hipIpcGetMemHandle fails otherwise): in the environment before HIP initialises, whoever
# launches the trainer (torchrun -m kai0_amd.train, bench.py, a notebook)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from .optim import lr_schedule
from .sharded import ShardedDataParallel


class Trainer:
    def __init__(self, model, *, world_size: int = 1, rank: int = 0, group=None, peak_lr=2.5e-5, warmup_steps=1000,
                 decay_steps=30000, end_lr=2.5e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-10, clip_norm=1.0,
                 shard_ops=None, bucket_bytes: int | None = None, mode: str | None = None, prefetch: int | None = None):  # fmt: skip
        """`bucket_bytes` / `prefetch` default per mode: zero2 512 MB buckets (few, large collectives; nothing waits on them
        inside forward / backward); fsdp 256 MB (~ one joint Gemma-2B + expert layer, SURVEY.md §8e's unit) gathered two
        buckets ahead, so that a bucket's gather has two layers of compute to hide behind and ~0.75 GB of full parameters
        are live at a time."""
        self.model = model
        self.world, self.rank = world_size, rank
        self.sched = dict(warmup_steps=warmup_steps, peak_lr=peak_lr, decay_steps=decay_steps, end_lr=end_lr)
        # `gemma_expert.lm_head` is never used by the forward: it can never receive a gradient,
        # so it is left out of the optimizer (torch.optim.AdamW skips grad-less parameters as well).
        dead = model.paligemma_with_expert.gemma_expert.lm_head.weight
        self._param_order = [n for n, _ in model.named_parameters()]  # torch's optimizer indexing (ties once)
        named = [(n, p) for n, p in model.named_parameters() if p is not dead]
        units = model.sharding_units() if hasattr(model, "sharding_units") else None
        # north_star's partition — optimizer state, gradients AND parameters sharded (fsdp) — is the default whenever there are peers;
        # KAI0_SHARD_MODE=zero2 / mode="zero2" keeps the parameters replicated (optimizer state and gradients sharded only)
        mode = mode or os.environ.get("KAI0_SHARD_MODE") or ("fsdp" if world_size > 1 else "zero2")
        if bucket_bytes is None:
            bucket_bytes = int(os.environ.get("KAI0_BUCKET_MB", "256" if mode == "fsdp" else "512")) << 20
        if prefetch is None:
            prefetch = int(os.environ.get("KAI0_FSDP_PREFETCH", "2" if mode == "fsdp" else "1"))
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
        self.engine.begin_step()
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
    def save_checkpoint(self, checkpoint_dir: str, *, data_config=None, config=None, norm_stats=None, asset_id=None) -> str:
        """train_pytorch.py:149-194: rank 0 writes model.safetensors, optimizer.pt (gathered, world-size independent),
        metadata.pt (global_step, config, timestamp) and the norm stats the run was trained with under
        `assets/<asset_id>` — where `policy.create_trained_policy` looks for them (policy_config.py:64-69); every rank takes
        part in the gathers.  `data_config` (DataConfig: norm_stats + asset_id) or the explicit pair name the stats."""
        from . import normalize as _normalize
        from .checkpoint import save_model_safetensors

        if data_config is not None:
            norm_stats = norm_stats if norm_stats is not None else data_config.norm_stats
            asset_id = asset_id if asset_id is not None else data_config.asset_id
        step = self.global_step
        final = os.path.join(checkpoint_dir, f"{step}")
        tmp = os.path.join(checkpoint_dir, f"tmp_{step}")
        self.engine.wait_params()
        opt_sd = self.engine.state_dict(self._param_order)
        rng = _gather_rng_states(self.world, self.engine.device)
        if self.rank == 0:
            if os.path.exists(tmp):
                shutil.rmtree(tmp)
            os.makedirs(tmp, exist_ok=True)
            save_model_safetensors(self.model, os.path.join(tmp, "model.safetensors"))
            torch.save(opt_sd, os.path.join(tmp, "optimizer.pt"))
            # (not in the reference's metadata) the ranks' RNG states: with them and `skip_batches` a resumed run draws the same
            # noise / time / augmentation and sees the same batches as the run that was interrupted
            meta = {"global_step": step, "world_size": self.world, "timestamp": time.time(), "rng_state": rng}
            if config is not None:
                meta["config"] = _plain(dataclasses.asdict(config) if dataclasses.is_dataclass(config) else config)
            torch.save(meta, os.path.join(tmp, "metadata.pt"))
            if norm_stats is not None and asset_id is not None:
                _normalize.save(os.path.join(tmp, "assets", asset_id), norm_stats)
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

        steps = [int(d) for d in os.listdir(checkpoint_dir) if d.isdigit() and os.path.isdir(os.path.join(checkpoint_dir, d))]
        if not steps:
            raise FileNotFoundError(f"No checkpoints found in {checkpoint_dir}")
        step = max(steps)
        d = os.path.join(checkpoint_dir, str(step))
        self.engine.wait_params()
        load_model_safetensors(self.model, os.path.join(d, "model.safetensors"))  # in place: the flat buffers keep their views
        # tensors, numbers, lists and strings only: no pickle execution from a checkpoint directory
        opt = torch.load(os.path.join(d, "optimizer.pt"), map_location="cpu", weights_only=True)
        self.engine.load_state_dict(opt, self._param_order)
        # metadata.pt written by the REFERENCE holds dataclasses.asdict(config) (train_pytorch.py:172-177): model / transform classes,
        # nnx.Nothing() ... which the weights-only unpickler refuses.  Only global_step (and this trainer's rng_state) are needed:
        # fall back to the directory name, never to weights_only=False.
        import pickle

        try:
            meta = torch.load(os.path.join(d, "metadata.pt"), map_location="cpu", weights_only=True)
        except (pickle.UnpicklingError, RuntimeError, AttributeError, ModuleNotFoundError, FileNotFoundError) as e:
            logging.getLogger(__name__).warning("metadata.pt of %s not loadable weights-only (%s): step taken from the directory name, "
                                                "no RNG state", d, type(e).__name__)  # fmt: skip
            meta = {}
        self.global_step = int(meta.get("global_step", step))
        self.resumed_rng_state = meta.get("rng_state")  # train_loop restores it when the world size is the one that wrote it
        if getattr(self.model, "_engine", None) is not None:
            self.model.invalidate_inference_engine()
        return self.global_step


def _gather_rng_states(world: int, device) -> list:
    """[{cpu, cuda}] per rank (uint8 tensors); collective when world > 1."""
    mine = {"cpu": torch.get_rng_state()}
    if torch.device(device).type == "cuda":
        mine["cuda"] = torch.cuda.get_rng_state(device)
    if world <= 1:
        return [mine]
    box = [None] * world
    torch.distributed.all_gather_object(box, mine)
    return box


def _restore_rng_state(states, world: int, rank: int, device) -> bool:
    if not states or len(states) != world:
        return False
    st = states[rank]
    torch.set_rng_state(st["cpu"])
    if "cuda" in st and torch.device(device).type == "cuda":
        torch.cuda.set_rng_state(st["cuda"], device)
    return True


def _plain(x):
    """config -> containers of str / int / float / bool / None only, so that metadata.pt loads with weights_only=True"""
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, (str, int, float, bool)) or x is None:
        return x
    return repr(x)


# ------------------------------------------------------------------------------------------------ the training loop
def _setup_distributed():
    """train_pytorch.py:86-108: one process per GPU under torchrun (env:// rendezvous); backend nccl (= RCCL over xGMI)."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    device = torch.device(f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        if device.type == "cuda":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
        else:
            dist.init_process_group(backend="gloo", init_method="env://")
    rank = dist.get_rank() if dist.is_initialized() else 0
    return world, rank, device


def build_model(config, device):
    """train_pytorch.py:399-458: `config.model` in `pytorch_training_precision`, on `device`, with `pytorch_weight_path`'s
    `model.safetensors` loaded (strict unless an AdvantageEstimator is initialised from a policy checkpoint)."""
    from .checkpoint import load_model_safetensors

    model_cfg = dataclasses.replace(config.model, dtype=config.pytorch_training_precision)
    with torch.device(device):
        model = model_cfg._model_class()(model_cfg)
    if hasattr(model, "gradient_checkpointing_enable") and os.environ.get("KAI0_REMAT", "0") == "1":
        model.gradient_checkpointing_enable()  # the reference force-enables it; 288 GB of HBM make it optional here
    if hasattr(model, "trim_prompt_padding") and os.environ.get("KAI0_TRIM_PROMPT", "1") != "0":
        # the prompt slots no sample of the batch uses are not computed (kai0's task prompts fill 10-40 of the 200 slots; padded slots are
        # invisible keys and unread rows: loss and gradients unchanged beyond summation order, tests/test_model_gpu.py).  bench.py
        # measures with it off — every one of the 200 slots computed, as the reference does — and reports the trimmed rate beside it.
        model.trim_prompt_padding = True
    if config.pytorch_weight_path is not None:
        load_model_safetensors(model, os.path.join(config.pytorch_weight_path, "model.safetensors"),
                               strict=not config.advantage_estimator)  # fmt: skip
        logging.info(f"Loaded PyTorch weights from {config.pytorch_weight_path}")
    return model


def train_loop(config, *, device=None, shard_ops=None, model=None, log=None):
    """scripts/train_pytorch.py:309-633 over this framework's pieces.

    TrainConfig -> (resume | overwrite) the experiment's checkpoint directory -> `create_data_loader(config)` (global batch
    `config.batch_size`, every rank `batch_size // world` of its DistributedSampler shard) behind a `DeviceFeeder` -> model
    (`pytorch_weight_path`) -> `Trainer` from `config.lr_schedule` / `config.optimizer` -> for every batch: one train step; mean
    loss / lr / grad-norm logged every `log_interval` steps (the step itself never synchronises: the scalars of an interval are
    read when it is logged); checkpoint every `save_interval` steps and at the last step, with the norm stats; `resume`
    continues from the newest checkpoint on any number of GPUs.  Returns the list of logged records (rank 0)."""
    from .data_loader import DeviceFeeder
    from .lerobot_dataset import create_data_loader

    world, rank, dev = _setup_distributed()
    device = torch.device(device) if device is not None else dev
    is_main = rank == 0
    say = log or (logging.info if is_main else (lambda *_: None))
    torch.manual_seed(config.seed + rank)  # train_pytorch.py:118-122

    ckpt_dir = config.checkpoint_dir
    resuming = False
    if config.resume:
        if not ckpt_dir.exists():
            raise FileNotFoundError(f"Experiment checkpoint directory {ckpt_dir} does not exist for resume")
        if not any(d.name.isdigit() and d.is_dir() for d in ckpt_dir.iterdir()):
            raise FileNotFoundError(f"No valid checkpoints found in {ckpt_dir} for resume")
        resuming = True
    elif is_main:
        if config.overwrite and ckpt_dir.exists():
            shutil.rmtree(ckpt_dir)
            say(f"Overwriting checkpoint directory: {ckpt_dir}")
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    if world > 1:
        torch.distributed.barrier()

    loader = create_data_loader(config, shuffle=True, skip_norm_stats=config.skip_norm_stats)
    data_config = loader.data_config()
    if model is None:
        model = build_model(config, device)
    model.train()
    sch, opt = config.lr_schedule, config.optimizer
    trainer = Trainer(model, world_size=world, rank=rank, peak_lr=sch.peak_lr, warmup_steps=sch.warmup_steps,
                      decay_steps=sch.decay_steps, end_lr=sch.decay_lr, betas=(opt.b1, opt.b2), eps=opt.eps,
                      weight_decay=opt.weight_decay, clip_norm=opt.clip_gradient_norm, shard_ops=shard_ops)  # fmt: skip
    if resuming:
        step = trainer.load_checkpoint(str(ckpt_dir))
        exact = _restore_rng_state(getattr(trainer, "resumed_rng_state", None), world, rank, device)
        loader.skip_batches(step)  # every step consumed one batch of the seeded stream
        say(f"Resumed training from step {step}" + ("" if exact else " (no RNG state for this world size: noise / augmentation restart from the seed)"))
    say(f"world_size={world} batch_size={config.batch_size} (per GPU {config.batch_size // world}) num_train_steps={config.num_train_steps} "
        f"mode={trainer.engine.mode} lr: warmup={sch.warmup_steps} peak={sch.peak_lr:.2e} decay_steps={sch.decay_steps} end={sch.decay_lr:.2e}")  # fmt: skip

    records, pending, t0 = [], [], time.time()
    while trainer.global_step < config.num_train_steps:
        produced = False
        for observation, actions in DeviceFeeder(loader, device):
            if trainer.global_step >= config.num_train_steps:
                break
            produced = True
            step = trainer.global_step
            lr = trainer.lr()
            loss = trainer.train_step(observation, actions.to(torch.float32))
            pending.append((loss, lr, trainer.last_grad_norm.clone()))
            if step % config.log_interval == 0:
                vals = [(float(l), r, float(g)) for l, r, g in pending]  # the interval's only host synchronisation
                rec = {"step": step, "loss": sum(v[0] for v in vals) / len(vals), "learning_rate": sum(v[1] for v in vals) / len(vals),
                       "grad_norm": sum(v[2] for v in vals) / len(vals), "time_per_step": (time.time() - t0) / len(vals)}  # fmt: skip
                if is_main:
                    records.append(rec)
                    say(f"step={step} loss={rec['loss']:.4f} lr={rec['learning_rate']:.2e} grad_norm={rec['grad_norm']:.2f} "
                        f"time/step={rec['time_per_step']:.3f}s")  # fmt: skip
                pending, t0 = [], time.time()
            gs = trainer.global_step
            if (gs % config.save_interval == 0 and gs > 0) or gs == config.num_train_steps:
                path = trainer.save_checkpoint(str(ckpt_dir), data_config=data_config, config=config)
                say(f"Saved checkpoint at step {gs} -> {path}")
        if not produced:
            raise RuntimeError("the data loader produced no batch")
    if world > 1:
        torch.distributed.barrier()
    return records


def main(argv=None):
    from .training_config import cli

    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    config = cli(argv)
    train_loop(config)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
