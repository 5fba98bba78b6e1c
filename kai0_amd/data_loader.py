"""Training data loader -> device feed: everything between a map-style dataset of per-sample dicts and the
`(Observation, actions)` batches that `Trainer.train_step` consumes, resident in HBM before the step that uses them starts.

Host side, same behaviour as `src/openpi/training/data_loader.py` (torch path):
  TransformedDataset :54-63, FakeDataset :100-128 (same value ranges per dtype; the reference draws from jax.random keyed by the
  index, this one from numpy's PCG64 keyed by the index — jax is not available, so samples are equally distributed, not
  bit-identical), `_collate_fn` :538-542 (stack leaves into numpy batches), TorchDataLoader :448-535 (torch DataLoader with
  drop_last, a seeded generator, spawn workers; `num_batches` loops over the dataset until that many batches were produced,
  None = forever), the DistributedSampler hook-up of `create_torch_data_loader` :376-404 (one process per GPU: the global batch
  is divided by the world size, the sampler shards the indices and does the shuffling), DataLoaderImpl :597-607
  (`Observation.from_dict(batch), batch["actions"]`).
What the LeRobot parquet / AV1 video source (`create_torch_dataset` :131-152) would plug into is the `dataset` argument: any
object with `__len__` and `__getitem__` returning the per-sample dict; `lerobot` itself is not installed here.

MI355X side (not in the reference, whose loop copies each batch synchronously inside the step, train_pytorch.py:520-538):
`DeviceFeeder` keeps `depth` batches in flight — a background thread collates into pinned host memory and enqueues the H2D
copies on a side HIP stream, the training stream only waits on the copy's event — so PCIe never sits on the step's critical
path (one pi0.5 batch of 32 is 3 x 32 x 224 x 224 x 3 floats = 58 MB: ~1 ms of a 558 ms step even un-overlapped, but it
must not serialise with the dataset's own decode time)."""

from __future__ import annotations

import logging
import multiprocessing
import queue
import threading
from collections.abc import Mapping, Sequence

import numpy as np
import torch

from . import transforms as _transforms
from .preprocessing import IMAGE_KEYS, Observation


def _tree_map(fn, *trees):
    t0 = trees[0]
    if isinstance(t0, Mapping):
        return {k: _tree_map(fn, *(t[k] for t in trees)) for k in t0}
    return fn(*trees)


class TransformedDataset:
    def __init__(self, dataset, transforms: Sequence):
        self._dataset = dataset
        self._transform = _transforms.compose(transforms)

    def __getitem__(self, index):
        return self._transform(self._dataset[index])

    def __len__(self) -> int:
        return len(self._dataset)


class FakeDataset:
    """Random samples with the model's input shapes: float32 leaves U(-1, 1), int32 leaves U{0..2047}, everything else
    (the boolean masks) zeros — what the reference's FakeDataset produces per dtype.  One deliberate difference: images are
    uint8 HWC by default (`uint8_images`), which `Observation.from_dict` turns into the float NCHW [-1, 1] tensors the torch
    model needs; the reference's float32 NHWC fake images pass through `from_dict` untouched and cannot feed its own
    Conv2d.  `valid_masks=True` sets the masks to True so that the fake batch exercises the whole path."""

    def __init__(self, model_config, num_samples: int, image_size: int | None = None, uint8_images: bool = True,
                 valid_masks: bool = True):  # fmt: skip
        self._n = num_samples
        hw = image_size or getattr(getattr(model_config, "siglip", None), "image_size", 224)
        c = model_config
        img = ((hw, hw, 3), np.uint8 if uint8_images else np.float32)
        self._valid = valid_masks
        self._spec = {"image": {k: img for k in IMAGE_KEYS}, "image_mask": {k: ((), np.bool_) for k in IMAGE_KEYS},
                      "state": ((c.action_dim,), np.float32), "tokenized_prompt": ((c.max_token_len,), np.int32),
                      "tokenized_prompt_mask": ((c.max_token_len,), np.bool_),
                      "actions": ((c.action_horizon, c.action_dim), np.float32)}  # fmt: skip

    def __getitem__(self, index) -> dict:
        rng = np.random.Generator(np.random.PCG64(int(index.__index__())))

        def make(spec):
            shape, dtype = spec
            if dtype == np.float32:
                return rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
            if dtype == np.int32:
                return rng.integers(0, 2048, size=shape, dtype=np.int32)
            if dtype == np.uint8:
                return rng.integers(0, 256, size=shape, dtype=np.uint8)
            return np.full(shape, self._valid, dtype=dtype)

        return _tree_map(make, self._spec)

    def __len__(self) -> int:
        return self._n


def _collate_fn(items):
    """Per-sample dicts -> one dict of stacked numpy arrays."""
    return _tree_map(lambda *xs: np.stack([np.asarray(x) for x in xs], axis=0), *items)


class _SkippingBatchSampler(torch.utils.data.Sampler):
    """A BatchSampler whose next iterations drop their first `skip` index batches WITHOUT loading them: the sampler's (seeded)
    index stream is advanced, `dataset.__getitem__` is never called for a dropped batch and no global RNG is consumed."""

    def __init__(self, inner):
        self.inner, self.skip, self.skipped = inner, 0, 0

    def __len__(self) -> int:
        return len(self.inner)

    def __iter__(self):
        it = iter(self.inner)
        while self.skip > 0:
            try:
                next(it)
            except StopIteration:  # a whole pass dropped: the caller starts the next one (same generator draws as a real pass)
                return
            self.skip -= 1
            self.skipped += 1
        yield from it


class TorchDataLoader:
    def __init__(self, dataset, local_batch_size: int, *, shuffle: bool = False, sampler=None, num_batches: int | None = None,
                 num_workers: int = 0, seed: int = 0):  # fmt: skip
        if len(dataset) < local_batch_size:
            raise ValueError(f"Local batch size ({local_batch_size}) is larger than the dataset size ({len(dataset)}).")
        self._num_batches = num_batches
        self._skip = 0
        generator = torch.Generator()
        generator.manual_seed(seed)
        # the samplers torch.utils.data.DataLoader(batch_size=, shuffle=, sampler=, drop_last=True, generator=) builds itself, made
        # explicit so that the batch sampler can be advanced without loading (skip_batches); same index stream for the same seed
        if sampler is None:
            sampler = (torch.utils.data.RandomSampler(dataset, generator=generator) if shuffle
                       else torch.utils.data.SequentialSampler(dataset))  # fmt: skip
        self._batch_sampler = _SkippingBatchSampler(torch.utils.data.BatchSampler(sampler, local_batch_size, drop_last=True))
        self._data_loader = torch.utils.data.DataLoader(
            dataset, batch_sampler=self._batch_sampler, num_workers=num_workers,
            multiprocessing_context=multiprocessing.get_context("spawn") if num_workers > 0 else None,
            persistent_workers=num_workers > 0, collate_fn=_collate_fn, generator=generator)  # fmt: skip

    @property
    def torch_loader(self) -> torch.utils.data.DataLoader:
        return self._data_loader

    def skip_batches(self, n: int) -> None:
        """Resume support: the next iteration drops its first `n` batches, i.e. continues the (seeded, hence reproducible) batch
        stream where a run that had consumed `n` batches stopped.  Dropped at the INDEX level: the batch sampler is advanced, no
        sample is loaded, decoded or transformed and no global RNG is consumed (resuming at step 20k costs 20k index batches,
        milliseconds, not 20k global batches of video decoding)."""
        self._skip = max(0, int(n))

    def __iter__(self):
        produced = 0
        skip, self._skip = self._skip, 0
        if skip:
            logging.getLogger(__name__).info("data loader: skipping %d batches at the index level (resume)", skip)
        self._batch_sampler.skip = skip
        while True:  # a new pass over the dataset whenever the previous one is exhausted
            for batch in self._data_loader:
                if self._num_batches is not None and produced >= self._num_batches:
                    return
                produced += 1
                yield _tree_map(torch.as_tensor, batch)
            if self._num_batches is not None and produced >= self._num_batches:
                return


class DataLoaderImpl:
    """Batches as the trainer wants them: `(Observation, actions)`."""

    def __init__(self, data_loader, data_config=None):
        self._data_loader, self._data_config = data_loader, data_config

    def data_config(self):
        return self._data_config

    def skip_batches(self, n: int) -> None:
        self._data_loader.skip_batches(n)

    def __iter__(self):
        for batch in self._data_loader:
            yield Observation.from_dict(batch), batch["actions"]


def create_torch_data_loader(dataset, batch_size: int, *, transforms: Sequence = (), shuffle: bool = False,
                             num_batches: int | None = None, num_workers: int = 0, seed: int = 0, data_config=None) -> DataLoaderImpl:  # fmt: skip
    """`batch_size` is the GLOBAL batch.  With torch.distributed initialised (one process per GPU) each rank gets
    batch_size // world_size samples per step from its DistributedSampler shard (which then owns the shuffling)."""
    if transforms:
        dataset = TransformedDataset(dataset, transforms)
    sampler, local = None, batch_size
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle,
                                                                  drop_last=True)  # fmt: skip
        local = batch_size // world
    loader = TorchDataLoader(dataset, local, shuffle=(sampler is None and shuffle), sampler=sampler, num_batches=num_batches,
                             num_workers=num_workers, seed=seed)  # fmt: skip
    return DataLoaderImpl(loader, data_config)


# ------------------------------------------------------------------------------------------------------------ device feed
def _to_device_tree(tree, device, stream, pin: bool):
    def move(x):
        if not isinstance(x, torch.Tensor):
            return x
        if pin and x.device.type == "cpu" and not x.is_pinned():
            x = x.pin_memory()
        return x.to(device, non_blocking=pin)

    if isinstance(tree, Observation):
        kw = {}
        for f in ("images", "image_masks", "state", "tokenized_prompt", "tokenized_prompt_mask", "token_ar_mask", "token_loss_mask",
                  "episode_index", "frame_index", "progress", "episode_length", "image_original"):  # fmt: skip
            v = getattr(tree, f, None)
            kw[f] = _tree_map(move, v) if isinstance(v, Mapping) else (move(v) if v is not None else None)
        return Observation(**kw)
    if isinstance(tree, Mapping):
        return _tree_map(move, tree)
    return move(tree)


class DeviceFeeder:
    """Iterate `(Observation, actions)` with every tensor already on `device`.  `depth` batches are prepared ahead by a
    background thread: host collation / decode and the H2D copies issued on a side stream; the consumer's stream waits on the
    copy's event only (no host synchronisation, no copy on the training stream).
    The copies are plain blocking copies from pageable memory — the worker thread is the one that blocks.  Round 6 measured the
    textbook form (`pin_memory()` + `non_blocking=True`) on this stack: every second or third batch a later device synchronisation
    takes ~65 ms longer (12 ms steps: 11 77 12 11 77 ... against a flat 10 with pageable copies, `profiles/r06_pinned_h2d_stall.txt`;
    the same stall showed in the serve path, `kai0_amd/policy.py::_to_device`).  `pin=True` keeps the old form for a re-test."""

    _END = object()

    def __init__(self, loader, device, depth: int = 2, pin: bool = False):
        self._loader, self._device, self._depth = loader, torch.device(device), max(1, depth)
        self._cuda = self._device.type == "cuda"
        self._pin = bool(pin)

    def __iter__(self):
        q: queue.Queue = queue.Queue(maxsize=self._depth)
        stop = threading.Event()
        side = torch.cuda.Stream(self._device) if self._cuda else None

        def work():
            try:
                for obs, actions in self._loader:
                    if stop.is_set():
                        return
                    if self._cuda:
                        with torch.cuda.stream(side):
                            item = (_to_device_tree(obs, self._device, side, self._pin), _to_device_tree(actions, self._device, side, self._pin))
                            ev = torch.cuda.Event()
                            ev.record(side)
                    else:
                        item, ev = (_to_device_tree(obs, self._device, None, False), _to_device_tree(actions, self._device, None, False)), None
                    q.put((item, ev))
                q.put((self._END, None))
            except BaseException as e:  # surface loader errors in the consumer
                q.put((e, None))

        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                item, ev = q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                if ev is not None:
                    torch.cuda.current_stream(self._device).wait_event(ev)
                    for x in _leaves(item):
                        x.record_stream(torch.cuda.current_stream(self._device))  # the side stream allocated these tensors
                yield item
        finally:
            stop.set()
            while t.is_alive():  # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)


def _leaves(tree):
    if isinstance(tree, torch.Tensor):
        yield tree
    elif isinstance(tree, Observation):
        for v in vars(tree).values():
            yield from _leaves(v)
    elif isinstance(tree, Mapping):
        for v in tree.values():
            yield from _leaves(v)
    elif isinstance(tree, (tuple, list)):
        for v in tree:
            yield from _leaves(v)
