"""LeRobot-format dataset source for the training loader (`src/openpi/training/data_loader.py:131-228`,
`training/advantage_dataset.py:7-139`): what `create_torch_dataset` / `create_advantage_torch_dataset` hand to the transform
stack — per-frame dictionaries read from a LeRobot v2 dataset directory

    meta/info.json          fps, features {name: {dtype, shape}}, chunks_size, data_path / video_path templates
    meta/tasks.jsonl        {"task_index", "task"}            meta/episodes.jsonl   {"episode_index", "tasks", "length"}
    data/chunk-XXX/episode_YYYYYY.parquet                    one row per frame: state / action vectors, indices, timestamps
    videos/chunk-XXX/<camera>/episode_YYYYYY.mp4             camera streams (AV1); or image columns inside the parquet

with the semantics of `lerobot.common.datasets.lerobot_dataset.LeRobotDataset` (pinned by the reference's uv.lock; restated
here because `lerobot` is not installable offline): `delta_timestamps` turn a key into a stacked window `[T, ...]` clamped to
the episode with a `<key>_is_pad` mask, scalars come back as 0-d tensors, images as float32 CHW in [0, 1], `task` as the
episode's instruction string.

The tabular side is native (pyarrow -> numpy, the whole selected split resident in host memory: 14-DoF x 2 vectors per frame
are ~150 B, i.e. ~0.5 GB per million frames).  Video decoding is a plug: `frame_decoder(path, timestamps_s, tolerance_s)
-> uint8 [T, H, W, 3]`; PyAV / OpenCV are used if importable, and a dataset with video features raises a clear error otherwise
(neither ships in this image).  Image features stored in the parquet (PNG / JPEG bytes) decode with PIL."""

from __future__ import annotations

import io
import json
import logging
import pathlib
import random
from collections.abc import Callable, Sequence

import numpy as np
import torch

from . import transforms as _transforms

logger = logging.getLogger("kai0_amd")


class VideoDecodeUnavailable(RuntimeError):
    pass


def _read_jsonl(path: pathlib.Path) -> list[dict]:
    with open(path) as f:
        return [json.loads(line) for line in f if line.strip()]


class LeRobotDatasetMetadata:
    """`LeRobotDatasetMetadata(repo_id)` for a local dataset directory (`repo_id` = its path; nothing is downloaded)."""

    def __init__(self, repo_id: str | pathlib.Path, root: str | pathlib.Path | None = None):
        self.repo_id = str(repo_id)
        self.root = pathlib.Path(root if root is not None else repo_id)
        info_path = self.root / "meta" / "info.json"
        if not info_path.exists():
            raise FileNotFoundError(f"{info_path} not found: {self.root} is not a LeRobot v2 dataset directory")
        self.info = json.loads(info_path.read_text())
        self.tasks = {int(t["task_index"]): t["task"] for t in _read_jsonl(self.root / "meta" / "tasks.jsonl")}
        self.episodes = {int(e["episode_index"]): e for e in _read_jsonl(self.root / "meta" / "episodes.jsonl")}

    @property
    def fps(self) -> int:
        return self.info["fps"]

    @property
    def features(self) -> dict:
        return self.info["features"]

    @property
    def video_keys(self) -> list[str]:
        return [k for k, ft in self.features.items() if ft["dtype"] == "video"]

    @property
    def image_keys(self) -> list[str]:
        return [k for k, ft in self.features.items() if ft["dtype"] == "image"]

    @property
    def camera_keys(self) -> list[str]:
        return [k for k, ft in self.features.items() if ft["dtype"] in ("video", "image")]

    @property
    def total_episodes(self) -> int:
        return self.info.get("total_episodes", len(self.episodes))

    def episode_chunk(self, ep: int) -> int:
        return ep // self.info.get("chunks_size", 1000)

    def data_file(self, ep: int) -> pathlib.Path:
        return self.root / self.info["data_path"].format(episode_chunk=self.episode_chunk(ep), episode_index=ep)

    def video_file(self, ep: int, key: str) -> pathlib.Path:
        return self.root / self.info["video_path"].format(episode_chunk=self.episode_chunk(ep), video_key=key, episode_index=ep)


def _default_frame_decoder(path: pathlib.Path, timestamps: Sequence[float], tolerance_s: float) -> np.ndarray:
    """uint8 [T, H, W, 3] frames nearest to `timestamps` (seconds); PyAV first, OpenCV second."""
    try:
        import av  # type: ignore
    except ImportError:
        av = None
    if av is not None:
        with av.open(str(path)) as container:
            stream = container.streams.video[0]
            first, last = min(timestamps), max(timestamps)
            container.seek(int(max(first - 1.0, 0.0) / stream.time_base), stream=stream, backward=True, any_frame=False)
            loaded, ts = [], []
            for frame in container.decode(stream):
                t = float(frame.pts * stream.time_base)
                loaded.append(frame.to_ndarray(format="rgb24"))
                ts.append(t)
                if t >= last:
                    break
        ts = np.asarray(ts)
        idx = [int(np.abs(ts - q).argmin()) for q in timestamps]
        worst = max(abs(ts[i] - q) for i, q in zip(idx, timestamps))
        if worst > tolerance_s:
            raise ValueError(f"{path}: no frame within {tolerance_s}s of the requested timestamps (worst {worst:.4f}s)")
        return np.stack([loaded[i] for i in idx])
    try:
        import cv2  # type: ignore
    except ImportError:
        cv2 = None
    if cv2 is not None:
        cap = cv2.VideoCapture(str(path))
        fps = cap.get(cv2.CAP_PROP_FPS)
        out = []
        for q in timestamps:
            cap.set(cv2.CAP_PROP_POS_FRAMES, round(q * fps))
            ok, frame = cap.read()
            if not ok:
                raise ValueError(f"{path}: cannot decode the frame at {q}s")
            out.append(frame[..., ::-1])
        cap.release()
        return np.stack(out)
    raise VideoDecodeUnavailable(
        f"{path}: this dataset stores its cameras as video, and neither PyAV nor OpenCV is importable. Install one of them, "
        "pass `frame_decoder=` (path, timestamps_s, tolerance_s) -> uint8 [T, H, W, 3], or re-encode the cameras as image "
        "features inside the parquet files.")  # fmt: skip


def get_delta_indices(delta_timestamps: dict[str, list[float]], fps: int, tolerance_s: float = 1e-4) -> dict[str, list[int]]:
    """lerobot `check_delta_timestamps` + `get_delta_indices`: every offset must be a multiple of 1 / fps."""
    out = {}
    for key, deltas in delta_timestamps.items():
        bad = [d for d in deltas if abs(d * fps - round(d * fps)) / fps > tolerance_s]
        if bad:
            raise ValueError(f"delta_timestamps[{key!r}] has values that are not multiples of 1/fps = {1 / fps}: {bad[:4]}")
        out[key] = [round(d * fps) for d in deltas]
    return out


class LeRobotDataset:
    """Map-style dataset of frames (see the module docstring).  `episodes`: subset, in this order (default: all)."""

    def __init__(self, repo_id: str | pathlib.Path, root: str | pathlib.Path | None = None, episodes: list[int] | None = None,
                 image_transforms: Callable | None = None, delta_timestamps: dict[str, list[float]] | None = None,
                 tolerance_s: float = 1e-4, frame_decoder: Callable | None = None, **_ignored):  # fmt: skip
        import pyarrow.parquet as pq

        self.repo_id = str(repo_id)
        self.meta = LeRobotDatasetMetadata(repo_id, root)
        self.root = self.meta.root
        self.episodes = list(episodes) if episodes is not None else None
        self.image_transforms = image_transforms
        self.tolerance_s = tolerance_s
        self.frame_decoder = frame_decoder or _default_frame_decoder
        self.delta_timestamps = delta_timestamps
        self.delta_indices = get_delta_indices(delta_timestamps, self.meta.fps, tolerance_s) if delta_timestamps else None
        eps = self.episodes if self.episodes is not None else sorted(self.meta.episodes)
        tables = []
        lengths = []
        for ep in eps:
            t = pq.read_table(self.meta.data_file(ep))
            tables.append(t)
            lengths.append(t.num_rows)
        import pyarrow as pa

        table = pa.concat_tables(tables) if tables else None
        self._columns: dict[str, np.ndarray | list] = {}
        self._image_cols: dict[str, list] = {}
        for name in (table.column_names if table is not None else []):
            ft = self.meta.features.get(name, {})
            col = table.column(name)
            if ft.get("dtype") == "image":
                self._image_cols[name] = col.to_pylist()  # {"bytes": ..., "path": ...} per frame
                continue
            arr = col.to_numpy(zero_copy_only=False) if not pa.types.is_list(col.type) and not pa.types.is_fixed_size_list(col.type) else None
            if arr is None:
                arr = np.stack([np.asarray(x) for x in col.to_pylist()]) if len(col) else np.zeros((0,))
            if arr.dtype == np.float64 and ft.get("dtype") == "float32":
                arr = arr.astype(np.float32)
            self._columns[name] = arr
        ends = np.cumsum(lengths)
        self.episode_data_index = {"from": torch.as_tensor(ends - np.asarray(lengths), dtype=torch.int64),
                                   "to": torch.as_tensor(ends, dtype=torch.int64)}  # fmt: skip
        self._ep_pos = {ep: i for i, ep in enumerate(eps)}  # episode index -> position in episode_data_index
        self._num_frames = int(ends[-1]) if len(ends) else 0

    # ------------------------------------------------------------------------------------------------ LeRobotDataset API
    @property
    def fps(self) -> int:
        return self.meta.fps

    @property
    def num_frames(self) -> int:
        return self._num_frames

    @property
    def num_episodes(self) -> int:
        return len(self._ep_pos)

    def __len__(self) -> int:
        return self._num_frames

    def _row(self, idx: int) -> dict:
        item = {}
        for k, arr in self._columns.items():
            item[k] = torch.as_tensor(np.asarray(arr[idx]))
        return item

    def _get_query_indices(self, idx: int, ep_pos: int):
        start, end = int(self.episode_data_index["from"][ep_pos]), int(self.episode_data_index["to"][ep_pos])
        query = {k: [max(start, min(end - 1, idx + d)) for d in deltas] for k, deltas in self.delta_indices.items()}
        padding = {f"{k}_is_pad": torch.BoolTensor([(idx + d < start) | (idx + d >= end) for d in deltas])
                   for k, deltas in self.delta_indices.items()}  # fmt: skip
        return query, padding

    def _query_hf_dataset(self, query_indices: dict[str, list[int]]) -> dict:
        out = {}
        for k, q in query_indices.items():
            if k in self.meta.video_keys:
                continue
            if k in self._image_cols:
                out[k] = torch.stack([self._decode_image(self._image_cols[k][i]) for i in q])
            else:
                out[k] = torch.as_tensor(np.stack([np.asarray(self._columns[k][i]) for i in q]))
        return out

    @staticmethod
    def _decode_image(cell) -> torch.Tensor:
        from PIL import Image

        raw = cell["bytes"] if isinstance(cell, dict) else cell
        img = np.array(Image.open(io.BytesIO(raw)).convert("RGB"))
        return torch.from_numpy(img).permute(2, 0, 1).to(torch.float32) / 255.0  # CHW in [0, 1], as lerobot's hf_transform

    def _query_videos(self, query_timestamps: dict[str, list[float]], ep: int) -> dict:
        out = {}
        for k, ts in query_timestamps.items():
            frames = self.frame_decoder(self.meta.video_file(ep, k), ts, self.tolerance_s)
            t = torch.from_numpy(np.ascontiguousarray(frames)).permute(0, 3, 1, 2).to(torch.float32) / 255.0
            out[k] = t.squeeze(0) if t.shape[0] == 1 else t
        return out

    def _get_query_timestamps(self, current_ts: float, query_indices: dict | None) -> dict[str, list[float]]:
        out = {}
        for k in self.meta.video_keys:
            if query_indices is not None and k in query_indices:
                out[k] = [float(self._columns["timestamp"][i]) for i in query_indices[k]]
            else:
                out[k] = [current_ts]
        return out

    def get_sample_with_imgs_from_idx(self, idx: int) -> dict:
        """One frame with its cameras decoded (no action window)."""
        item = self._row(idx)
        ep = int(item["episode_index"])
        for k, cells in self._image_cols.items():
            item[k] = self._decode_image(cells[idx])
        if self.meta.video_keys:
            item = {**self._query_videos(self._get_query_timestamps(float(item["timestamp"]), None), ep), **item}
        if self.image_transforms is not None:
            for cam in self.meta.camera_keys:
                item[cam] = self.image_transforms(item[cam])
        return item

    def __getitem__(self, idx: int) -> dict:
        if idx < 0:
            idx += len(self)
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        item = self._row(idx)
        ep = int(item["episode_index"])
        query_indices = None
        if self.delta_indices is not None:
            query_indices, padding = self._get_query_indices(idx, self._ep_pos[ep])
            item = {**item, **padding, **self._query_hf_dataset(query_indices)}
        for k, cells in self._image_cols.items():
            if query_indices is None or k not in query_indices:
                item[k] = self._decode_image(cells[idx])
        if self.meta.video_keys:
            frames = self._query_videos(self._get_query_timestamps(float(item["timestamp"]), query_indices), ep)
            item = {**frames, **item}
        if self.image_transforms is not None:
            for cam in self.meta.camera_keys:
                item[cam] = self.image_transforms(item[cam])
        item["task"] = self.meta.tasks[int(item["task_index"])]
        return item


class AdvantageLerobotDataset(LeRobotDataset):
    """`training/advantage_dataset.py:7-139`: every sample also carries a RANDOM other frame of the same episode under
    `his_-100_<key>` and the label `progress = stage_progress_gt - his_-100_stage_progress_gt` (Stage-Advantage estimator)."""

    def __getitem__(self, idx: int) -> dict:
        if idx < 0:
            idx += len(self)
        item = self.get_sample_with_imgs_from_idx(idx)
        ep = int(item["episode_index"])
        ts = float(item["timestamp"])
        level = {"episode_length": self.meta.episodes[ep]["length"]}
        if self.delta_indices is not None:
            query_indices, padding = self._get_query_indices(idx, self._ep_pos[ep])
            level = {**level, **padding, **self._query_hf_dataset(query_indices)}
        level["task"] = self.meta.tasks[int(item["task_index"])]
        pos = self._ep_pos[ep]
        start, end = int(self.episode_data_index["from"][pos]), int(self.episode_data_index["to"][pos])
        if end - start < 2:
            raise ValueError(f"episode {ep} has a single frame: no comparison frame to draw")
        while True:
            j = random.randint(start, end - 1)
            if j == idx:
                continue
            other = self.get_sample_with_imgs_from_idx(j)
            if int(other["episode_index"]) == ep and float(other["timestamp"]) != ts:
                break
        final = {**{f"his_-100_{k}": v for k, v in other.items()}, **item, **level}
        final["progress"] = float(final["stage_progress_gt"]) - float(final["his_-100_stage_progress_gt"])
        return final


# ------------------------------------------------------------------------------------------- data_loader.py:131-228
def episodes_split_through_task(repo_id, split_ratio: float = 0.9, split_type: str = "train", shuffle: bool = False,
                                random_seed: int = 42) -> list[int]:  # fmt: skip
    """data_loader.py:185-213: per task, the first `split_ratio` of its episodes train, the rest validate."""
    assert split_type in ["all", "train", "val"], f"Invalid split type '{split_type}'. Must be 'all', 'train' or 'val'."
    meta = LeRobotDatasetMetadata(repo_id)
    index = list(meta.episodes.keys())
    if split_type == "all":
        return index
    by_task: dict[str, list[int]] = {}
    for i in index:
        by_task.setdefault("".join(meta.episodes[i]["tasks"]), []).append(i)
    train, val = [], []
    for eps in by_task.values():
        n = int(len(eps) * split_ratio)
        train.extend(eps[:n])
        val.extend(eps[n:])
    return train if split_type == "train" else val


def create_torch_dataset(data_config, action_horizon: int, model_config, **dataset_kw):
    """data_loader.py:131-152."""
    from .data_loader import FakeDataset, TransformedDataset

    repo_id = data_config.repo_id
    if repo_id is None:
        raise ValueError("Repo ID is not set. Cannot create dataset.")
    if repo_id == "fake":
        return FakeDataset(model_config, num_samples=1024)
    meta = LeRobotDatasetMetadata(repo_id)
    dataset = LeRobotDataset(
        repo_id, episodes=getattr(data_config, "episodes", None),
        delta_timestamps={key: [t / meta.fps for t in range(action_horizon)] for key in data_config.action_sequence_keys},
        **dataset_kw)  # fmt: skip
    if data_config.prompt_from_task:
        dataset = TransformedDataset(dataset, [_transforms.PromptFromLeRobotTask(meta.tasks)])
    return dataset


def create_advantage_torch_dataset(data_config, action_horizon: int, model_config, config=None, **dataset_kw):
    """data_loader.py:154-182."""
    from .data_loader import FakeDataset, TransformedDataset

    split = getattr(config, "split", "all") or "all"
    repo_id = data_config.repo_id
    if repo_id is None:
        raise ValueError("Repo ID is not set. Cannot create dataset.")
    if repo_id == "fake":
        return FakeDataset(model_config, num_samples=1024)
    meta = LeRobotDatasetMetadata(repo_id)
    dataset = AdvantageLerobotDataset(
        repo_id, episodes=episodes_split_through_task(repo_id, split_type=split, shuffle=False),
        delta_timestamps={key: [t / meta.fps for t in range(action_horizon)] for key in data_config.action_sequence_keys},
        **dataset_kw)  # fmt: skip
    if data_config.prompt_from_task:
        dataset = TransformedDataset(dataset, [_transforms.PromptFromLeRobotTask(meta.tasks)])
    return dataset


def transform_dataset(dataset, data_config, *, skip_norm_stats: bool = False):
    """data_loader.py:230-251: repack -> robot transforms -> Normalize -> model transforms."""
    from .data_loader import TransformedDataset

    norm_stats = {}
    if data_config.repo_id != "fake" and not skip_norm_stats:
        if data_config.norm_stats is None:
            raise ValueError("Normalization stats not found. Make sure to run `scripts/compute_norm_stats.py --config-name=<your-config>`.")
        norm_stats = data_config.norm_stats
    return TransformedDataset(dataset, [*data_config.repack_transforms.inputs, *data_config.data_transforms.inputs,
                                        _transforms.Normalize(norm_stats, use_quantiles=data_config.use_quantile_norm),
                                        *data_config.model_transforms.inputs])  # fmt: skip


def create_data_loader(config, *, shuffle: bool = False, num_batches: int | None = None, skip_norm_stats: bool = False,
                       framework: str = "pytorch", **dataset_kw):  # fmt: skip
    """`create_data_loader(train_config)` (data_loader.py:291-330, torch path): TrainConfig -> dataset -> transforms -> batches of
    `(Observation, actions)`; under torch.distributed every rank gets batch_size / world_size samples of its own shard."""
    from .data_loader import create_torch_data_loader

    if framework != "pytorch":
        raise ValueError("kai0_amd loads data for the torch path only")
    data_config = config.data.create(config.assets_dirs, config.model)
    logger.info(f"data_config: {data_config}")
    if getattr(config, "advantage_estimator", False):
        dataset = create_advantage_torch_dataset(data_config, config.model.action_horizon, config.model, config, **dataset_kw)
    else:
        dataset = create_torch_dataset(data_config, config.model.action_horizon, config.model, **dataset_kw)
    dataset = transform_dataset(dataset, data_config, skip_norm_stats=skip_norm_stats or getattr(config, "skip_norm_stats", False))
    return create_torch_data_loader(dataset, config.batch_size, shuffle=shuffle, num_batches=num_batches,
                                    num_workers=config.num_workers, seed=config.seed, data_config=data_config)  # fmt: skip
