"""The LeRobot-format dataset source (kai0_amd.lerobot_dataset) on a synthetic v2 dataset directory written with pyarrow:
window / padding semantics of `delta_timestamps`, episode subsets, task prompts, image features, the video-decoder plug, the
Stage-Advantage dataset, and the whole `create_data_loader(train_config)` path down to `(Observation, actions)` batches."""

import io
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from kai0_amd import lerobot_dataset as lrd  # noqa: E402
from kai0_amd import normalize  # noqa: E402
from kai0_amd import training_config as tc  # noqa: E402

FPS = 30
LENGTHS = [7, 5, 9]
TASKS = ["Flatten and fold the cloth.", "Hang the cloth."]
EP_TASK = [0, 1, 0]
CAMS = ("top_head", "hand_left", "hand_right")


def _png(arr: np.ndarray) -> bytes:
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG")
    return buf.getvalue()


def _frame_image(ep: int, fr: int, cam: int) -> np.ndarray:
    img = np.zeros((12, 16, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10 * ep + 1, 5 * fr + 2, 40 * cam + 3
    return img


def make_dataset(root, *, video_cam: bool = False, extras: bool = False):
    import pyarrow as pa
    import pyarrow.parquet as pq

    root.mkdir(parents=True, exist_ok=True)
    (root / "meta").mkdir()
    feats = {"observation.state": {"dtype": "float32", "shape": [14]}, "action": {"dtype": "float32", "shape": [14]},
             "timestamp": {"dtype": "float32", "shape": [1]}, "frame_index": {"dtype": "int64", "shape": [1]},
             "episode_index": {"dtype": "int64", "shape": [1]}, "index": {"dtype": "int64", "shape": [1]},
             "task_index": {"dtype": "int64", "shape": [1]}}  # fmt: skip
    for c, cam in enumerate(CAMS):
        feats[f"observation.images.{cam}"] = {"dtype": "video" if (video_cam and c == 2) else "image", "shape": [12, 16, 3]}
    if extras:
        feats["stage_progress_gt"] = {"dtype": "float32", "shape": [1]}
        feats["progress_gt"] = {"dtype": "float32", "shape": [1]}
    info = {"codebase_version": "v2.1", "fps": FPS, "chunks_size": 2, "total_episodes": len(LENGTHS), "features": feats,
            "data_path": "data/chunk-{episode_chunk:03d}/episode_{episode_index:06d}.parquet",
            "video_path": "videos/chunk-{episode_chunk:03d}/{video_key}/episode_{episode_index:06d}.mp4"}  # fmt: skip
    (root / "meta" / "info.json").write_text(json.dumps(info))
    (root / "meta" / "tasks.jsonl").write_text("\n".join(json.dumps({"task_index": i, "task": t}) for i, t in enumerate(TASKS)))
    (root / "meta" / "episodes.jsonl").write_text(
        "\n".join(json.dumps({"episode_index": e, "tasks": [TASKS[EP_TASK[e]]], "length": n}) for e, n in enumerate(LENGTHS)))
    g = 0
    for e, n in enumerate(LENGTHS):
        cols = {"observation.state": [list(np.full(14, 100 * e + f, np.float32) / 1000) for f in range(n)],
                "action": [list(np.full(14, 100 * e + f, np.float32) / 100) for f in range(n)],
                "timestamp": np.arange(n, dtype=np.float32) / FPS, "frame_index": np.arange(n), "episode_index": np.full(n, e),
                "index": np.arange(g, g + n), "task_index": np.full(n, EP_TASK[e])}  # fmt: skip
        arrays = {"observation.state": pa.array(cols["observation.state"], type=pa.list_(pa.float32())),
                  "action": pa.array(cols["action"], type=pa.list_(pa.float32())),
                  **{k: pa.array(cols[k]) for k in ("timestamp", "frame_index", "episode_index", "index", "task_index")}}  # fmt: skip
        for c, cam in enumerate(CAMS):
            if feats[f"observation.images.{cam}"]["dtype"] == "image":
                arrays[f"observation.images.{cam}"] = pa.array([{"bytes": _png(_frame_image(e, f, c)), "path": None} for f in range(n)])
        if extras:
            arrays["stage_progress_gt"] = pa.array(np.linspace(0, 1, n).astype(np.float32))
            arrays["progress_gt"] = pa.array(np.linspace(0, 1, n).astype(np.float32))
        p = root / f"data/chunk-{e // 2:03d}/episode_{e:06d}.parquet"
        p.parent.mkdir(parents=True, exist_ok=True)
        pq.write_table(pa.table(arrays), p)
        g += n
    return root


def test_windows_padding_tasks_and_images(tmp_path):
    root = make_dataset(tmp_path / "ds")
    meta = lrd.LeRobotDatasetMetadata(root)
    assert meta.fps == FPS and meta.tasks == dict(enumerate(TASKS)) and meta.episodes[1]["length"] == 5
    assert meta.camera_keys == [f"observation.images.{c}" for c in CAMS] and meta.video_keys == []
    ds = lrd.LeRobotDataset(root, delta_timestamps={"action": [t / FPS for t in range(4)]})
    assert len(ds) == sum(LENGTHS) and ds.num_episodes == 3
    assert ds.episode_data_index["from"].tolist() == [0, 7, 12] and ds.episode_data_index["to"].tolist() == [7, 12, 21]
    it = ds[5]  # episode 0, frame 5: window 5, 6, then clamped to the last frame and flagged
    assert it["action"].shape == (4, 14) and it["action"].dtype == torch.float32
    assert torch.allclose(it["action"][:, 0], torch.tensor([0.05, 0.06, 0.06, 0.06]))
    assert it["action_is_pad"].tolist() == [False, False, True, True]
    assert it["observation.state"].shape == (14,) and abs(float(it["observation.state"][0]) - 0.005) < 1e-7
    assert it["episode_index"].ndim == 0 and int(it["frame_index"]) == 5 and it["task"] == TASKS[0]
    img = it["observation.images.hand_left"]
    assert img.shape == (3, 12, 16) and img.dtype == torch.float32
    assert torch.equal((img * 255).round().to(torch.uint8)[:, 0, 0], torch.tensor([1, 27, 43], dtype=torch.uint8))
    it = ds[7 + 4]  # last frame of episode 1 (task 1): the window never crosses into episode 2
    assert it["task"] == TASKS[1] and torch.allclose(it["action"][:, 0], torch.full((4,), 1.04))
    assert it["action_is_pad"].tolist() == [False, True, True, True]
    with pytest.raises(IndexError):
        ds[len(ds)]
    with pytest.raises(ValueError, match="multiples of 1/fps"):
        lrd.LeRobotDataset(root, delta_timestamps={"action": [0.0, 0.5 / FPS]})
    with pytest.raises(FileNotFoundError, match="not a LeRobot v2 dataset"):
        lrd.LeRobotDatasetMetadata(tmp_path)


def test_episode_subset_and_task_split(tmp_path):
    root = make_dataset(tmp_path / "ds")
    ds = lrd.LeRobotDataset(root, episodes=[2, 0], delta_timestamps={"action": [0.0, 1 / FPS]})
    assert len(ds) == 16 and ds.episode_data_index["from"].tolist() == [0, 9]
    assert int(ds[0]["episode_index"]) == 2 and int(ds[9]["episode_index"]) == 0
    assert ds[8]["action_is_pad"].tolist() == [False, True] and ds[15]["action_is_pad"].tolist() == [False, True]
    assert lrd.episodes_split_through_task(root, split_type="all") == [0, 1, 2]
    # per task: first 90 % train (int(2 * 0.9) = 1 of task 0's two episodes, int(1 * 0.9) = 0 of task 1's one)
    assert lrd.episodes_split_through_task(root, split_type="train") == [0]
    assert lrd.episodes_split_through_task(root, split_type="val") == [2, 1]


def test_video_features_go_through_the_decoder_plug(tmp_path):
    root = make_dataset(tmp_path / "ds", video_cam=True)
    calls = []

    def decoder(path, timestamps, tol):
        calls.append((path.name, path.parent.name, path.parent.parent.name, list(timestamps)))
        return np.stack([np.full((12, 16, 3), round(t * FPS), np.uint8) for t in timestamps])

    ds = lrd.LeRobotDataset(root, delta_timestamps={"action": [0.0, 1 / FPS]}, frame_decoder=decoder)
    it = ds[7 + 3]  # episode 1 (chunk 0), frame 3
    assert calls[-1][:3] == ("episode_000001.mp4", "observation.images.hand_right", "chunk-000")
    assert abs(calls[-1][3][0] - 3 / FPS) < 1e-6
    v = it["observation.images.hand_right"]
    assert v.shape == (3, 12, 16) and abs(float(v[0, 0, 0]) - 3 / 255) < 1e-6
    assert int(lrd.LeRobotDataset(root, frame_decoder=decoder)[20]["frame_index"]) == 8  # episode 2 lives in chunk-001
    assert calls[-1][2] == "chunk-001"
    with pytest.raises(lrd.VideoDecodeUnavailable, match="neither PyAV nor OpenCV"):
        lrd.LeRobotDataset(root)[0]


def test_advantage_dataset_draws_a_comparison_frame(tmp_path):
    import random

    root = make_dataset(tmp_path / "ds", extras=True)
    ds = lrd.AdvantageLerobotDataset(root, episodes=[0, 2], delta_timestamps={"action": [0.0, 1 / FPS]})
    random.seed(0)
    for idx in (0, 6, 7, 15):
        it = ds[idx]
        ep = int(it["episode_index"])
        assert int(it["his_-100_episode_index"]) == ep and int(it["his_-100_frame_index"]) != int(it["frame_index"])
        assert it["his_-100_observation.images.top_head"].shape == (3, 12, 16)
        want = float(it["stage_progress_gt"]) - float(it["his_-100_stage_progress_gt"])
        assert abs(it["progress"] - want) < 1e-7 and it["episode_length"] == LENGTHS[ep] and it["task"] == TASKS[0]
        assert it["action"].shape == (2, 14)


def test_create_data_loader_from_train_config(tmp_path):
    """TrainConfig -> LeRobot dataset -> repack -> AgilexInputs -> Normalize -> resize / tokenise / pad -> batches."""
    from test_training_config_cpu import _tok
    from tiny import tiny_cfgs

    root = make_dataset(tmp_path / "FlattenFold" / "base")
    pcfg, _ = tiny_cfgs(action_horizon=10, max_token_len=64)
    assets = tmp_path / "assets"
    cfg = tc.TrainConfig(name="tiny_ff", exp_name="t", model=pcfg, batch_size=4, num_workers=0, assets_base_dir=str(assets),
                         data=tc.LerobotAgilexDataConfig(repo_id=str(root), default_prompt="Flatten and fold the cloth.",
                                                         use_delta_joint_actions=False, tokenizer_model=_tok(),
                                                         assets=tc.AssetsConfig(asset_id="ff"),
                                                         base_config=tc.DataConfig(prompt_from_task=True)))  # fmt: skip
    with pytest.raises(ValueError, match="Normalization stats not found"):
        lrd.create_data_loader(cfg)
    stats = {k: normalize.NormStats(mean=np.zeros(32), std=np.ones(32), q01=-np.ones(32), q99=np.ones(32)) for k in ("state", "actions")}
    normalize.save(cfg.assets_dirs / "ff", stats)
    loader = lrd.create_data_loader(cfg, shuffle=False, num_batches=2)
    batches = list(loader)
    assert len(batches) == 2
    obs, actions = batches[0]
    # (float64 norm stats promote the normalised vectors, as in the reference; the model casts them on entry)
    assert actions.shape == (4, 10, 32) and actions.dtype in (torch.float32, torch.float64)
    assert obs.state.shape == (4, 32) and obs.tokenized_prompt.shape == (4, 64) and obs.tokenized_prompt_mask.dtype == torch.bool
    assert set(obs.images) == {"base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb"}
    assert obs.images["base_0_rgb"].shape == (4, 3, 224, 224) and obs.images["base_0_rgb"].dtype == torch.float32
    # frame 0 of episode 0: actions 0.00, 0.01, ... quantile-normalised with q01 = -1, q99 = 1 -> (x + 1) / (2 + 1e-6) * 2 - 1
    assert abs(float(actions[0, 3, 0]) - ((0.03 + 1) / (2 + 1e-6) * 2 - 1)) < 1e-5
    assert loader.data_config().repo_id == str(root)
