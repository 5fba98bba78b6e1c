"""The configuration / factory surface of SURVEY.md §8b: `TrainConfig`, `get_config`, `cli` (training/config.py:656-757,
1398-1412), `Pi0Config.load_pytorch` (models/model.py:276-280) and `create_trained_policy` (policy_config.py:16-94)."""

import dataclasses
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from kai0_amd import agilex_policy, normalize, policy, transforms  # noqa: E402
from kai0_amd import training_config as tc  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "host_pipeline.npz"))


def test_registry_and_lookup_errors():
    for name in ("pi05_flatten_fold_normal", "pi05_tee_shirt_sort_normal", "pi05_hang_cloth_normal", "pi05_flatten_fold_awbc",
                 "pi05_tee_shirt_sort_awbc", "pi05_hang_cloth_awbc", "ADVANTAGE_TORCH_KAI0_FLATTEN_FOLD", "debug", "debug_pi05"):  # fmt: skip
        assert tc.get_config(name).name == name
    c = tc.get_config("pi05_flatten_fold_normal")
    # config.py:1177-1190 and the TrainConfig defaults of :656-757
    assert (c.batch_size, c.num_train_steps, c.keep_period, c.num_workers) == (256, 100_000, 5000, 8)
    assert c.model.pi05 and c.model.model_type == "pi05" and c.model.max_token_len == 200 and c.model.discrete_state_input
    assert (c.lr_schedule.warmup_steps, c.lr_schedule.peak_lr, c.lr_schedule.decay_steps, c.lr_schedule.decay_lr) == (1000, 2.5e-5, 30_000, 2.5e-6)
    assert (c.optimizer.b1, c.optimizer.b2, c.optimizer.eps, c.optimizer.weight_decay, c.optimizer.clip_gradient_norm) == (0.9, 0.95, 1e-8, 1e-10, 1.0)
    assert c.ema_decay == 0.99 and c.seed == 42 and c.fsdp_devices == 1 and c.pytorch_training_precision == "bfloat16"
    assert c.data.default_prompt == "Flatten and fold the cloth." and not c.data.use_delta_joint_actions
    assert c.assets_dirs.name == "pi05_flatten_fold_normal"
    with pytest.raises(ValueError, match="--exp_name must be set"):
        _ = c.checkpoint_dir
    with pytest.raises(ValueError, match=r"Config 'pi05_flaten_fold_normal' not found\. Did you mean 'pi05_flatten_fold_normal'\?"):
        tc.get_config("pi05_flaten_fold_normal")
    est = tc.get_config("ADVANTAGE_TORCH_KAI0_FLATTEN_FOLD")
    assert est.advantage_estimator and est.skip_norm_stats and est.model.loss_value_weight == 1.0 and est.model.loss_action_weight == 0.0
    assert not est.model.discrete_state_input
    assert "his_-100_top_head" in est.data.repack_transforms.inputs[0].structure["images"]
    sched = c.lr_schedule.create()
    assert abs(sched(0) - 2.5e-5 / 1001) < 1e-12 and abs(sched(1000) - 2.5e-5) < 1e-12 and abs(sched(30_000) - 2.5e-6) < 1e-12


def test_cli_overrides():
    c = tc.cli(["pi05_flatten_fold_awbc", "--exp_name", "run1", "--batch-size", "64", "--no-wandb-enabled", "--resume"])
    assert (c.name, c.exp_name, c.batch_size, c.wandb_enabled, c.resume) == ("pi05_flatten_fold_awbc", "run1", 64, False, True)
    assert c.checkpoint_dir.parts[-2:] == ("pi05_flatten_fold_awbc", "run1")
    with pytest.raises(ValueError, match="Cannot resume and overwrite"):
        tc.cli(["debug_pi05", "--resume", "--overwrite"])
    with pytest.raises(SystemExit):
        tc.cli(["no_such_config"])


def _tok():
    return G["tok.model"].tobytes()


def _agilex_cfg(**kw):
    from tiny import tiny_cfgs

    pcfg, _ = tiny_cfgs(max_token_len=64)
    return tc.TrainConfig(name="tiny_agilex", exp_name="t", model=pcfg, policy_metadata={"robot": "agilex"},
                          data=tc.LerobotAgilexDataConfig(repo_id="tiny_agilex", default_prompt="Flatten and fold the cloth.",
                                                          tokenizer_model=_tok(), **kw))  # fmt: skip


def test_agilex_data_config_transform_stack(tmp_path):
    cfg = _agilex_cfg(use_delta_joint_actions=True, insert_advantage_into_prompt=True)
    dc = cfg.data.create(tmp_path, cfg.model)
    assert dc.repo_id == dc.asset_id == "tiny_agilex" and dc.norm_stats is None and dc.use_quantile_norm  # pi0.5: quantile norm
    assert [type(t).__name__ for t in dc.data_transforms.inputs] == ["InsertAdvantageIntoPrompt", "AgilexInputs", "DeltaActions"]
    assert [type(t).__name__ for t in dc.data_transforms.outputs] == ["AbsoluteActions", "AgilexOutputs"]
    assert dc.data_transforms.inputs[2].mask == transforms.make_bool_mask(6, -1, 6, -1)
    assert [type(t).__name__ for t in dc.model_transforms.inputs] == ["InjectDefaultPrompt", "ResizeImages", "TokenizePrompt", "PadStatesAndActions"]
    assert dc.action_sequence_keys == ("action",)
    # prompt_from_task: the dataset's task string is repacked as the prompt (config.py:410-418)
    cfg2 = dataclasses.replace(cfg, data=dataclasses.replace(cfg.data, base_config=tc.DataConfig(prompt_from_task=True)))
    dc2 = cfg2.data.create(tmp_path, cfg2.model)
    assert dc2.repack_transforms.inputs[0].structure["prompt"] == "prompt" and dc2.prompt_from_task
    # norm stats are picked up from <assets>/<asset_id>
    stats = {"state": normalize.NormStats(mean=np.zeros(14), std=np.ones(14), q01=-np.ones(14), q99=np.ones(14))}
    normalize.save(tmp_path / "tiny_agilex", stats)
    assert set(cfg.data.create(tmp_path, cfg.model).norm_stats) == {"state"}
    arx = dataclasses.replace(cfg, data=tc.LerobotARXDataConfig(repo_id="x", tokenizer_model=_tok()))
    assert isinstance(arx.data.create(tmp_path, arx.model).data_transforms.inputs[0], agilex_policy.ARXInputs)


def _checkpoint(tmp_path, cfg, seed=5):
    """A torch checkpoint directory as train_pytorch.py writes it: model.safetensors + assets/<asset_id>/norm_stats.json."""
    from kai0_amd.checkpoint import save_model_safetensors
    from kai0_amd.model import PI0Pytorch

    torch.manual_seed(seed)
    model = PI0Pytorch(cfg.model)
    ck = tmp_path / "ckpt" / "100"
    ck.mkdir(parents=True)
    save_model_safetensors(model, str(ck / "model.safetensors"))
    rng = np.random.default_rng(0)
    # norm stats are computed AFTER the robot transform, i.e. on the 32-dim padded vectors (compute_norm_stats.py)
    q = np.sort(rng.normal(size=(2, 32)), axis=0)
    stats = {k: normalize.NormStats(mean=rng.normal(size=32), std=rng.uniform(0.5, 2, 32), q01=q[0] - 1.5, q99=q[1] + 1.5)
             for k in ("state", "actions")}  # fmt: skip
    normalize.save(ck / "assets" / "tiny_agilex", stats)
    return model, ck, stats


def test_load_pytorch_and_create_trained_policy(tmp_path):
    cfg = _agilex_cfg(use_delta_joint_actions=False)
    model, ck, stats = _checkpoint(tmp_path, cfg)
    loaded = cfg.model.load_pytorch(cfg, str(ck / "model.safetensors"))
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert a.dtype == b.dtype and torch.equal(a, b), k
    pol = policy.create_trained_policy(cfg, ck, default_prompt="x", sample_kwargs={"num_steps": 5}, pytorch_device="cpu")
    names = [type(t).__name__ for t in pol._input_transform.transforms]
    assert names == ["InjectDefaultPrompt", "AgilexInputs", "Normalize", "InjectDefaultPrompt", "ResizeImages", "TokenizePrompt",
                     "PadStatesAndActions"]  # fmt: skip
    assert [type(t).__name__ for t in pol._output_transform.transforms] == ["Unnormalize", "AgilexOutputs"]
    assert pol._input_transform.transforms[0].prompt == "x" and pol._input_transform.transforms[2].use_quantiles
    assert pol.metadata == {"robot": "agilex"} and pol._sample_kwargs == {"num_steps": 5} and not pol._model.training
    got = pol._input_transform.transforms[2].norm_stats["actions"]
    assert np.array_equal(got.q99, stats["actions"].q99)  # the stats the run was trained with (checkpoint assets)
    # the input stack runs on a raw robot observation
    cams = {k[7:]: G[k] for k in G.files if k.startswith("ag.cam.") and "his_" not in k}
    out = pol._input_transform({"images": cams, "state": G["ag.state"].copy()})
    assert out["tokenized_prompt"].shape == (64,) and out["state"].shape == (32,) and out["image"]["base_0_rgb"].shape == (224, 224, 3)
    with pytest.raises(FileNotFoundError, match="model.safetensors"):
        policy.create_trained_policy(cfg, tmp_path / "ckpt", pytorch_device="cpu")
    with pytest.raises(ValueError, match="Asset id is required"):
        policy.create_trained_policy(dataclasses.replace(cfg, data=tc.FakeDataConfig(repo_id=None)), ck, pytorch_device="cpu")


def test_estimator_config_builds_the_estimator(tmp_path):
    from tiny import tiny_cfgs

    from kai0_amd.config import AdvantageEstimatorConfig
    from kai0_amd.checkpoint import save_model_safetensors
    from kai0_amd.model import AdvantageEstimator

    pcfg, _ = tiny_cfgs()
    ecfg = AdvantageEstimatorConfig(**{f.name: getattr(pcfg, f.name) for f in dataclasses.fields(pcfg)}, loss_value_weight=1.0,
                                    loss_action_weight=0.0)  # fmt: skip
    cfg = tc.TrainConfig(name="est", model=ecfg)
    m = AdvantageEstimator(ecfg)
    save_model_safetensors(m, str(tmp_path / "model.safetensors"))
    loaded = cfg.model.load_pytorch(cfg, str(tmp_path / "model.safetensors"))
    assert isinstance(loaded, AdvantageEstimator) and loaded.loss_value_weight == 1.0
    assert torch.equal(loaded.value_head[4].weight, m.value_head[4].weight)
