"""`train_loop(config)` and the trainer's checkpoint contract on CPU (scripts/train_pytorch.py:149-259,309-633).

The HIP model cannot run on the CPU, so the pure-torch oracle (same parameter tree and dtypes) stands in for it and the shard
arithmetic is the torch restatement of the HIP optimizer kernels (tests/test_sharded_cpu.py::TorchShardOps): what is tested is
the loop — TrainConfig -> data loader -> steps -> log records -> checkpoints (with norm stats, config, RNG state) -> resume.
The same loop over the HIP model runs on the GPU in tests/test_model_gpu.py::test_train_loop_debug_pi05_resume_is_exact."""

import dataclasses
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from test_sharded_cpu import TorchShardOps  # noqa: E402

from kai0_amd import normalize, policy  # noqa: E402
from kai0_amd import training_config as tc  # noqa: E402


def _cfgs():
    from kai0_amd.config import Pi0Config, SiglipConfig
    from oracle.pi0_oracle import OracleConfig, SiglipCfg

    sk = dict(hidden_size=32, num_layers=1, num_heads=2, intermediate_size=64, patch_size=14, image_size=28, projection_dim=64,
              layer_norm_eps=1e-6)  # fmt: skip
    common = dict(dtype="bfloat16", paligemma_variant="dummy", action_expert_variant="dummy", action_dim=32, action_horizon=6,
                  max_token_len=12, pi05=True, vocab_size=2048)  # FakeDataset draws token ids below 2048
    return Pi0Config(siglip=SiglipConfig(**sk), **common), OracleConfig(siglip=SiglipCfg(**sk), **common)


def _stand_in(ocfg, seed=0):
    """The oracle behind the HIP model's call signature: noise / time are drawn from torch's global generator when not given
    (pi0_pytorch.py:316-373), which is the state a resumed run must restore."""
    from oracle.pi0_oracle import OraclePI0, synthetic_weights_

    class StandIn(OraclePI0):
        def forward(self, observation, actions, noise=None, time=None):
            if noise is None:
                noise = torch.randn(actions.shape)
            if time is None:
                time = torch.rand(actions.shape[0]) * 0.999 + 0.001
            return super().forward(observation, actions, noise, time)

    m = StandIn(ocfg)
    synthetic_weights_(m, seed=seed)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() >= 2:
                p.mul_(4.0)
    return m


def _config(tmp_path, **kw):
    pcfg, _ = _cfgs()
    base = dict(name="tiny_loop", exp_name="e", model=pcfg, data=tc.FakeDataConfig(), batch_size=2, num_workers=0, num_train_steps=6,
                log_interval=1, save_interval=100, checkpoint_base_dir=str(tmp_path / "ckpt"), assets_base_dir=str(tmp_path / "assets"),
                lr_schedule=tc.CosineDecaySchedule(warmup_steps=2, peak_lr=1e-3, decay_steps=10, decay_lr=1e-4), wandb_enabled=False)  # fmt: skip
    base.update(kw)
    return tc.TrainConfig(**base)


def _run(cfg, seed=0):
    from kai0_amd.train import train_loop

    torch.set_num_threads(2)
    _, ocfg = _cfgs()
    return train_loop(cfg, device="cpu", shard_ops=TorchShardOps(), model=_stand_in(ocfg, seed))


@pytest.mark.timeout(300)
def test_train_loop_runs_logs_checkpoints_and_resumes_exactly(tmp_path):
    full = _run(_config(tmp_path, exp_name="full", overwrite=True))
    assert [r["step"] for r in full] == list(range(6)) and all(np.isfinite(r["loss"]) for r in full)
    assert abs(full[0]["learning_rate"] - 1e-3 / 3) < 1e-12 and abs(full[2]["learning_rate"] - 1e-3) < 1e-12  # warm-up from peak / (w + 1)
    ck = tmp_path / "ckpt" / "tiny_loop" / "full"
    assert sorted(os.listdir(ck)) == ["6"]  # save_interval 100: only the final step
    assert sorted(os.listdir(ck / "6")) == ["metadata.pt", "model.safetensors", "optimizer.pt"]  # fake data: no norm stats
    meta = torch.load(ck / "6" / "metadata.pt", weights_only=True)
    assert meta["global_step"] == 6 and meta["config"]["name"] == "tiny_loop" and meta["config"]["batch_size"] == 2
    assert meta["timestamp"] > 0 and len(meta["rng_state"]) == 1

    # the same run cut after 4 steps ("killed"), then resumed to 6: steps 4 and 5 are the uninterrupted run's
    part = _run(_config(tmp_path, exp_name="cut", num_train_steps=4, overwrite=True))
    assert [r["loss"] for r in part] == [r["loss"] for r in full[:4]]
    assert sorted(os.listdir(tmp_path / "ckpt" / "tiny_loop" / "cut")) == ["4"]
    rest = _run(_config(tmp_path, exp_name="cut", resume=True), seed=123)  # (the resumed process builds its own initial weights)
    assert [r["step"] for r in rest] == [4, 5]
    for a, b in zip(rest, full[4:]):
        assert a["loss"] == b["loss"] and a["grad_norm"] == b["grad_norm"] and a["learning_rate"] == b["learning_rate"], (a, b)
    assert sorted(os.listdir(tmp_path / "ckpt" / "tiny_loop" / "cut")) == ["4", "6"]
    with pytest.raises(FileNotFoundError, match="does not exist for resume"):
        _run(_config(tmp_path, exp_name="nope", resume=True))


def test_checkpoint_carries_the_norm_stats_the_policy_loads(tmp_path):
    """train -> save -> create_trained_policy.  The stats go to <ckpt>/<step>/assets/<asset_id> (train_pytorch.py:181-184),
    which is where policy_config.create_trained_policy reads them."""
    from kai0_amd.train import Trainer

    G = np.load(os.path.join(HERE, "golden", "host_pipeline.npz"))
    pcfg, ocfg = _cfgs()
    pcfg = dataclasses.replace(pcfg, max_token_len=64)
    ocfg = dataclasses.replace(ocfg, max_token_len=64)
    cfg = tc.TrainConfig(name="tiny_agilex", exp_name="t", model=pcfg, checkpoint_base_dir=str(tmp_path / "ckpt"),
                         data=tc.LerobotAgilexDataConfig(repo_id="tiny_agilex", default_prompt="Flatten and fold the cloth.",
                                                         tokenizer_model=G["tok.model"].tobytes(), use_delta_joint_actions=False))  # fmt: skip
    rng = np.random.default_rng(0)
    q = np.sort(rng.normal(size=(2, 32)), axis=0)
    stats = {k: normalize.NormStats(mean=rng.normal(size=32), std=rng.uniform(0.5, 2, 32), q01=q[0] - 1.5, q99=q[1] + 1.5)
             for k in ("state", "actions")}  # fmt: skip
    data_config = dataclasses.replace(cfg.data.create(tmp_path / "assets", cfg.model), norm_stats=stats)
    tr = Trainer(_stand_in(ocfg), shard_ops=TorchShardOps(), bucket_bytes=64 << 10)
    path = tr.save_checkpoint(str(cfg.checkpoint_dir), data_config=data_config, config=cfg)
    assert os.path.isfile(os.path.join(path, "assets", "tiny_agilex", "norm_stats.json"))
    pol = policy.create_trained_policy(cfg, path, pytorch_device="cpu")  # no norm_stats argument: read from the checkpoint
    got = [t for t in pol._input_transform.transforms if type(t).__name__ == "Normalize"][0].norm_stats["actions"]
    assert np.array_equal(got.q99, stats["actions"].q99) and np.array_equal(got.mean, stats["actions"].mean)
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), pol._model.state_dict().items()):
        # create_trained_policy re-applies to_bfloat16_for_selected_params (policy_config.py:55): like the reference's, it rounds the
        # f32-kept tensors of `paligemma_with_expert` through bf16 on the way; the heads outside it are untouched
        rounded = a.dtype == torch.float32 and k.startswith("paligemma_with_expert.")
        assert a.dtype == b.dtype and torch.equal(a.bfloat16().float() if rounded else a, b), k


def test_optimizer_state_without_entries_for_gradless_parameters_loads():
    """torch.optim.AdamW holds no state for parameters that never received a gradient (in the reference: the last
    layer's prefix o_proj / MLP / post-attention norm and language_model.norm).  Such a file loads: zero moments and the current
    parameter as master for the missing ones — and an entry WITHOUT `master` does not discard the masters other entries carry."""
    from kai0_amd.sharded import ShardedDataParallel

    class TwoBranch(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.a, self.dead = torch.nn.Linear(6, 6), torch.nn.Linear(6, 6)

        def forward(self, x):
            return self.a(x).pow(2).sum()

    ref = TwoBranch()
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0)
    x = torch.linspace(-1, 1, 12).reshape(2, 6)
    ref(x).backward()
    ropt.step()
    sd = ropt.state_dict()
    assert sorted(sd["state"]) == [0, 1]  # a.weight, a.bias only: `dead` never received a gradient
    names = [n for n, _ in ref.named_parameters()]
    m = TwoBranch()
    m.load_state_dict(ref.state_dict())
    eng = ShardedDataParallel(list(m.named_parameters()), world_size=1, rank=0, ops=TorchShardOps(), weight_decay=0.0, max_grad_norm=None)
    for b in eng.buckets:
        b.exp_avg.fill_(7.0), b.exp_avg_sq.fill_(7.0)  # must be overwritten: file values or zeros
    eng.load_state_dict(sd, param_order=names)
    where = {n: eng._where[p] for n, p in m.named_parameters()}
    for n, p in m.named_parameters():
        b, o = where[n]
        ea, ms = b.exp_avg[o : o + p.numel()], b.master[o : o + p.numel()]
        if n.startswith("dead"):
            assert float(ea.abs().max()) == 0.0
        else:
            assert torch.allclose(ea.view(p.shape), sd["state"][names.index(n)]["exp_avg"])
        assert torch.equal(ms.view(p.shape), p.detach().float())
    # mixed file: one entry carries an f32 master that differs from the (bf16-rounded) parameter, another has none
    own = eng.state_dict(names)
    own["state"][0]["master"] = own["state"][0]["master"] + 0.125
    del own["state"][1]["master"]
    eng.load_state_dict(own)
    b, o = where["a.weight"]
    assert torch.equal(b.master[o : o + 36].view(6, 6), m.a.weight.detach() + 0.125)  # kept, not overwritten from the parameters
    b, o = where["a.bias"]
    assert torch.equal(b.master[o : o + 6], m.a.bias.detach())


def test_begin_step_clears_state_left_by_an_aborted_backward():
    from kai0_amd.sharded import ShardedDataParallel

    m = torch.nn.Linear(4, 4)
    eng = ShardedDataParallel(list(m.named_parameters()), world_size=1, rank=0, ops=TorchShardOps())
    m(torch.ones(1, 4)).sum().backward()  # a backward that is never followed by step() (evaluation with gradients)
    assert eng._in_backward
    eng.begin_step()
    assert not eng._in_backward
