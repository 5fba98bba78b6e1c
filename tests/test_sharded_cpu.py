"""world_size-2 gloo tests of the sharded data-parallel engine (collective / partition logic on CPU).

The shard arithmetic is injected (`TorchShardOps`, a torch restatement of the HIP kernels' math — test-side
oracle); the product default (`HipShardOps`) needs the GPU and is covered by tests/test_model_gpu.py."""

import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class TorchShardOps:
    def sumsq(self, grad, out):
        out += grad.float().pow(2).sum()

    def clip_coef(self, sumsq, max_norm, coef, norm):
        norm.copy_(sumsq.sqrt())
        coef.copy_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))

    def adamw(self, master, m, v, grad, param, *, lr, beta1, beta2, eps, wd, step, clip_coef):
        g = grad.float() * (clip_coef if clip_coef is not None else 1.0)
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        master.mul_(1 - lr * wd)
        denom = v.sqrt() / (1 - beta2**step) ** 0.5 + eps
        master.addcdiv_(m, denom, value=-lr / (1 - beta1**step))
        param.copy_(master.to(param.dtype))


def _toy_model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.Tanh(), torch.nn.Linear(40, 8))
    m[0].weight.data = m[0].weight.data.to(torch.bfloat16).float()  # keep values bf16-representable
    return m


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kai0_amd.sharded import ShardedDataParallel

    model = _toy_model()
    ref = _toy_model()
    eng = ShardedDataParallel(model.parameters(), world_size=world, rank=rank, ops=TorchShardOps(), weight_decay=1e-2,
                              max_grad_norm=0.5, bucket_bytes=1024)
    assert len(eng.buckets) > 1
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-2)
    g = torch.Generator().manual_seed(1)
    data = torch.randn(3, world, 6, 24, generator=g)
    for step in range(3):
        # each rank sees its slice; the reference sees the whole global batch
        model(data[step, rank]).pow(2).mean().backward()
        norm = eng.step(1e-2)
        ref.zero_grad()
        ref(data[step].reshape(-1, 24)).pow(2).mean().backward()
        rn = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        ropt.step()
        assert abs(float(norm) - float(rn)) < 1e-5 * max(1.0, float(rn)), (float(norm), float(rn))
        for p, r in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p, r, atol=2e-6, rtol=1e-5), (step, (p - r).abs().max())
            assert p.grad is None
    # every rank holds the same full parameters, and 1/world of the optimizer state
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    total = sum(p.numel() for p in model.parameters())
    assert eng.optimizer_state_bytes() < 12 * total / world + 12 * 2 * 256 * world
    sd = eng.state_dict()
    eng.load_state_dict(sd)
    if rank == 0:
        open(os.path.join(tmp, "ok"), "w").write("1")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_engine_matches_single_process_adamw():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        assert os.path.exists(os.path.join(tmp, "ok"))


def test_sharded_engine_world1_equals_reference():
    from kai0_amd.sharded import ShardedDataParallel

    model, ref = _toy_model(), _toy_model()
    eng = ShardedDataParallel(model.parameters(), world_size=1, rank=0, ops=TorchShardOps(), weight_decay=0.0, max_grad_norm=1.0)
    ropt = torch.optim.AdamW(ref.parameters(), lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0)
    x = torch.randn(5, 24)
    for _ in range(2):
        model(x).pow(2).mean().backward()
        eng.step(3e-3)
        ref.zero_grad()
        ref(x).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        ropt.step()
    for p, r in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p, r, atol=2e-6, rtol=1e-5)


def test_flat_gradients_of_parameters_without_a_new_gradient_count_as_zero(monkeypatch):
    """The flat gradient buffers are not cleared after a step (producers overwrite their slices): a parameter that got a
    gradient in step t but none in step t+1 must enter step t+1 as zero, not with its old gradient.  Reference: the same
    engine with the full clear after every step (KAI0_ZERO_GRADS=full)."""
    from kai0_amd.sharded import ShardedDataParallel

    class TwoBranch(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.a, self.b = torch.nn.Linear(6, 6), torch.nn.Linear(6, 6)

        def forward(self, x, use_b):
            return self.a(x).sum() + (self.b(x).pow(2).sum() if use_b else 0.0)

    def run(mode):
        if mode:
            monkeypatch.setenv("KAI0_ZERO_GRADS", mode)
        else:
            monkeypatch.delenv("KAI0_ZERO_GRADS", raising=False)
        m = TwoBranch()
        eng = ShardedDataParallel(m.parameters(), world_size=1, rank=0, ops=TorchShardOps(), weight_decay=0.0, max_grad_norm=10.0,
                                  bucket_bytes=64)  # fmt: skip
        norms = []
        x = torch.linspace(-1, 1, 12).reshape(2, 6)
        for use_b in (True, False, False, True):
            m(x, use_b).backward()
            norms.append(float(eng.step(1e-2)))
        return [p.detach().clone() for p in m.parameters()], norms

    lazy, n_lazy = run(None)
    full, n_full = run("full")
    assert n_lazy == n_full and n_lazy[1] < n_lazy[0]  # steps 2, 3 see no gradient for branch b
    assert all(torch.equal(a, b) for a, b in zip(lazy, full))


def test_checkpoint_helpers_roundtrip_with_flat_buffers(tmp_path):
    """Parameters that are views into flat buffers (as in the trainer) still save/load bit-exactly, tied weights once."""
    from safetensors.torch import load_file
    from tiny import build_pair

    from kai0_amd.checkpoint import load_model_safetensors, save_model_safetensors
    from kai0_amd.sharded import ShardedDataParallel

    model, _, _, _ = build_pair("cpu", seed=3)
    ShardedDataParallel(model.parameters(), world_size=1, rank=0, ops=TorchShardOps())
    path = str(tmp_path / "model.safetensors")
    save_model_safetensors(model, path)
    raw = load_file(path)
    pw = "paligemma_with_expert.paligemma."
    assert (pw + "lm_head.weight" in raw) != (pw + "model.language_model.embed_tokens.weight" in raw)  # tied: stored once
    model2, _, _, _ = build_pair("cpu", seed=4)
    load_model_safetensors(model2, path)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert a.dtype == b.dtype and torch.equal(a, b), k
