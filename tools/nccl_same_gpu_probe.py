"""Single-GPU validation of the RCCL call pattern of ShardedDataParallel (reduce_scatter_tensor AVG issued from the
backward, in-place all_gather_into_tensor, async work handles).  RCCL refuses two ranks on one device ("Duplicate GPU
detected"), so this runs ONE rank with KAI0_FORCE_COLLECTIVES=1 and compares against the collective-free engine.
usage: KAI0_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_same_gpu_probe.py"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", init_method="env://")
from kai0_amd.sharded import ShardedDataParallel
torch.manual_seed(0)
ps = [torch.nn.Parameter(torch.randn(257, 64, device=dev).bfloat16()), torch.nn.Parameter(torch.randn(1000, device=dev)),
      torch.nn.Parameter(torch.randn(64, 64, device=dev).bfloat16())]
ref = [p.detach().clone() for p in ps]
eng = ShardedDataParallel(ps, world_size=world, rank=rank, bucket_bytes=16384)
assert eng.collectives and eng.backend == "nccl", (eng.collectives, eng.backend)
os.environ.pop("KAI0_FORCE_COLLECTIVES")
ps2 = [torch.nn.Parameter(r.clone()) for r in ref]
eng2 = ShardedDataParallel(ps2, world_size=1, rank=0, bucket_bytes=16384)  # plain single-rank engine
assert not eng2.collectives
for step in range(3):
    for e, pp in ((eng, ps), (eng2, ps2)):
        loss = sum((p.float() * (1 + i) * (step + 1)).pow(2).sum() for i, p in enumerate(pp))
        loss.backward()
        n = e.step(1e-2)
torch.cuda.synchronize()
for p, q, r in zip(ps, ps2, ref):
    assert torch.equal(p.detach(), q.detach()), "RCCL path and single-rank path disagree"
    assert not torch.equal(p.detach(), r)
print(f"rank {rank}: RCCL reduce-scatter(AVG)/all-gather path == single-rank path, grad norm {float(n):.4f}", flush=True)
dist.destroy_process_group()
