import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from kai0_amd.config import Pi0Config
from kai0_amd.train import Trainer
dev = torch.device("cuda:0")
cfg = Pi0Config()
B = int(sys.argv[1])
model = bench.build_model(cfg, dev, seed=0); model.train()
tr = Trainer(model, world_size=1, rank=0, peak_lr=2.5e-5, warmup_steps=1000, decay_steps=30000, end_lr=2.5e-6, weight_decay=1e-10, clip_norm=1.0)
obs, actions = bench.synthetic_batch(cfg, B, seed=1000, device=dev)
torch.cuda.reset_peak_memory_stats()
try:
    for _ in range(2):
        loss = tr.train_step(obs, actions)
    torch.cuda.synchronize()
    print(f"B={B}: peak allocated {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.max_memory_reserved()/2**30:.1f} GiB, loss {float(loss):.5f}")
except Exception as e:
    print(f"B={B}: FAILED {type(e).__name__}: {str(e)[:300]}")
