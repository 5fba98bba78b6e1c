"""Does the HBM-bound AdamW pass hide behind compute-bound GEMMs when it runs on a second (low-priority) stream?
Times a GEMM sequence alone, the AdamW pass alone, and both together.  usage: python tools/overlap_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kai0_amd import _lib, optim
from kai0_amd.ops import gemm
dev = torch.device("cuda:0"); BF16 = torch.bfloat16; F32 = torch.float32
lib = _lib.load()
n = 1 << 28  # 268 M parameters per "bucket", 4 buckets ~ 1.07 G parameters (a third of the model): ~ 5.5 ms
NB = 4
bk = [dict(master=torch.zeros(n, dtype=F32, device=dev), m=torch.zeros(n, dtype=F32, device=dev), v=torch.zeros(n, dtype=F32, device=dev),
           g=torch.zeros(n, dtype=BF16, device=dev), p=torch.zeros(n, dtype=BF16, device=dev)) for _ in range(NB)]

def adam():
    for b in bk:
        optim.adamw_step_(b["master"], b["m"], b["v"], b["g"], b["p"], lr=1e-5, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.0, step=3)

def mk(M, N, K):
    A = (torch.randn(M, K, device=dev)).to(BF16); W = (torch.randn(N, K, device=dev) * 0.03).to(BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    return lambda: gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)

seqs = {"siglip (24576x4304x1152, 1152x4304) x20": [mk(24576, 4304, 1152), mk(24576, 1152, 4304)] * 20,
        "gemma qkv/o (30976x2560x2048, 2048x2048) x20": [mk(30976, 2560, 2048), mk(30976, 2048, 2048)] * 20,
        "gemma mlp persistent (30976x16384x2048) x6": [mk(30976, 16384, 2048)] * 6}

def wall(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3

for prio in (0, -1):
    # torch: lower number = higher priority; the side stream gets the LOWER priority when main is raised instead
    side = torch.cuda.Stream(device=dev, priority=0)
    main = torch.cuda.Stream(device=dev, priority=prio)
    for name, seq in seqs.items():
        def gemms():
            with torch.cuda.stream(main):
                for f in seq: f()
        def adam_side():
            with torch.cuda.stream(side):
                adam()
        def both():
            adam_side(); gemms()
        for _ in range(2): gemms(); adam_side()
        tg = min(wall(gemms) for _ in range(3)); ta = min(wall(adam_side) for _ in range(3)); tb = min(wall(both) for _ in range(3))
        print(f"main priority {prio}: {name:48s} gemms {tg:6.2f} ms  adamw {ta:5.2f} ms  together {tb:6.2f} ms  (sum {tg + ta:6.2f}, hidden {tg + ta - tb:5.2f})", flush=True)
