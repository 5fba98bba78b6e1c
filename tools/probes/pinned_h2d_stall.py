"""Does the DeviceFeeder (pinned memory + non_blocking H2D on a side stream) show the periodic synchronisation stall the serve path's
pinned staging showed?  B = 32 batches of FakeDataset through DeviceFeeder; a ~15 ms GPU 'step' per batch; per-iteration wall time with a
device synchronisation, pinned (default) vs the feeder patched to pageable copies."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kai0_amd import data_loader as dl
from kai0_amd.config import Pi0Config
dev = torch.device("cuda:0")
cfg = Pi0Config()
w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
def step():
    x = w
    for _ in range(12): x = (x @ w) * 1e-4
    return x
_ds = dl.FakeDataset(cfg, 4096)
_batches = list(dl.create_torch_data_loader(_ds, 32, num_batches=6))  # pre-collated CPU batches: the loader's own cost is out of the picture
class Cycle:
    def __iter__(self):
        for i in range(60): yield _batches[i % len(_batches)]
def run(label):
    feeder = dl.DeviceFeeder(Cycle(), dev, depth=2, pin=PIN)
    ts = []
    t0 = time.perf_counter()
    for i, (obs, actions) in enumerate(feeder):
        step(); (torch.cuda.synchronize() if SYNC == 'device' else torch.cuda.current_stream().synchronize())
        t1 = time.perf_counter(); ts.append((t1 - t0) * 1e3); t0 = t1
    print(label, " ".join(f"{t:.0f}" for t in ts[2:]), flush=True)
step(); torch.cuda.synchronize()
for sync in ("device", "stream"):
    SYNC = sync
    for PIN in (True, False):
        run(f"{'pin_memory + non_blocking' if PIN else 'pageable, blocking      '} | {sync} synchronise per step: ")
