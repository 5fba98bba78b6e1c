"""Per-shape GEMM time inside one B=1 action chunk (eager launches, HIP events)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KAI0_INFER_GRAPH"] = "0"
import bench
from kai0_amd.config import Pi0Config
dev = torch.device("cuda:0")
cfg = Pi0Config()
model = bench.build_model(cfg, dev, 0).eval()
obs, _ = bench.synthetic_batch(cfg, 1, seed=123, device=dev)
noise = torch.randn(1, cfg.action_horizon, cfg.action_dim, device=dev)
for _ in range(2):
    model.sample_actions(dev, obs, noise=noise)
t = bench.GemmTimer(); t.install()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); model.sample_actions(dev, obs, noise=noise); e.record(); torch.cuda.synchronize(); t.uninstall()
rows = t.breakdown(); tot = sum(r["ms"] for r in rows)
print(f"chunk {s.elapsed_time(e):.2f} ms (eager, with event overhead); bf16 GEMM total {tot:.2f} ms over {sum(r['calls'] for r in rows)} launches")
for r in rows[:30]:
    lay = {(1, 1): "NT", (1, 0): "NN", (0, 0): "TN", (0, 1): "TNa"}[(r["a_kc"], r["b_kc"])]
    print(f"{lay} M={r['M']:6d} N={r['N']:6d} K={r['K']:6d} b={r['batch']:3d} calls={r['calls']:4d} ms={r['ms']:7.3f} avg_us={1e3*r['ms']/r['calls']:7.1f} {r['tflops']:7.1f} TF/s")
