"""GPU idle gaps inside one training step of a rocprofv3 kernel-trace db: total, and grouped by the kernel that FOLLOWS the gap."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "embed_grad" in r[0]]
seg = rows[idx[-2]:idx[-1]]
span = (seg[-1][2] - seg[0][1]) / 1e6
busy = sum(r[2] - r[1] for r in seg) / 1e6
print(f"step span {span:.1f} ms, busy {busy:.1f} ms, idle {span - busy:.1f} ms, kernels {len(seg)}")
g = collections.Counter(); c = collections.Counter()
big = 0.0
for a, b in zip(seg, seg[1:]):
    gap = b[1] - a[2]
    if gap > 0:
        k = b[0].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        g[k] += gap; c[k] += 1
        if gap > 20000: big += gap
print(f"gaps > 20 us total {big / 1e6:.2f} ms")
for k, v in g.most_common(14):
    print(f"{k:62s} n={c[k]:5d} {v / 1e6:7.3f} ms avg {v / c[k] / 1e3:6.1f} us")
