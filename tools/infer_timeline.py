"""Timeline of the last action chunk in a rocprofv3 kernel-trace db: phases, per-kernel totals, gaps.
usage: python tools/infer_timeline.py gpurun_out/prof_inf/inf_results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "im2col" in r[0]]
seg = rows[idx[-1]:]
names = [r[0] for r in seg]
print(len(seg), "kernels; span ms", (seg[-1][2] - seg[0][1]) / 1e6, "busy", sum(r[2] - r[1] for r in seg) / 1e6)
c, t = collections.Counter(), collections.Counter()
for r in seg:
    k = r[0].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    c[k] += 1
    t[k] += r[2] - r[1]
for k, v in sorted(t.items(), key=lambda x: -x[1])[:22]:
    print(f"{k:66s} {c[k]:5d} {v / 1e6:7.3f} ms {v / c[k] / 1e3:6.1f} us")
e = [i for i, n in enumerate(names) if "euler" in n]
i0 = next(i for i, n in enumerate(names) if "rmsnorm_fwd" in n)
print("siglip ms", (seg[i0][1] - seg[0][1]) / 1e6, "kernels", i0)
if e:
    print("to first euler ms", (seg[e[0]][2] - seg[0][1]) / 1e6, "kernels", e[0])
    print("per step ms", [round((seg[e[i + 1]][2] - seg[e[i]][2]) / 1e6, 3) for i in range(len(e) - 1)], "kernels/step", e[1] - e[0] if len(e) > 1 else 0)
    a, b = e[0], e[1]
    for r0, r1 in zip(seg[a + 1:b + 1][:60], seg[a + 2:b + 2][:60]):
        print(f"   {r0[0].replace('(anonymous namespace)::', '')[:50]:52s} dur {(r0[2] - r0[1]) / 1e3:6.1f} us  gap {(r1[1] - r0[2]) / 1e3:6.1f}")


def dump(title, a, n):
    print(title)
    for r in seg[a : a + n]:
        print(f"   {r[0].replace('(anonymous namespace)::', '').replace('void ', '')[:64]:66s} dur {(r[2] - r[1]) / 1e3:6.1f} us")


dump("SigLIP tower, first kernels:", 0, 26)
dump("prefix pass, first kernels after the tower:", i0, 30)
