"""Where one steady training step's wall time goes, from a rocprofv3 --kernel-trace database: the step (between two embed_grad
kernels) is cut into intervals at every kernel start / end; an interval is charged to 'gemm' when a bf16 GEMM kernel is running in it,
else to the (first) kernel that is running, else to 'idle'.  Prints the exposed (non-GEMM-covered) time per kernel name with launch
counts, i.e. what shortening or removing a kernel can actually give back.
usage: python tools/step_anatomy.py trace.db [steps_back=1]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "embed_grad" in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seg = rows[idx[-1 - back] : idx[-back]]
t0, t1 = seg[0][1], max(r[2] for r in seg)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:70]


ev = []
for i, (n, s, e) in enumerate(seg):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
running = set()
exposed = collections.Counter()
count = collections.Counter(short(r[0]) for r in seg)
total = collections.Counter()
for n, s, e in seg:
    total[short(n)] += e - s
prev = t0
for t, kind, i in ev:
    if t > prev:
        dt = t - prev
        if not running:
            exposed["(idle)"] += dt
        elif any("gemm_bf16_kernel" in seg[j][0] for j in running):
            exposed["(bf16 GEMM running)"] += dt
        else:
            exposed[short(seg[min(running)][0])] += dt
        prev = t
    if kind == 1:
        running.add(i)
    else:
        running.discard(i)
span = (t1 - t0) / 1e6
print(f"step span {span:.1f} ms, {len(seg)} kernels")
print(f"{'kernel':72s} {'launches':>8s} {'total ms':>9s} {'exposed ms':>10s}")
for k, v in exposed.most_common(45):
    print(f"{k:72s} {count.get(k, 0):8d} {total.get(k, 0) / 1e6:9.2f} {v / 1e6:10.2f}")
