"""How much of a kernel trace ran concurrently: per training step, the sum of kernel durations against the union of their
intervals, and which kernels the overlapped time belongs to (rocprofv3 --kernel-trace rocpd database).
usage: python tools/overlap_summary.py <results.db> [n_last_steps]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
extra = [c for c in ("stream_id", "queue_id") if c in cols]
rows = db.execute(f"select name, start, end{''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
# steps are delimited by the AdamW launches over the big buckets (one run of them per step)
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
bounds = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]  # last AdamW launch of each step
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = list(zip(bounds[:-1], bounds[1:]))[-n_last:]
print(f"# {sys.argv[1]}: {len(rows)} kernels, {len(bounds)} steps found; last {len(steps)} analysed")
for a, b in steps:
    seg = rows[a + 1 : b + 1]
    span = (seg[-1][2] - seg[0][1]) / 1e6
    total = sum(r[2] - r[1] for r in seg) / 1e6
    # union of intervals + time covered by >= 2 kernels, attributed to the SHORTER kernel of each overlap
    ev = sorted([(r[1], 1, i) for i, r in enumerate(seg)] + [(r[2], -1, i) for i, r in enumerate(seg)])
    active, last_t, union, multi = set(), None, 0, 0
    over = collections.Counter()
    for t, d, i in ev:
        if last_t is not None and active:
            dt = t - last_t
            union += dt
            if len(active) >= 2:
                multi += dt
                short = min(active, key=lambda j: seg[j][2] - seg[j][1])
                over[re.sub(r"\(anonymous namespace\)::|void ", "", seg[short][0])[:60]] += dt
        last_t = t
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
    streams = collections.Counter(tuple(r[3:]) for r in seg) if extra else {}
    print(f"step: span {span:7.2f} ms  sum of kernel durations {total:7.2f}  union {union / 1e6:7.2f}  >= 2 kernels running {multi / 1e6:6.2f} ms"
          f"  idle {span - union / 1e6:5.2f}  kernels {len(seg)}  {dict(streams) if extra else ''}")
    for k, v in over.most_common(8):
        print(f"      overlapped (shorter kernel) {v / 1e6:6.2f} ms  {k}")
