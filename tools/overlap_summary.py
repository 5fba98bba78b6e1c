"""How much of a kernel trace ran concurrently: per training step, the sum of kernel durations against the union of their
intervals, and which kernels the overlapped time belongs to (rocprofv3 --kernel-trace rocpd database).  RCCL kernels
(ncclDevKernel* / rccl* / mscclKernel*: the reduce-scatters, all-gathers and the norm all-reduce of kai0_amd.sharded) are
classified apart: per step their busy time, how much of it ran while a compute kernel was running (hidden) and how much with
no compute kernel on the chip (exposed) — the timeline evidence SURVEY.md §8e asks for at N > 1 (run one rank's trace).
usage: python tools/overlap_summary.py <results.db> [n_last_steps]"""
import collections
import re
import sqlite3
import sys

COMM = re.compile(r"nccl|rccl|msccl", re.I)


def is_comm(name: str) -> bool:
    return bool(COMM.search(name))


def union_len(iv):
    tot, end = 0, None
    for a, b in sorted(iv):
        if end is None or a > end:
            tot += b - a
            end = b
        elif b > end:
            tot += b - end
            end = b
    return tot


def intersect_len(x, y):
    """total length of (union of x) ∩ (union of y)"""
    def merged(iv):
        out = []
        for a, b in sorted(iv):
            if out and a <= out[-1][1]:
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        return out

    mx, my, i, j, tot = merged(x), merged(y), 0, 0, 0
    while i < len(mx) and j < len(my):
        lo, hi = max(mx[i][0], my[j][0]), min(mx[i][1], my[j][1])
        if lo < hi:
            tot += hi - lo
        if mx[i][1] < my[j][1]:
            i += 1
        else:
            j += 1
    return tot


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
    comm_iv = [(r[1], r[2]) for r in seg if is_comm(r[0])]
    comp_iv = [(r[1], r[2]) for r in seg if not is_comm(r[0])]
    if comm_iv:
        busy, hidden = union_len(comm_iv), intersect_len(comm_iv, comp_iv)
        print(f"      RCCL: {len(comm_iv)} kernels, busy {busy / 1e6:7.2f} ms; compute∩comm {hidden / 1e6:7.2f} ms hidden behind compute kernels, "
              f"{(busy - hidden) / 1e6:6.2f} ms with no compute kernel running (exposed); compute busy {union_len(comp_iv) / 1e6:7.2f} ms")
