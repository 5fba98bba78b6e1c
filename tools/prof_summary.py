"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish).
usage: python tools/prof_summary.py gpurun_out/prof/r1_results.db [n_steps_total] > profiles/xxx.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:110]


print(f"# rocprofv3 --kernel-trace --stats summary ({sys.argv[1]})")
print(f"total kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, c, t, a, mn, mx in rows[:60]:
    print(f"| `{short(n)}` | {c} | {t / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / total:.2f} |")
