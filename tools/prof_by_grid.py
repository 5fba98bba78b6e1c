"""per (kernel, grid) stats from a rocprofv3 rocpd db: usage prof_by_grid.py db [name substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
gcols = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size", "workgroup_size_x", "grid_z", "grid_size_z")]
sel = ", ".join(gcols) if gcols else "''"
q = f"select name, {sel}, count(*), avg(end-start), min(end-start) from kernels where name like ? group by name, {sel} order by 1"
for r in cur.execute(q, (f"%{sub}%",)):
    print(r[0][:60].replace("(anonymous namespace)::", ""), r[1:-3], "calls", r[-3], "avg us %.1f min %.1f" % (r[-2] / 1e3, r[-1] / 1e3))
if not gcols:
    print("columns:", cols)
