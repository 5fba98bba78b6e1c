"""Per-kernel averages of rocprofv3 --pmc counters (csv output).  usage: pmc_summary.py <dir with p_counter_collection.csv> [n_top]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1] + "/p_counter_collection.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:80]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
top = sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]
print(f"# rocprofv3 --pmc summary ({sys.argv[1]}): per-kernel totals and per-dispatch averages")
for k, v in top:
    n = len(disp[k])
    print(f"{k}  dispatches={n}")
    for c, x in sorted(v.items()):
        print(f"    {c:28s} total {x:.6g}   per dispatch {x / n:.6g}")
