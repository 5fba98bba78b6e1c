"""Per-kernel register / spill / scratch / occupancy report from hipcc (-Rpass-analysis=kernel-resource-usage), no GPU needed.
usage: python tools/kernel_resources.py attention attn_bwd ..."""
import re
import subprocess
import sys

for name in sys.argv[1:]:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", f"kai0_amd/csrc/{name}.hip",
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS Size"):
            nm = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            print(f"{nm[:110]:110s} vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} spill {cur.get('VGPRs Spill'):>4} scratch {cur.get('ScratchSize'):>5} occ {cur.get('Occupancy')}")
