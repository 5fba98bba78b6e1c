"""Build libkai0hip.so in-tree: `python -m kai0_amd.build` (hipcc cross-compiles gfx950 without a GPU)."""

from __future__ import annotations

import pathlib
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
SRC = ["runtime.hip", "gemm_bf16.hip", "gemm_f32.hip", "attention.hip", "skinny.hip", "attn_decode.hip", "attn_siglip.hip", "attn_bwd.hip", "norm.hip", "elementwise.hip", "optim.hip"]
OUT = HERE / "lib" / "libkai0hip.so"


# -ffp-contract=off: torch's eager ops never fuse a multiply into a following add, and hipcc's default
# (-ffp-contract=fast) even contracts ACROSS an explicit bf16 round trip (it narrows fpext*fpext products to
# bf16 ops and then forms a bf16 fma), which silently removes the bf16 rounding points this library emulates.
def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    """Compile every csrc/*.hip to an object (in parallel, only the stale ones) and link libkai0hip.so."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    srcs = [HERE / "csrc" / s for s in SRC]
    common = [HERE / "csrc" / "common.h", HERE.parent / "include" / "kai0hip.h", pathlib.Path(__file__)]
    extra = os.environ.get("KAI0_HIPCC_FLAGS", "").split()  # e.g. -DKAI0_SK2_TRACE (tools/probes/sk2_phases.py)
    objdir = OUT.parent / ("obj" + ("_" + "_".join(f.strip("-") for f in extra) if extra else ""))
    objdir.mkdir(parents=True, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", *extra]
    newest_common = max(d.stat().st_mtime for d in common)

    def compile_one(src: pathlib.Path) -> tuple[pathlib.Path, bool]:
        obj = objdir / (src.stem + ".o")
        if not force and obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, newest_common):
            return obj, False
        cmd = ["hipcc", *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        results = list(pool.map(compile_one, srcs))
    objs = [o for o, _ in results]
    # which flag set the library on disk was linked from (a KAI0_HIPCC_FLAGS build leaves up-to-date objects of the DEFAULT build behind it:
    # without this stamp the next default build would find nothing stale and keep the diagnostic library)
    stamp = OUT.parent / "linked_with_flags.txt"
    same_flags = stamp.exists() and stamp.read_text() == " ".join(extra)
    if not force and OUT.exists() and same_flags and not any(c for _, c in results) and all(OUT.stat().st_mtime >= o.stat().st_mtime for o in objs):
        return OUT
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", *map(str, objs), "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(" ".join(extra))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
