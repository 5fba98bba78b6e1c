"""Build libkai0hip.so in-tree: `python -m kai0_amd.build` (hipcc cross-compiles gfx950 without a GPU)."""

from __future__ import annotations

import pathlib
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
SRC = ["runtime.hip", "gemm_bf16.hip", "gemm_f32.hip", "attention.hip", "skinny.hip", "attn_decode.hip", "attn_bwd_siglip.hip", "attn_bwd.hip", "norm.hip", "elementwise.hip", "optim.hip"]
OUT = HERE / "lib" / "libkai0hip.so"


# -ffp-contract=off: torch's eager ops never fuse a multiply into a following add, and hipcc's default
# (-ffp-contract=fast) even contracts ACROSS an explicit bf16 round trip (it narrows fpext*fpext products to
# bf16 ops and then forms a bf16 fma), which silently removes the bf16 rounding points this library emulates.
def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    srcs = [HERE / "csrc" / s for s in SRC]
    deps = srcs + [HERE / "csrc" / "common.h", HERE.parent / "include" / "kai0hip.h"]
    if not force and OUT.exists() and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return OUT
    OUT.parent.mkdir(parents=True, exist_ok=True)
    import os

    extra = os.environ.get("KAI0_HIPCC_FLAGS", "").split()  # e.g. -DKAI0_SK2_TRACE (tools/probes/sk2_phases.py)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", *extra,
           *map(str, srcs), "-o", str(OUT)]  # fmt: skip
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
