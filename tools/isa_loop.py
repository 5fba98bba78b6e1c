"""Condensed instruction-category sequence of the hottest basic block (most MFMAs) of a kernel in a hipcc -S listing.
M mfma, E v_exp, r ds_read, W ds_write, p ds_bpermute/swizzle, D buffer/global load, S store, w s_waitcnt, B s_barrier, v other VALU, s SALU.
usage: python tools/isa_loop.py file.s <substring of the mangled kernel name> [--all]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sys.argv[2]) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
blocks, cur = [], []
for l in body:
    if re.match(r"^\.LBB", l):
        blocks.append(cur)
        cur = [l]
    else:
        cur.append(l)
blocks.append(cur)


def cat(x):
    x = x.strip()
    for pre, c in (("v_mfma", "M"), ("v_exp", "E"), ("ds_read", "r"), ("ds_write", "W"), ("ds_bpermute", "p"), ("ds_swizzle", "p"),
                   ("buffer_load", "D"), ("global_load", "D"), ("buffer_store", "S"), ("global_store", "S"), ("s_waitcnt", "w"),
                   ("s_barrier", "B"), ("scratch_", "!"), ("v_", "v"), ("s_", "s")):
        if x.startswith(pre):
            return c
    return ""


sel = blocks if "--all" in sys.argv else [max(blocks, key=lambda b: sum("v_mfma" in x for x in b))]
for b in sel:
    seq = "".join(cat(x) for x in b)
    if not seq:
        continue
    print(b[0][:40], "instr", len(seq), {k: seq.count(k) for k in "MErWpDSwB!v"})
    print(re.sub(r"(.{160})", r"\1\n", seq))
