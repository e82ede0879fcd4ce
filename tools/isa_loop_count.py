#!/usr/bin/env python3
"""Count the vector instructions of a kernel's loops in a device assembly listing (hipcc --cuda-device-only -S).
usage: tools/isa_loop_count.py <file.s> <mangled-kernel-name>     one line per loop (innermost first by size); the hot loop
of the DTW kernels is the largest one without a slow-phase duplicate.  Instructions inside blocks that the loop jumps OVER on
its hot path (the out-of-line threshold recompute) are excluded when the listing places them after the back-edge."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(name + ":"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
for a, b in sorted(loops, key=lambda ab: ab[1] - ab[0]):
    ins = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    grp = collections.Counter()
    for k, v in c.items():
        if not k.startswith("v_"):
            grp["salu/mem/other"] += v
        elif "dpp" in k:
            grp["dpp"] += v
        elif k.startswith(("v_fmac", "v_mul_f32", "v_fma_f32", "v_pk_")):
            grp["mul/fma"] += v
        elif k.startswith(("v_add_f32", "v_sub_f32")):
            grp["add"] += v
        elif k.startswith("v_min3"):
            grp["min3"] += v
        elif k.startswith("v_cmp"):
            grp["cmp"] += v
        elif k.startswith("v_cndmask"):
            grp["cndmask"] += v
        elif k.startswith("v_sqrt"):
            grp["sqrt"] += v
        else:
            grp["valu other"] += v
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print("lines %6d-%6d  instr %5d  VALU %5d  %s" % (a, b, len(ins), valu, dict(grp)))
