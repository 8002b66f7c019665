#!/usr/bin/env python3
"""Basic-block instruction counts of one kernel in a -save-temps .s file (VALU / SALU / LDS / VMEM per block, with the
branch targets), to see where a VALU-bound kernel spends its issue slots.  usage: isa_blocks.py file.s kernel_substring"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0])
blocks, cur = [], ["entry", []]
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur[1].append(s.split(";")[0].strip())
blocks.append(cur)
tot = dict(v=0, s=0, ds=0, vm=0)
for name, ins in blocks:
    v = sum(1 for i in ins if i.startswith("v_"))
    s = sum(1 for i in ins if i.startswith("s_") and not i.startswith("s_waitcnt") and not i.startswith("s_nop"))
    ds = sum(1 for i in ins if i.startswith("ds_"))
    vm = sum(1 for i in ins if i.startswith("global_") or i.startswith("buffer_") or i.startswith("flat_"))
    br = [i.split()[-1] for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
    tot["v"] += v; tot["s"] += s; tot["ds"] += ds; tot["vm"] += vm
    print(f"{name:12s} valu {v:4d} salu {s:4d} lds {ds:3d} vmem {vm:3d}  -> {' '.join(br)}")
print("total", tot)
