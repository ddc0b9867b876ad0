#!/usr/bin/env python3
"""Instruction mix of one kernel in hipcc -S output: python tools/isa_stats.py file.s <substring of the kernel symbol>"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m and pat in m.group(1) and "rocprim" not in m.group(1):
        start = i; name = m.group(1); break
if start is None: sys.exit("kernel not found")
cls = collections.Counter(); n = 0
for l in lines[start + 1:]:
    if l.startswith("\t.section") or re.match(r"^\.Lfunc_end", l): break
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
    op = t[0]; n += 1
    k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cls[k] += 1
print(name[:80], "static instructions:", n, dict(cls))
for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill", "lds_size", "Occupancy", "ScratchSize"):
    for l in lines[start:]:
        if key in l and l.strip().startswith((";", ".")):
            print("  ", l.strip()); break
