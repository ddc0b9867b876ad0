#!/usr/bin/env python3
"""GPU timeline of a rocprofv3 --kernel-trace run: for every kernel its duration and the idle gap in front of it
(time since the previous kernel ended).  usage: trace_gaps.py <kernel_trace.csv> [min_gap_ms]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
rows.sort()
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t0 = rows[0][0]; prev_end = rows[0][0]; busy = 0
for s, e, k in rows:
    gap = (s - prev_end) / 1e6; dur = (e - s) / 1e6; busy += e - max(s, prev_end) if e > prev_end else 0
    if gap >= mingap or dur >= 5: print("t=%9.2f ms  gap %7.2f  dur %8.2f  %s" % ((s - t0) / 1e6, gap, dur, k))
    prev_end = max(prev_end, e)
print("span %.1f ms, GPU busy %.1f ms" % ((prev_end - t0) / 1e6, busy / 1e6))
