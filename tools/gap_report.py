#!/usr/bin/env python3
"""Idle time of the GPU between consecutive kernels, from a rocprofv3 --kernel-trace CSV:
   python tools/gap_report.py <kernel_trace.csv> [first_kernel_of_a_step] [steps_from_the_end]
prints, for the last steps (a step starts at the named kernel, default k_table_words), busy / idle time and the timeline of the
last step: every kernel with the idle time in front of it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_table_words"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]) for r in rows))
starts = [i for i, e in enumerate(ev) if e[2].startswith(first)]
if len(starts) < nsteps + 1: sys.exit("not enough steps in the trace")
lo, hi = starts[-nsteps - 1], starts[-1]           # whole steps only
seg = ev[lo:hi]
busy = sum(e - s for s, e, _ in seg); span = ev[hi][0] - seg[0][0]
print("%d steps: %.2f ms per step, busy %.2f, idle %.2f ms per step (%.1f %%)" % (nsteps, span / nsteps / 1e6, busy / nsteps / 1e6, (span - busy) / nsteps / 1e6, 100.0 * (span - busy) / span))
gaps = collections.Counter(); cnt = collections.Counter()
for k in range(lo + 1, hi + 1):
    g = ev[k][0] - ev[k - 1][1]
    if g > 0: gaps[(ev[k - 1][2], ev[k][2])] += g; cnt[(ev[k - 1][2], ev[k][2])] += 1
for (a, b), g in gaps.most_common(12):
    print("  %7.3f ms per step in %4.1f gaps per step (%5.0f us each)  %s -> %s" % (g / nsteps / 1e6, cnt[(a, b)] / nsteps, g / cnt[(a, b)] / 1e3, a, b))
print("timeline of the last whole step (idle us in front, kernel, duration us):")
l2 = starts[-2]
for k in range(l2, hi + 1):
    g = (ev[k][0] - ev[k - 1][1]) / 1e3
    print("  %8.0f  %-36s %9.0f" % (g, ev[k][2], (ev[k][1] - ev[k][0]) / 1e3))
