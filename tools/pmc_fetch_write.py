#!/usr/bin/env python3
"""Merge the two rocprofv3 PMC passes of one bench.py command (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with
--kernel-trace only) into the per-kernel table bench.py reads (profiles/*pmc_fetch_write.csv):
kernel, launches, FETCH_SIZE and WRITE_SIZE per launch (counter units: KiB), average duration in the fetch pass.
usage: pmc_fetch_write.py <fetch counter_collection.csv glob> <write counter_collection.csv glob> > out.csv"""
import csv, glob, sys, collections

def short(name):
    n = name.split("(")[0]
    if "radix_sort" in n or "onesweep" in n or "histogram" in n: return "rocprim:radix_sort"
    if "rocprim" in n and "scan" in n: return "rocprim:scan"
    if "rocprim" in n: return "rocprim:other"
    return n.replace("void ", "").split("<")[0]

def load(pattern, counter):
    tot = collections.defaultdict(float); n = collections.Counter(); ms = collections.defaultdict(float)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"]); n[k] += 1
            ms[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return tot, n, ms

ft, fn, fms = load(sys.argv[1], "FETCH_SIZE")
wt, wn, _ = load(sys.argv[2], "WRITE_SIZE")
print("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,avg_ms_in_fetch_pass")
for k in sorted(ft, key=lambda k: -fms[k]):
    print("%s,%d,%.0f,%.0f,%.3f" % (k, fn[k], ft[k] / fn[k], (wt.get(k, 0.0) / wn[k]) if wn.get(k) else 0.0, fms[k] / fn[k]))
