#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter over the dispatches."""
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter()
for path in sys.argv[1:]:
    for f in glob.glob(path, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            d = (k, r.get("Dispatch_Id"))
            if d not in seen: seen.add(d); nd[k] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    print(k, "dispatches=%d" % nd[k], " ".join("%s=%.4g" % (c, v) for c, v in sorted(agg[k].items())))
