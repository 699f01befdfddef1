#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import csv, sys, collections, glob, os
path = sys.argv[1]
files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    with open(fn) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "cumf" not in k:
                continue
            k = k.split("(")[0][-60:]
            agg[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for k, cs in agg.items():
    print(k)
    for c, vals in sorted(cs.items()):
        per = collections.defaultdict(float)
        for d, v in vals:
            per[d] += v
        xs = list(per.values())
        print(f"   {c:36s} n={len(xs):3d} mean={sum(xs)/len(xs):.6g} min={min(xs):.6g} max={max(xs):.6g}")
