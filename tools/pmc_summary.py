#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per (kernel, grid size): mean per dispatch.
The X-side and Theta-side launches of a half-iteration kernel differ in grid size, so they come
out as separate rows (report the two sides separately)."""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for fn in files:
    with open(fn) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "cumf" not in k:
                continue
            key = (k.split("(")[0][-70:], row.get("Grid_Size", "?"))
            agg[key][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
for (k, grid), cs in sorted(agg.items()):
    print(f"{k}  grid={grid}")
    for c, per in sorted(cs.items()):
        xs = list(per.values())
        print(f"   {c:36s} n={len(xs):3d} mean={sum(xs) / len(xs):.6g} min={min(xs):.6g} max={max(xs):.6g}")
