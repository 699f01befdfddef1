#!/usr/bin/env python3
"""Effective shader clock per kernel = GRBM_GUI_ACTIVE (per XCD) / kernel duration, from one
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE run (csv)."""
import csv, sys, glob, os, collections
path = sys.argv[1]
dur = {}
for fn in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        dur[int(row["Dispatch_Id"])] = (row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
cyc = collections.defaultdict(float)
for fn in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[int(row["Dispatch_Id"])] += float(row["Counter_Value"])
for d, (name, ns) in sorted(dur.items()):
    if "cumf" in name and d in cyc:
        print(f"{name[:70]:70s} {ns/1e6:8.3f} ms  GUI_ACTIVE={cyc[d]:.4g}  clock(if /8 XCD)={cyc[d]/8/ns:.3f} GHz")
