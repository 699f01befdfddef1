#!/usr/bin/env python3
"""profiles/<round>/*/traffic.json -> profiles/traffic.json = {"kernels": [entry, ...]}, one entry per dispatched
Gram kernel (later directories win for the same kernel name).  bench.py looks its dispatched kernel up by name.
  python tools/merge_traffic.py profiles/r03"""
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
entries = {}
for rnd in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(rnd, "*", "traffic.json"))):
        try:
            ent = json.load(open(path))
        except ValueError:
            continue
        if "kernel" in ent and "bytes_per_launch" in ent:
            entries[ent["kernel"]] = ent
json.dump({"kernels": list(entries.values())}, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print("profiles/traffic.json:", list(entries))
