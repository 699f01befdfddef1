#!/bin/bash
# usage: tools/kernel_trace_list.sh "<python command>" <kernel-name-substring>  -> every launch: grid, ms
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_l
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l -o k -- $1 > /dev/null 2>&1
python - "$2" <<PY
import csv, glob, sys
pat = sys.argv[1]
f = glob.glob("/tmp/prof_l/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
    print(r["Kernel_Name"][:50], "grid", g, "ms", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
PY
