#!/bin/bash
# usage: kt.sh "<python command>"  -> per-kernel total time
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- $1 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_k/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:70].ljust(70), "calls", r["Calls"], "total_ms", round(float(r["TotalDurationNs"])/1e6,1), "avg_ms", round(float(r["AverageNs"])/1e6,2))
PY
