#!/bin/bash
# per-kernel average durations of one bench configuration (rocprofv3 --kernel-trace): tools/kernel_times.sh "<bench args>" [ENV=VALUE ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="$1"; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gram-leg --no-fast-leg --allow-missing-traffic $ARGS > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pk/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "cumf" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(agg.items()):
    if sum(v) / len(v) > 1e5: print(f"{k:72s} grid={g:>10s} n={len(v):3d} avg_ms={sum(v)/len(v)/1e6:.4f}")
PY
