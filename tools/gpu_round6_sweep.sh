#!/bin/bash
# round 6: one bench line per BASELINE configuration (5 steps each, no CPU legs) -> gpurun_out/r06/sweep_<tag>.txt
# usage: tools/gpu_round6_sweep.sh <tag> [extra env assignments are inherited, e.g. CUMF_ALS_PRESPLIT=0]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06; mkdir -p $O
TAG=${1:-sweep}
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic"
for cfg in "--f 100 --solver lu" "--f 100 --solver cg" "--f 64 --solver lu" "--f 64 --solver cg" "--f 128 --solver cg" "--f 128 --solver lu" "--f 200 --solver cg" "--f 200 --solver lu" "--shape hugewiki --scheme reduce --solver cg" "--shape hugewiki --scheme reduce --solver lu" "--shape hugewiki --scheme reduce --solver cg --reference-solvers" "--shape ml10m --f 10 --solver lu" "--shape ml10m --f 10 --solver cg"; do
  python bench.py --steps 5 --warmup 1 $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg: ms', round(d['ms_per_step'],3), 'G ratings/s', round(d['value']/1e9,3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), 'dominant', r['dominant'], 'bound', r['bound'], 'frac', round(r['frac'],3), 'x:', r['x_side']['kernel'], 'theta:', r['theta_side']['kernel'])"
done > $O/sweep_$TAG.txt 2>&1
cat $O/sweep_$TAG.txt
