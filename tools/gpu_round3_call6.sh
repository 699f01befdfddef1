#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=10 > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log; grep "^FAILED" gpurun_out/gpu_tests.log
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic"
for cfg in "--f 100 --solver lu" "--f 100 --solver cg" "--f 64 --solver cg" "--f 200 --solver cg" "--shape hugewiki --scheme reduce --solver cg"; do
  python bench.py --steps 5 --warmup 1 $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg: ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), 'frac', round(r['frac'],3), r['kernel'])"
done > gpurun_out/sweep_call6.log 2>&1
cat gpurun_out/sweep_call6.log
export ROUND=r03
TAG=lu FULL=1 BENCH_ARGS="--solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=gram_only GRAM_ONLY=1 BENCH_ARGS="--f 100 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
cat gpurun_out/profiles_r03/lu/kernel_trace_by_side.txt gpurun_out/profiles_r03/lu/traffic.json
