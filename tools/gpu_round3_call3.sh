#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python tools/calib_adversarial.py > gpurun_out/calib_adversarial.log 2>&1
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic"
for i in 1 2; do
  python bench.py --steps 10 --warmup 2 $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('default lib : ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), r['theta_side']['kernel'])"
  CUMF_ALS_LIB=$R/variants/libALS_l7ms.so python bench.py --steps 10 --warmup 2 $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('l7 misched  : ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), r['theta_side']['kernel'])"
done > gpurun_out/ab_l7.log 2>&1
cat gpurun_out/ab_l7.log
for cfg in "--f 100 --solver cg" "--f 64 --solver cg" "--f 64 --solver lu" "--f 200 --solver cg" "--f 200 --solver lu" "--shape hugewiki --scheme reduce --solver cg" "--shape hugewiki --scheme reduce --solver lu"; do
  python bench.py --steps 5 --warmup 1 $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg: ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), 'frac', round(r['frac'],3), r['kernel'])"
done > gpurun_out/sweep_call3.log 2>&1
cat gpurun_out/sweep_call3.log
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=10 > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log; grep "^FAILED" gpurun_out/gpu_tests.log
