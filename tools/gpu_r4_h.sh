#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rA -s -k "(headline and lu) or (sampled_rows and lu)" > gpurun_out/h_full.log 2>&1; echo "fullsize rc=$?"
grep -E "headline doALS|per-row relative|passed|failed|FAILED|Error" gpurun_out/h_full.log | cut -c1-500
for i in 1 2 3 4; do
for L in cumf_als_amd/csrc/libALS.so variants/libALS_lu_serial.so; do
  CUMF_ALS_LIB=$R/$L python bench.py --no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2 --f 100 --solver lu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
done; done > gpurun_out/h_ab.txt 2>&1
cat gpurun_out/h_ab.txt
