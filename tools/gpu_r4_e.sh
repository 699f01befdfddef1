#!/bin/bash
# round 4, call E: blocked LU (bf16x3 trailing updates) vs pipelined vs serial: parity tests, time
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lu or fused or doals or whole_row or sse" > gpurun_out/e_parity.log 2>&1; echo "parity rc=$?"; tail -15 gpurun_out/e_parity.log | cut -c1-300
for i in 1 2 3; do
for L in cumf_als_amd/csrc/libALS.so variants/libALS_lu_pipe.so variants/libALS_lu_serial.so; do
  CUMF_ALS_LIB=$R/$L python bench.py --no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2 --f 100 --solver lu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), 'rmse', d['rmse'])"
done; done > gpurun_out/e_ab.txt 2>&1
cat gpurun_out/e_ab.txt
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2"
for cfg in "--f 64 --solver lu" "--f 32 --solver lu" "--f 96 --solver lu"; do
  python bench.py $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg: ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
done > gpurun_out/e_other_f.txt 2>&1
cat gpurun_out/e_other_f.txt
