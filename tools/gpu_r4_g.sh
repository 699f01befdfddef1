#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lu or fused or doals or whole_row or sse" > gpurun_out/g_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/g_parity.log | cut -c1-300
CUMF_ALS_LIB=$PWD/cumf_als_amd/csrc/libALS_ablate.so python tools/lu_alone.py --extra 256 512 1024 2048 2>&1 | tail -1 > gpurun_out/g_lu_parts.txt; cat gpurun_out/g_lu_parts.txt
for i in 1 2; do
for L in cumf_als_amd/csrc/libALS.so variants/libALS_lu_serial.so; do
  CUMF_ALS_LIB=$R/$L python bench.py --no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2 --f 100 --solver lu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
done; done > gpurun_out/g_ab.txt 2>&1
cat gpurun_out/g_ab.txt
