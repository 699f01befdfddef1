#!/bin/bash
# round 4, call B: software-pipelined lu_wave vs the panel-serial one (variants/libALS_lu_serial.so): bits, time, parity
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python tools/lib_equal.py cumf_als_amd/csrc/libALS.so variants/libALS_lu_serial.so 100 lu 0.3 > gpurun_out/b_equal.txt 2>&1
python tools/lib_equal.py cumf_als_amd/csrc/libALS.so variants/libALS_lu_serial.so 110 lu 0.2 >> gpurun_out/b_equal.txt 2>&1
cat gpurun_out/b_equal.txt
tools/ab_libs.sh "--f 100 --solver lu" cumf_als_amd/csrc/libALS.so variants/libALS_lu_serial.so 3 > gpurun_out/b_ab.txt 2>&1
cat gpurun_out/b_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lu or fused or doals or whole_row" > gpurun_out/b_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/b_parity.log
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2"
for cfg in "--f 64 --solver lu" "--f 32 --solver lu" "--f 96 --solver lu"; do
  python bench.py $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg: ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
done > gpurun_out/b_other_f.txt 2>&1
cat gpurun_out/b_other_f.txt
