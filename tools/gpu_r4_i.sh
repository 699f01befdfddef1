#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2"
for i in 1 2 3; do
for L in variants/libALS_w50.so variants/libALS_w51.so variants/libALS_w52.so; do
  CUMF_ALS_LIB=$R/$L python bench.py $Q --f 64 --solver lu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L f64', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
done
for L in variants/libALS_lu_serial.so variants/libALS_lu_pipe.so variants/libALS_lu_blocked.so; do
  CUMF_ALS_LIB=$R/$L python bench.py $Q --f 100 --solver lu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L f100', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
done
done > gpurun_out/i_ab.txt 2>&1
sort gpurun_out/i_ab.txt
