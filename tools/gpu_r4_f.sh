#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for L in cumf_als_amd/csrc/libALS_ablate.so variants/libALS_ablate_pipe.so variants/libALS_ablate_serial.so; do
  CUMF_ALS_LIB=$R/$L python tools/lu_alone.py 2>&1 | tail -1
done > gpurun_out/f_lu_alone.txt
CUMF_ALS_LIB=$R/cumf_als_amd/csrc/libALS_ablate.so python tools/lu_alone.py --solver cg 2>&1 | tail -1 >> gpurun_out/f_lu_alone.txt
cat gpurun_out/f_lu_alone.txt
