#!/bin/bash
# round 5, call B: is the LU bound by the CU's LDS pipe (ds_bpermute)?  Profiling build, switches 4096 (panel rows right of
# the diagonal tile without their broadcast) and 8192 (... and formed by one fp32 MFMA instead of 4 FMA + 1 mul)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/r05b; mkdir -p $O
export CUMF_ALS_LIB=$R/cumf_als_amd/csrc/libALS_ablate.so
python tools/lu_alone.py --reps 3 --extra 4096 8192 --raw 4096 8192 > $O/lu_lds_ablation.json 2> $O/err.txt
cat $O/lu_lds_ablation.json; tail -3 $O/err.txt
