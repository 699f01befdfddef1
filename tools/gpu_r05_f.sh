#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/r05f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "large_f or fused_half or chunked or exact or lu" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
for rep in 1 2; do for l in cumf_als_amd/csrc/libALS.so variants/libALS_r04.so; do
  for f in 200 160 128; do CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --f $f --solver lu --reps 3 >> $O/lu_b128.txt 2>>$O/err.txt; done
done; done
cat $O/lu_b128.txt | cut -c1-200; tail -3 $O/err.txt
