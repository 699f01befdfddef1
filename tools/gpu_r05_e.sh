#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/r05e; mkdir -p $O
for rep in 1 2; do for l in cumf_als_amd/csrc/libALS.so variants/libALS_inplace13.so variants/libALS_r04.so; do
  CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --f 200 --solver lu --reps 3 >> $O/f200_lu.txt 2>>$O/err.txt
done; done
for l in cumf_als_amd/csrc/libALS.so variants/libALS_r04.so; do
  CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --f 200 --solver cg --reps 3 >> $O/f200_lu.txt 2>>$O/err.txt
  CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --f 128 --solver lu --reps 3 >> $O/f200_lu.txt 2>>$O/err.txt
  CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --f 64 --solver lu --reps 3 >> $O/f200_lu.txt 2>>$O/err.txt
done
cat $O/f200_lu.txt; tail -3 $O/err.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "headline" > $O/headline.log 2>&1; echo "headline rc=$?"; grep -E "passed|failed|^E  " $O/headline.log | head
