#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/r05g; mkdir -p $O; rm -f $O/*
python tools/check_large_lu.py 144 150 160 176 190 192 196 200 206 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_f or fused_half or chunked or sse or lu" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
for rep in 1 2; do
  for f in 200 160 144; do
    timeout 300 python tools/time_halves.py --f $f --solver lu --reps 3 >> $O/lu_rows.txt 2>>$O/err.txt
    CUMF_ALS_WG_LU=1 timeout 300 python tools/time_halves.py --f $f --solver lu --reps 3 >> $O/lu_rows.txt 2>>$O/err.txt
  done
done
cat $O/lu_rows.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "sampled_rows and (200-lu or 160-lu)" > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; grep -E "passed|failed|^E  |per-row" $O/fullsize.log | cut -c1-330 | head
