#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/j_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/j_parity.log | cut -c1-300
python -m cumf_als_amd.datagen --shape netflix /tmp/netflix_synth > /dev/null 2>&1
cd cumf_als_amd/csrc
for mode in fused kernel; do
  for solver in lu cg; do
    if [ $mode = kernel ]; then export CUMF_ALS_RMSE=kernel; else unset CUMF_ALS_RMSE; fi
    CUMF_ALS_SOLVER=$solver ./main 17770 480189 100 99072112 1408395 0.048 1 3 /tmp/netflix_synth > $R/gpurun_out/main_${solver}_${mode}.log 2>&1
    echo "$solver $mode: $(grep 'doALS takes' $R/gpurun_out/main_${solver}_${mode}.log) | $(grep 'Train RMSE in iter 9' $R/gpurun_out/main_${solver}_${mode}.log) | $(grep 'Test RMSE in iter 9' $R/gpurun_out/main_${solver}_${mode}.log)"
  done
done
