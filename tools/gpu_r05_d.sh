#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -k "headline or sampled_rows or hugewiki_slab" --durations=10 > $O/fullsize.log 2>&1; echo "fullsize rc=$?"
grep -E "headline doALS|CG\(|passed|failed|Error|assert" $O/fullsize.log | cut -c1-2500 | tail -40
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_dist_gpu.py -q -k "near_perfect or fused_half or quadratic or bench_world2 or train_sse" --durations=5 > $O/parity.log 2>&1; echo "parity rc=$?"; tail -8 $O/parity.log
