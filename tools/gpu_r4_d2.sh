#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rA -s -k "fused_train_sse or rmse_log_fused" > gpurun_out/d_sse.log 2>&1; echo "sse tests rc=$?"
grep -E "fused train SSE|doALS (lu|cg)|passed|failed|Error|assert" gpurun_out/d_sse.log | cut -c1-300 | head -60
