#!/bin/bash
# ablation of the item kernel (results are numerically wrong by construction): DBG bits
# 1=no gathers 2=no colidx loads 4=no MFMA 8=no LDS stage writes 16=no barriers
for S in ${SOLVERS:-cg lu}; do for D in ${DBGS:-0 31}; do
  CUMF_ALS_DBG=$D python bench.py --steps 2 --warmup 1 --solver $S --cg-iters ${CGI:-6} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$S dbg=$D x_ms=%.2f theta_ms=%.2f reduce=%.2f'%(r['x_side_ms'],r['theta_side_ms'],r['reduce_kernel_ms_x_side']))"
done; done
