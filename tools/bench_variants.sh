#!/bin/bash
# bench.py (Netflix f=100) with every library under variants/ (tools/lu_variants.sh build ...)
cd "$(dirname "$0")/.."
for so in variants/libALS_*.so; do
  CUMF_ALS_LIB=$so python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$so  step=%.2f x_ms=%.2f theta_ms=%.2f'%(j['ms_per_step'], r['x_side_ms'],r['theta_side_ms']))"
done
