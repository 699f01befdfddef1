#!/bin/bash
# round 4, first GPU call: the new parity tests (headline RMSE log, tightened LU rows), the N > 1 bench line, baseline bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rA -s -k "headline or (sampled_rows and 100)" > gpurun_out/a_headline.log 2>&1; echo "headline rc=$?"
grep -E "headline doALS|per-row relative|passed|failed|FAILED|Error" gpurun_out/a_headline.log | cut -c1-600
timeout 600 python -m pytest tests/test_dist_gpu.py -m gpu -q -rA -k "bench_world2" > gpurun_out/a_dist.log 2>&1; echo "dist rc=$?"
tail -5 gpurun_out/a_dist.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"
python - <<'E'
import json
d=json.loads(open('gpurun_out/a_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('ms', d['ms_per_step'], 'x', r['x_side_ms'], 'theta', r['theta_side_ms'], 'frac', r['frac'])
print(d.get('parity_at_scale'))
E
