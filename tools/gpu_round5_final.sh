#!/bin/bash
# final evidence of round 5: the default bench line, rocprofv3 passes for every BASELINE config, the sweep over the
# configurations, ./main end to end, the N = 2 bench line (gloo stand-in on one GPU) with its hugewiki leg.
# Everything lands under gpurun_out/r05final (copied to profiles/r05/ afterwards).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05final; mkdir -p $O
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
export ROUND=r05
TAG=lu FULL=1 BENCH_ARGS="--solver lu" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=cg BENCH_ARGS="--solver cg" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=gram_only GRAM_ONLY=1 BENCH_ARGS="--f 100 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f200_cg BENCH_ARGS="--f 200 --solver cg" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f200_lu BENCH_ARGS="--f 200 --solver lu" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f64_lu BENCH_ARGS="--f 64 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=hugewiki_cg BENCH_ARGS="--shape hugewiki --scheme reduce --solver cg" timeout 600 tools/collect_profiles.sh > /dev/null 2>&1
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic"
for cfg in "--f 100 --solver lu" "--f 100 --solver cg" "--f 64 --solver lu" "--f 64 --solver cg" "--f 128 --solver cg" "--f 128 --solver lu" "--f 200 --solver cg" "--f 200 --solver lu" "--shape hugewiki --scheme reduce --solver cg" "--shape hugewiki --scheme reduce --solver lu" "--shape hugewiki --scheme reduce --solver cg --reference-solvers" "--shape ml10m --f 10 --solver lu" "--shape ml10m --f 10 --solver cg"; do
  python bench.py --steps 5 --warmup 1 $Q $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg: ms', round(d['ms_per_step'],3), 'G ratings/s', round(d['value']/1e9,3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3), 'dominant', r['dominant'], 'frac', round(r['frac'],3), 'step mean frac', round(r['step_mean']['frac'],3), r['kernel'])"
done > $O/final_sweep.txt 2>&1
cat $O/final_sweep.txt
CUMF_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; echo "2-rank bench rc=$?"
python -m cumf_als_amd.datagen --shape netflix /tmp/netflix_synth > /dev/null 2>&1
cd cumf_als_amd/csrc
for solver in lu cg; do
  CUMF_ALS_TIMING=1 CUMF_ALS_SOLVER=$solver ./main 17770 480189 100 99072112 1408395 0.048 1 3 /tmp/netflix_synth > $O/main_netflix_${solver}.log 2>&1
  grep "doALS takes" $O/main_netflix_${solver}.log
done
