#!/bin/bash
# final evidence of round 6: the default bench line, rocprofv3 passes for every BASELINE config, the sweep over the
# configurations, ./main end to end, the N = 2 bench line and an N = 8 one at a small scale (gloo stand-in on one GPU).
# Everything lands under gpurun_out/r06final (copied to profiles/r06/ afterwards).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06final; mkdir -p $O $R/gpurun_out/r06
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
export ROUND=r06
TAG=lu FULL=1 BENCH_ARGS="--solver lu" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=cg BENCH_ARGS="--solver cg" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=gram_only GRAM_ONLY=1 BENCH_ARGS="--f 100 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f200_cg BENCH_ARGS="--f 200 --solver cg" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f200_lu BENCH_ARGS="--f 200 --solver lu" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f64_lu BENCH_ARGS="--f 64 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=hugewiki_cg BENCH_ARGS="--shape hugewiki --scheme reduce --solver cg" timeout 600 tools/collect_profiles.sh > /dev/null 2>&1
# LDS / instruction counters of the headline launches (their own pass)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_x
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/prof_x -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gram-leg --no-fast-leg --allow-missing-traffic > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/prof_x > $R/gpurun_out/profiles_r06/lu/pmc_lds.txt
cd $R
tools/gpu_round6_sweep.sh final > /dev/null 2>&1; cp $R/gpurun_out/r06/sweep_final.txt $O/final_sweep.txt; cat $O/final_sweep.txt
CUMF_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; echo "2-rank bench rc=$?"
CUMF_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --steps 2 --warmup 1 --scale 0.1 --no-cpu-baseline > $O/bench_8ranks_gloo.json 2> $O/bench_8ranks_gloo.err; echo "8-rank bench rc=$?"
python -m cumf_als_amd.datagen --shape netflix /tmp/netflix_synth > /dev/null 2>&1
cd cumf_als_amd/csrc
for solver in lu cg; do
  CUMF_ALS_TIMING=1 CUMF_ALS_SOLVER=$solver ./main 17770 480189 100 99072112 1408395 0.048 1 3 /tmp/netflix_synth > $O/main_netflix_${solver}.log 2>&1
  grep "doALS takes" $O/main_netflix_${solver}.log
done
