#!/bin/bash
# second evidence run of round 6 (after the side rows of the one-wave LU): GPU test suite, the default bench line, rocprofv3
# passes of the configurations whose kernels changed (f = 100 LU, f = 64 LU), the LDS / instruction counter pass, the sweep,
# the N = 2 / N = 8 gloo lines, ./main end to end.  Lands under gpurun_out/r06final2 (copied to profiles/r06/ afterwards).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06final3; mkdir -p $O $R/gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
export ROUND=r06
TAG=lu FULL=1 BENCH_ARGS="--solver lu" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f64_lu BENCH_ARGS="--f 64 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_x
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/prof_x -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gram-leg --no-fast-leg --allow-missing-traffic > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/prof_x > $R/gpurun_out/profiles_r06/lu/pmc_lds.txt
cd $R
tools/gpu_round6_sweep.sh final3 > /dev/null 2>&1; cp $R/gpurun_out/r06/sweep_final3.txt $O/final_sweep.txt; cat $O/final_sweep.txt
CUMF_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; echo "2-rank bench rc=$?"
CUMF_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --steps 2 --warmup 1 --scale 0.1 --no-cpu-baseline > $O/bench_8ranks_gloo.json 2> $O/bench_8ranks_gloo.err; echo "8-rank bench rc=$?"
python -m cumf_als_amd.datagen --shape netflix /tmp/netflix_synth > /dev/null 2>&1
cd cumf_als_amd/csrc
for solver in lu cg; do
  CUMF_ALS_TIMING=1 CUMF_ALS_SOLVER=$solver ./main 17770 480189 100 99072112 1408395 0.048 1 3 /tmp/netflix_synth > $O/main_netflix_${solver}.log 2>&1
  grep "doALS takes" $O/main_netflix_${solver}.log
done
