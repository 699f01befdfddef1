#!/bin/bash
# first GPU call of round 3: full -m gpu suite (new parity tests), the default bench line, profiles of configs 2-4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -rA --durations=15 > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
export ROUND=r03
TAG=lu FULL=1 BENCH_ARGS="--solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=gram_only GRAM_ONLY=1 BENCH_ARGS="--f 100 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f200_cg BENCH_ARGS="--f 200 --solver cg" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f200_lu BENCH_ARGS="--f 200 --solver lu" timeout 500 tools/collect_profiles.sh > /dev/null 2>&1
TAG=f64_lu BENCH_ARGS="--f 64 --solver lu" timeout 400 tools/collect_profiles.sh > /dev/null 2>&1
TAG=hugewiki_cg BENCH_ARGS="--shape hugewiki --scheme reduce --solver cg" timeout 600 tools/collect_profiles.sh > /dev/null 2>&1
ls gpurun_out/profiles_r03/*
