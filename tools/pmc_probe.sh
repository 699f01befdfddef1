#!/bin/bash
# exploratory PMC passes on the GPU box: tools/pmc_probe.sh <tag> "<counters pass 1>" ["<pass 2>" ...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
i=0
for C in "$@"; do
  rm -rf /tmp/prof_p
  rocprofv3 --pmc $C --output-format csv -d /tmp/prof_p -o p -- $BENCH > /dev/null 2> $OUT/pass$i.err
  python $R/tools/pmc_summary.py /tmp/prof_p > $OUT/pass$i.txt
  cat $OUT/pass$i.txt
  i=$((i+1))
done
