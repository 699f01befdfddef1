#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for bench.py's default command.
#   1. --kernel-trace --stats  : per-kernel time (csv)             -> kernel_stats.csv
#   2. --pmc pass A (own run)  : MFMA busy / issue / wave cycles   -> pmc_sq.txt
#   3. --pmc pass B (own run)  : FETCH_SIZE (TCC, 3 slots)         -> pmc_fetch.txt
#   4. --pmc pass C (own run)  : WRITE_SIZE                        -> pmc_write.txt
# Counters are collected in their own runs (no sys/runtime/hip tracing with --pmc).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_${ROUND:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $BENCH > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
python - <<PY > $OUT/kernel_stats.csv
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_kt/**/*kernel_stats.csv", recursive=True)[0])))
keep = [r for r in rows if "cumf" in r["Name"]] + [r for r in rows if "cumf" not in r["Name"]][:3]
w = csv.DictWriter(__import__("sys").stdout, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq -o p -- $BENCH > /dev/null 2> $OUT/sq.err
python $R/tools/pmc_summary.py /tmp/prof_sq > $OUT/pmc_sq.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
python $R/tools/pmc_summary.py /tmp/prof_f > $OUT/pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o p -- $BENCH > /dev/null 2> $OUT/write.err
python $R/tools/pmc_summary.py /tmp/prof_w > $OUT/pmc_write.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/prof_t -o p -- $BENCH > /dev/null 2> $OUT/tcc.err
python $R/tools/pmc_summary.py /tmp/prof_t > $OUT/pmc_tcc.txt
rm -f $OUT/*.err
ls -la $OUT
