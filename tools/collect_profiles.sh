#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for ONE bench.py configuration.
#   1. --kernel-trace --stats  : per-kernel time (csv)             -> kernel_stats.csv, kernel_trace_by_side.txt
#   2. --pmc pass A (own run)  : MFMA busy / issue / wave cycles   -> pmc_sq.txt
#   3. --pmc pass B (own run)  : FETCH_SIZE (TCC, 3 slots)         -> pmc_fetch.txt
#   4. --pmc pass C (own run)  : WRITE_SIZE                        -> pmc_write.txt
#   5. --pmc pass D (own run)  : TCC hit / miss  (FULL=1 only)     -> pmc_tcc.txt
# Counters are collected in their own runs (no sys/runtime/hip tracing with --pmc).  Every pmc_*.txt
# lists the X-side and the Theta-side launches of a kernel separately (they differ in grid size).
# Usage: ROUND=r03 TAG=lu BENCH_ARGS="--solver lu" tools/collect_profiles.sh
#        ROUND=r03 TAG=gram_only GRAM_ONLY=1 BENCH_ARGS="--f 100 --solver lu" tools/collect_profiles.sh
#          (the Gram pass alone: tools/gram_pass_alone.py on the profiling build libALS_ablate.so; its own directory,
#           so that the per-kernel averages of the production launches stay clean)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_${ROUND:-r03}/${TAG:-lu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${GRAM_ONLY:-0}" = "1" ]; then
  export CUMF_ALS_LIB=$R/cumf_als_amd/csrc/libALS_ablate.so
  BENCH="python $R/tools/gram_pass_alone.py ${BENCH_ARGS:-}"
else
  BENCH="python $R/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-gram-leg --no-fast-leg --allow-missing-traffic ${BENCH_ARGS:-}"
fi
echo "$BENCH" > $OUT/command.txt
rm -rf /tmp/prof_kt /tmp/prof_sq /tmp/prof_f /tmp/prof_w /tmp/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $BENCH > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
python - <<PY > $OUT/kernel_stats.csv
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_kt/**/*kernel_stats.csv", recursive=True)[0])))
keep = [r for r in rows if "cumf" in r["Name"]] + [r for r in rows if "cumf" not in r["Name"]][:3]
w = csv.DictWriter(__import__("sys").stdout, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
python - <<PY > $OUT/kernel_trace_by_side.txt
# average duration per (kernel, grid size): the X-side and Theta-side launches separately
import csv, glob, collections
f = glob.glob("/tmp/prof_kt/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "cumf" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(agg.items()):
    print(f"{k:62s} grid={g:>10s} n={len(v):3d} avg_ms={sum(v)/len(v)/1e6:.4f} min_ms={min(v)/1e6:.4f} max_ms={max(v)/1e6:.4f}")
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_sq -o p -- $BENCH > /dev/null 2> $OUT/sq.err
python $R/tools/pmc_summary.py /tmp/prof_sq > $OUT/pmc_sq.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
python $R/tools/pmc_summary.py /tmp/prof_f > $OUT/pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o p -- $BENCH > /dev/null 2> $OUT/write.err
python $R/tools/pmc_summary.py /tmp/prof_w > $OUT/pmc_write.txt
if [ "${FULL:-0}" = "1" ]; then
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/prof_t -o p -- $BENCH > /dev/null 2> $OUT/tcc.err
  python $R/tools/pmc_summary.py /tmp/prof_t > $OUT/pmc_tcc.txt
fi
python $R/tools/make_traffic_json.py $OUT > $OUT/traffic.json
rm -f $OUT/*.err
ls $OUT
