#!/bin/bash
# round 5, call A: (1) parity of the diagonal-tile products, (2) A/B of the stage variants on one box,
# (3) counter passes of the solve-only build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=$R/gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --durations=8 > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log; tail -4 $O/parity.log
L="cumf_als_amd/csrc/libALS.so variants/libALS_r04.so variants/libALS_we1.so variants/libALS_we2.so variants/libALS_wd0e1.so variants/libALS_wd0e2.so"
CUMF_ALS_LIB=$R/variants/libALS_r04.so python tools/time_halves.py --save /tmp/ref_lu.pt > $O/ab.txt 2>$O/ab.err
for rep in 1 2; do for l in $L; do CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --check /tmp/ref_lu.pt >> $O/ab.txt 2>>$O/ab.err; done; done
for sp in 1 2; do CUMF_ALS_SPLIT_LAUNCH=$sp timeout 300 python tools/time_halves.py --check /tmp/ref_lu.pt >> $O/ab.txt 2>>$O/ab.err; done
for l in cumf_als_amd/csrc/libALS.so variants/libALS_r04.so variants/libALS_we2.so; do CUMF_ALS_LIB=$R/$l timeout 300 python tools/time_halves.py --solver cg >> $O/ab.txt 2>>$O/ab.err; done
cat $O/ab.txt
# counters of the solve alone (profiling build, switch 2)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters_available.txt
export CUMF_ALS_LIB=$R/cumf_als_amd/csrc/libALS_ablate.so
B="python $R/tools/lu_alone.py --only solve_only --reps 2"
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p1 -o p -- $B > $O/solve_only_1.json 2> $O/p1.err
python $R/tools/pmc_summary.py /tmp/p1 > $O/pmc_solve_only_insts.txt
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/p2 -o p -- $B > $O/solve_only_2.json 2> $O/p2.err
python $R/tools/pmc_summary.py /tmp/p2 > $O/pmc_solve_only_active.txt
B="python $R/tools/lu_alone.py --only gram_only --reps 2"
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/p3 -o p -- $B > $O/gram_only_2.json 2> $O/p3.err
python $R/tools/pmc_summary.py /tmp/p3 > $O/pmc_gram_only_active.txt
tail -3 $O/p1.err $O/p2.err; cat $O/pmc_solve_only_insts.txt $O/pmc_solve_only_active.txt $O/pmc_gram_only_active.txt
