#!/bin/bash
# per-phase DYNAMIC instruction table of the in-kernel LU of the Theta side (lu_wave_blocked +
# back_substitute_tiles, 480 189 systems of 100 x 100), from rocprofv3 --pmc passes of the profiling build with its solve-only
# switches: 2 = the solve alone, +256 no panel preparation, +512 no fp32 MFMAs, +1024 no trailing update, +2048 no back
# substitution.  One rocprofv3 run per variant (counters only: no tracing beside --pmc); tools/lu_phase_table.py turns the
# files into the table.  Output: gpurun_out/r06/lu_phase/<variant>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/lu_phase; mkdir -p $O
export CUMF_ALS_LIB=$R/cumf_als_amd/csrc/libALS_ablate.so
cd /tmp && export TMPDIR=/tmp
for v in 0 256 512 1024 2048 3840; do
  if [ $v = 0 ]; then ARGS="--only solve_only"; NAME=solve_only; else ARGS="--extra $v --only solve_only+$v"; NAME=solve_only+$v; fi
  rm -rf /tmp/prof_lu
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_lu -o p -- python $R/tools/lu_alone.py --reps 2 $ARGS > $O/$NAME.json 2>/dev/null
  python $R/tools/pmc_summary.py /tmp/prof_lu > $O/$NAME.txt
done
ls $O
