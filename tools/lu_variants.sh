#!/bin/bash
# Build libALS variants of the fused LU into variants/ (NB = 7 kernels only, one translation
# unit, ~40 s each) and time them: TAG:ABLATION pairs, TAG = free label (-DCUMF_VARIANT_TAG),
# ABLATION = CUMF_VARIANT_A bits (2: no MFMA update, 8: no back substitution, 32: no elimination,
# 64: no per-block row forming; results are then wrong).
# Usage: tools/lu_variants.sh build "0:0 0:2 0:8 0:32 0:40 0:64"
#        tools/lu_variants.sh run            (batched solve only, tools/bench_solve.py)
#        tools/bench_variants.sh             (bench.py, Netflix f=100, with every variant)
set -e
cd "$(dirname "$0")/.."
C=cumf_als_amd/csrc
if [ "$1" = build ]; then
  mkdir -p variants
  for v in $2; do
    M=${v%%:*}; A=${v##*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
      -DCUMF_VARIANT_TAG=$M -DCUMF_VARIANT_A=$A -Iinclude -I$C -DCUMF_ONLY_NB=${NBONLY:-7} -c $C/als_kernels.hip -o variants/k_${M}_${A}.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libALS_${M}_${A}.so variants/k_${M}_${A}.o \
      $C/als_plan.o $C/als_driver.o $C/host_utilities.o
    rm variants/k_${M}_${A}.o
  done
else
  for so in variants/libALS_*.so; do
    echo "== $so"
    CUMF_ALS_LIB=$so python tools/bench_solve.py ${2:-100} ${3:-60000} lu 2>&1 | grep " LU "
  done
fi
