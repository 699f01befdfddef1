#!/bin/bash
# Build libALS variants of the register-resident LU (panel height / ablations) into variants/
# and, with "run", time the batched solve with each (tools/bench_solve.py, LU line only).
# Usage: tools/lu_variants.sh build "1:0 2:0 4:0 2:1 2:2 2:4 2:8"   (PANEL:ABLATION pairs)
#        tools/lu_variants.sh run
set -e
cd "$(dirname "$0")/.."
C=cumf_als_amd/csrc
if [ "$1" = build ]; then
  mkdir -p variants
  for v in $2; do
    M=${v%%:*}; A=${v##*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
      -DCUMF_BACK_RING=$M -DCUMF_VARIANT_A=$A -Iinclude -I$C -DCUMF_ONLY_NB=${NBONLY:-7} -c $C/als_kernels.hip -o variants/k_${M}_${A}.o
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
