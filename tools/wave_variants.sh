#!/bin/bash
# Build variants of the one-wave kernels (extra compiler flags / -D switches through EXTRA) for one feature-block count
# into variants/libALS_w<V>.so next to the default build; run on the GPU box with
#   CUMF_ALS_LIB=variants/libALS_w<V>.so python bench.py ...      (or tools/ab_libs.sh)
# usage: [NB=7] tools/wave_variants.sh 1 2 4 ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/cumf_als_amd/csrc
NB=${NB:-7}
mkdir -p $R/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I$R/include -I$C"
for V in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -DCUMF_WAVE_PART=0 -DCUMF_WAVE_NB=$NB ${EXTRA:-} -c $C/als_wave.hip -o $R/variants/als_wave_w${NB}_v$V.o &
  /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -DCUMF_WAVE_PART=1 -DCUMF_WAVE_NB=$NB ${EXTRA:-} -c $C/als_wave.hip -o $R/variants/als_wave_l${NB}_v$V.o &
  wait
  OBJS=$(ls $C/*.o | grep -v "_ablate.o" | grep -v "als_wave_w${NB}.o" | grep -v "als_wave_l${NB}.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libALS_w$V.so $OBJS $R/variants/als_wave_w${NB}_v$V.o $R/variants/als_wave_l${NB}_v$V.o
  echo built variants/libALS_w$V.so
done
