#!/bin/bash
# Build variants of the wave kernels (CUMF_WAVE_VARIANT bit switches in als_wave.hip) into
# variants/libALS_w<V>.so next to the default build; run on the GPU box with
#   CUMF_ALS_LIB=variants/libALS_w<V>.so python bench.py ...
# usage: tools/wave_variants.sh 1 2 3 ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/cumf_als_amd/csrc
mkdir -p $R/variants
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I$R/include -I$C \
    -fno-slp-vectorize -DCUMF_WAVE_NB=7 -DCUMF_WAVE_VARIANT=$V ${EXTRA:-} -c $C/als_wave.hip -o $R/variants/als_wave_w7_v$V.o
  OBJS=$(ls $C/*.o | grep -v als_wave_w7.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libALS_w$V.so $OBJS $R/variants/als_wave_w7_v$V.o
  echo built variants/libALS_w$V.so
done
