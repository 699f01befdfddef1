// Does a SIMD overlap one wave's VALU with another wave's (or its own) MFMAs?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o gpurun_out/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  bf16x8 ab;
  for (int i = 0; i < 8; ++i) ab[i] = (short)(threadIdx.x + i);
  float c1 = 1.0f + threadIdx.x * 1e-6f, c2 = threadIdx.x * 1e-3f;  // VALU constants in registers of their own
  asm volatile("" : "+v"(c1), "+v"(c2));
  for (int it = 0; it < iters; ++it) {
    // INTERLEAVE: NV / NM VALU ops behind every MFMA (the shadow the matrix pipe leaves)
#pragma unroll
    for (int m = 0; m < (NM ? NM : 1); ++m) {
      if constexpr (NM > 0) {
        if constexpr (KIND == 0)
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[m & 7]) : "v"(ab));
      }
#pragma unroll
      for (int k = 0; k < NV / (NM ? NM : 1); ++k)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(c1), "v"(c2));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NV, int KIND>
static void run(const char* name, int waves_per_simd, float* d, int grid = 256) {
  const int iters = 4000;
  const int block = 64 * 4 * waves_per_simd;  // one workgroup per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NM, NV, KIND><<<grid, block>>>(d, 10);
  hipEventRecord(e0);
  probe<NM, NV, KIND><<<grid, block>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // cycles per loop iteration per SIMD at 2.4 GHz
  printf("grid %3d %-22s waves/SIMD %d  NM %2d NV %3d  %.3f ms  %.0f cycles/iter/SIMD (per wave-iter %.0f)\n", grid, name, waves_per_simd, NM,
         NV, ms, ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / waves_per_simd);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
  for (int grid : {1, 256})
  for (int w = 1; w <= 2; ++w) {
    run<8, 0, 0>("f32 mfma only", w, d, grid);
    run<0, 64, 0>("valu only", w, d, grid);
    run<8, 64, 0>("f32 mfma + valu", w, d, grid);
    run<8, 32, 0>("f32 mfma + valu", w, d, grid);
    run<8, 16, 0>("f32 mfma + valu", w, d, grid);
    run<8, 0, 1>("bf16 mfma only", w, d, grid);
    run<8, 32, 1>("bf16 mfma + valu", w, d, grid);
    run<8, 64, 1>("bf16 mfma + valu", w, d, grid);
  }
  return 0;
}
