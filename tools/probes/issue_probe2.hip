// Which VALU instructions hide behind v_mfma_f32_16x16x32_bf16 on gfx950, and which add to it?
// Per loop iteration: 8 MFMAs (independent accumulators), each followed by NV / 8 instructions of one kind.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe2.hip -o gpurun_out/issue_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum { kFma, kAdd, kMul, kAnd, kCvtPk, kCvtUbyte, kLdexp, kDot2c, kMov, kPerm, kLshl, kCndmask, kNone };
static const char* kNames[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_and_b32", "v_cvt_pk_bf16_f32", "v_cvt_f32_ubyte1",
                               "v_ldexp_f32", "v_dot2c_f32_bf16", "v_mov_b32", "v_perm_b32", "v_lshlrev_b32", "v_cndmask_b32", "(none)"};

template <int OP>
__device__ __forceinline__ void valu(float& v, float c1, float c2) {
  if constexpr (OP == kFma) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2));
  if constexpr (OP == kAdd) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(c1));
  if constexpr (OP == kMul) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(c1));
  if constexpr (OP == kAnd) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v) : "v"(c1));
  if constexpr (OP == kCvtPk) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(c1));
  if constexpr (OP == kCvtUbyte) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v));
  if constexpr (OP == kLdexp) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v) : "v"(c2));
  if constexpr (OP == kDot2c) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2));
  if constexpr (OP == kMov) asm volatile("v_mov_b32 %0, %1" : "+v"(v) : "v"(c1));
  if constexpr (OP == kPerm) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2));
  if constexpr (OP == kLshl) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v));
  if constexpr (OP == kCndmask) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(c1) : );
}

template <int NM, int NV, int OP>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  bf16x8 pa[4], pb[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      pa[q][i] = (short)(threadIdx.x + i + q);
      pb[q][i] = (short)(threadIdx.x * 3 + i + q);
    }
  for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(pa[q]), "+v"(pb[q]));
  float c1 = 1.0f + threadIdx.x * 1e-6f, c2 = threadIdx.x * 1e-3f;
  asm volatile("" : "+v"(c1), "+v"(c2));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < (NM ? NM : 8); ++m) {
      // distinct A and B operands, rotating over four register quads each (as a tile row / column walk does)
      if constexpr (NM > 0)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(pa[m & 3]), "v"(pb[(m + 1) & 3]));
#pragma unroll
      for (int k = 0; k < NV / 8; ++k) valu<OP>(v[(m * (NV / 8) + k) & 7], c1, c2);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NV, int OP>
static float run(int waves_per_simd, float* d) {
  const int iters = 4000, grid = 256, block = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NM, NV, OP><<<grid, block>>>(d, 10);
  hipEventRecord(e0);
  probe<NM, NV, OP><<<grid, block>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.1e9f / iters;  // cycles per loop iteration per SIMD at ~2.1 GHz (relative numbers matter)
}

template <int OP>
static void row(float* d) {
  for (int w = 1; w <= 2; ++w) {
    const float m = run<8, 0, kNone>(w, d);
    const float v16 = run<0, 16, OP>(w, d), v32 = run<0, 32, OP>(w, d);
    const float b16 = run<8, 16, OP>(w, d), b32 = run<8, 32, OP>(w, d);
    printf("%-20s waves/SIMD %d | 8 mfma %5.0f | 16 valu alone %5.0f, with mfma %5.0f (+%4.0f) | 32 valu alone %5.0f, with mfma %5.0f (+%4.0f)\n",
           kNames[OP], w, m, v16, b16, b16 - m, v32, b32, b32 - m);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
  row<kFma>(d); row<kAdd>(d); row<kMul>(d); row<kAnd>(d); row<kCvtPk>(d); row<kCvtUbyte>(d); row<kLdexp>(d);
  row<kDot2c>(d); row<kMov>(d); row<kPerm>(d); row<kLshl>(d); row<kCndmask>(d);
  return 0;
}
