// mfma_k16_probe.hip -- does the K = 16 bf16 MFMA (v_mfma_f32_16x16x16_bf16) cost half of the K = 32 one
// (v_mfma_f32_16x16x32_bf16) on gfx950?  If so, a tail stage with <= 16 ratings could run at half the matrix-pipe
// time.  One wave, 8 independent accumulators, REP back-to-back rounds, s_memtime around them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, long long* cycles, int rep) {
  f32x4 acc[8];
  for (auto& a : acc) a = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x;
  s16x4 a4, b4;
  bf16x8 a8, b8;
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + lane + i); b4[i] = (short)(0x3f00 + lane + 2 * i); }
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(1.0f + 0.01f * (lane + i)); b8[i] = (__bf16)(0.5f + 0.01f * (lane + 2 * i)); }
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // inline asm with VGPR accumulators: the builtins make the compiler shuffle AGPR copies and s_nops into the loop
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a8), "v"(b8));
      if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a4), "v"(b4));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (auto& a : acc) s += a[0] + a[1] + a[2] + a[3];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc((void**)&out, 4096 * 64 * 4);
  hipMalloc((void**)&cyc, 4096 * 8);
  const int rep = 2000;
  for (int blocks : {1, 1024, 2048, 4096, 8192}) {
    for (int kind = 0; kind < 2; ++kind) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        if (kind == 0) k<0><<<blocks, 64>>>(out, cyc, rep);
        else k<1><<<blocks, 64>>>(out, cyc, rep);
        hipEventRecord(e1);
        hipDeviceSynchronize();
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long c;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s blocks %4d: %.1f s_memtime ticks per MFMA (wave 0), kernel %.3f ms = %.2f ns per MFMA per wave\n",
             kind == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x16_bf16", blocks, (double)c / (8.0 * rep), ms,
             ms * 1e6 / (8.0 * rep));
    }
  }
  return 0;
}
