// mfma_peak.hip -- calibrates the fp32 matrix-core roof used in DESIGN.md / bench.py:
// sustained v_mfma_f32_16x16x4_f32 rate with 7 independent accumulators per wave (the
// Gram kernel's shape), (a) register operands only, (b) operands re-read from LDS with the
// same read-ahead pattern as als_item_kernel.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LDS_OPERANDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[36 * 112];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 36 * 112; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[7];
  for (auto& a : acc) a = f32x4{0, 0, 0, 0};
  const float* rowp = lds + (lane >> 4) * 112 + (lane & 15);
  float b0 = rowp[0], b1 = rowp[16], b2 = rowp[32], b3 = rowp[48];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float n0 = b0, n1 = b1, n2 = b2, n3 = b3;
      if (LDS_OPERANDS) {
        const float* rp = rowp + ((g + 1) & 7) * 4 * 112;
        n0 = rp[0]; n1 = rp[16]; n2 = rp[32]; n3 = rp[48];
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, b3, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, b2, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, b3, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(b3, b3, acc[6], 0, 0, 0);
      if (LDS_OPERANDS) __builtin_amdgcn_sched_barrier(0);
      b0 = n0; b1 = n1; b2 = n2; b3 = n3;
    }
  }
  float s = 0;
  for (auto& a : acc) s += a[0] + a[1] + a[2] + a[3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d;
  hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int wgs_per_cu : {1, 2, 3, 4}) {
    for (int mode = 0; mode < 2; ++mode) {
      const int grid = 256 * wgs_per_cu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode) k<1><<<grid, 256>>>(d, iters); else k<0><<<grid, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)grid * 4 * iters * 8 * 7 * 2048.0;
      printf("wgs/cu=%d lds_operands=%d  %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, mode, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
