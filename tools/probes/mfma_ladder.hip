// mfma_ladder.hip -- adds the ingredients of als_item_kernel's stage loop one at a time to
// the bare fp32-MFMA loop to see which one costs matrix-pipe utilisation.
//   F & 1: LDS operand reads (read-ahead by one group)   F & 2: one barrier per stage
//   F & 4: 4 x ds_write_b128 per thread per stage         F & 8: 4 x 16-byte row gathers
//   F & 16: gathered data is what gets stored (one stage later), like the real kernel
//   F & 32: column indices loaded one stage ahead (else the same index every stage)
//   F & 64: sched_group_barrier interleave of the slice with the MFMAs
//   F & 128: 'strip' variant: 5 MFMAs per group + 1 shared tile every 4th group + 10 DPP-broadcast
//            VALU FMAs per group (models moving the 5-column tail block off the matrix pipe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ table, const int* __restrict__ idx,
                                         int stages) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int LD = 112, SF = 32 * LD;
  for (int i = tid; i < 2 * SF + 8 * LD; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[7];
  for (auto& a : acc) a = f32x4{0, 0, 0, 0};
  f32x4 v[4];
  for (auto& x : v) x = f32x4{0, 0, 0, 0};
  float strip[10];
  for (auto& x : strip) x = 0.f;
  const int pc = tid & 31, rsub = tid >> 5;
  const int* myidx = idx + (size_t)blockIdx.x * stages * 32;
  int cols[4], cols_nx[4];
  for (int p = 0; p < 4; ++p) cols[p] = cols_nx[p] = myidx[rsub + 8 * p];
  float* dummy = lds + 2 * SF + 4 * LD + (tid & 15) * 4;
  for (int s = 0; s < stages; ++s) {
    const float* rowp = lds + (s & 1) * SF + (lane >> 4) * LD + (lane & 15);
    float* nxt = lds + ((s + 1) & 1) * SF;
    float b0 = rowp[0], b1 = rowp[16], b2 = rowp[32], b3 = rowp[48];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float n0 = b0, n1 = b1, n2 = b2, n3 = b3;
      if (F & 1) {
        const float* rp = rowp + (g + 1) * 4 * LD;
        n0 = rp[0]; n1 = rp[16]; n2 = rp[32]; n3 = rp[48];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, b3, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, b2, acc[4], 0, 0, 0);
      if (!(F & 128)) {
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, b3, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(b3, b3, acc[6], 0, 0, 0);
      } else {
        if ((g & 3) == (tid >> 6)) acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, b3, acc[5], 0, 0, 0);
#define STRIP_A(a)                                                                                              \
  {                                                                                                             \
    const float ba = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b3), 0x150 + a, 0xf, 0xf, false)); \
    strip[a] = fmaf(b0, ba, strip[a]);                                                                          \
    strip[5 + a] = fmaf(b1, ba, strip[5 + a]);                                                                  \
  }
        STRIP_A(0) STRIP_A(1) STRIP_A(2) STRIP_A(3) STRIP_A(4)
      }
      if (!(F & 64)) __builtin_amdgcn_sched_barrier(0);
      if ((g & 1) == 0) {
        const int p = g >> 1;
        if (F & 4) {
          float* dst = (pc < 28) ? nxt + (rsub + 8 * p) * LD + 4 * pc : dummy;
          *reinterpret_cast<f32x4*>(dst) = v[p];
        }
        if (F & 8) {
          const unsigned off = (unsigned)cols[p] * 100u + (pc < 25 ? 4u * pc : 0u);
          f32x4 x = *reinterpret_cast<const f32x4*>(table + off);
          if (F & 16) v[p] = x; else v[p] = f32x4{x[0] * 0.f, 0, 0, 0};
        }
      }
      if ((F & 32) && g == 0) {
        const int* base = myidx + (s + 2 < stages ? s + 2 : s) * 32;
#pragma unroll
        for (int p = 0; p < 4; ++p) cols_nx[p] = base[rsub + 8 * p];
      }
      if (F & 64) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x3F6, 5, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      b0 = n0; b1 = n1; b2 = n2; b3 = n3;
    }
    if (F & 32) for (int p = 0; p < 4; ++p) cols[p] = cols_nx[p];
    if (F & 2) __syncthreads();
  }
  float sum = 0;
  for (auto& a : acc) sum += a[0] + a[1] + a[2] + a[3];
  for (auto& x : v) sum += x[0];
  for (auto& x : strip) sum += x;
  out[blockIdx.x * 256 + tid] = sum;
}

template <int F>
void run(float* d, float* table, int* idx, int wgs_per_cu, int stages) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  const size_t lds = (2 * 32 * 112 + 8 * 112) * 4;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    k<F><<<grid, 256, lds>>>(d, table, idx, stages);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = (double)grid * 4 * stages * 8 * 7 * 2048.0;  // nominal (7 tiles/group) so that times compare
  printf("F=%3d wgs/cu=%d  %.3f ms  %.1f TFLOP/s\n", F, wgs_per_cu, ms, flops / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, 4096 * 256 * 4);
  const int rows = 480189, stages = 1200;
  float* table;
  hipMalloc(&table, (size_t)rows * 400);
  hipMemset(table, 0, (size_t)rows * 400);
  int* idx;
  const size_t nidx = (size_t)256 * 5 * stages * 32;
  int* h = (int*)malloc(nidx * 4);
  srand(1);
  for (size_t i = 0; i < nidx; ++i) h[i] = rand() % rows;
  hipMalloc(&idx, nidx * 4);
  hipMemcpy(idx, h, nidx * 4, hipMemcpyHostToDevice);
  for (int w : {1, 3}) {
    run<1>(d, table, idx, w, stages);
    run<3>(d, table, idx, w, stages);
    run<7>(d, table, idx, w, stages);
    run<7 + 64>(d, table, idx, w, stages);
    run<15>(d, table, idx, w, stages);   // gathers issued, 1 dword consumed
    run<31>(d, table, idx, w, stages);   // full 16 B consumed a stage later (same row each stage)
    run<63>(d, table, idx, w, stages);   // + random rows with prefetched indices
    run<63 + 128>(d, table, idx, w, stages);
  }
  return 0;
}
