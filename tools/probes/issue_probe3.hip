// Issue model for the LU of lu_wave (round 4): does the fp32 rank-4 MFMA (v_mfma_f32_16x16x4_f32) leave room for VALU / DS
// work of the SAME wave, or of the OTHER wave of the SIMD?  And what do ds_bpermute / v_readlane / v_rcp cost?
//   same  : every wave runs MFMAs with NV filler instructions interleaved (NV / 8 behind each MFMA)
//   split : waves 0..3 of a 512-thread block run only the MFMAs, waves 4..7 only the fillers (wave w sits on SIMD w % 4:
//           one MFMA wave + one filler wave per SIMD)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe3.hip -o tools/_bin/issue_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { kFma, kPkFma, kCndmask, kBperm, kReadlane, kRcp, kDpp, kNone };
static const char* kOp[] = {"v_fma_f32", "v_pk_fma_f32", "v_cndmask_b32", "ds_bpermute_b32", "v_readlane_b32", "v_rcp_f32", "v_add_f32 dpp", "(none)"};
enum { kF32x4, kBf16K32, kBf16K16, kF32_32x32x2 };
static const char* kMf[] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x16_bf16", "v_mfma_f32_32x32x2_f32"};

template <int OP>
__device__ __forceinline__ void filler(float& v, f32x2& v2, float c1, float c2, int addr) {
  if constexpr (OP == kFma) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2));
  if constexpr (OP == kPkFma) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(v2));
  if constexpr (OP == kCndmask) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(c1));
  if constexpr (OP == kBperm) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(4)" : "+v"(v) : "v"(addr));
  if constexpr (OP == kReadlane) {
    int s;
    asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(v));
    asm volatile("" ::"s"(s));
  }
  if constexpr (OP == kRcp) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
  if constexpr (OP == kDpp) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
}

template <int MF>
__device__ __forceinline__ void mfma(f32x4& acc, float a, float b, bf16x8 pa, bf16x8 pb) {
  if constexpr (MF == kF32x4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  if constexpr (MF == kBf16K32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(pa), "v"(pb));
  if constexpr (MF == kBf16K16) {
    bf16x4 a4 = {pa[0], pa[1], pa[2], pa[3]}, b4 = {pb[0], pb[1], pb[2], pb[3]};
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a4), "v"(b4));
  }
}

// MODE 0 = same wave, 1 = split by wave (w < 4: MFMAs, w >= 4: fillers)
template <int MF, int NM, int NV, int OP, int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  f32x2 v2[8];
  for (int i = 0; i < 8; ++i) {
    v[i] = threadIdx.x * 0.001f + i + 1.f;
    v2[i] = f32x2{v[i], v[i] * 0.5f};
  }
  bf16x8 pa, pb;
  for (int i = 0; i < 8; ++i) {
    pa[i] = (short)(threadIdx.x + i);
    pb[i] = (short)(threadIdx.x * 3 + i);
  }
  asm volatile("" : "+v"(pa), "+v"(pb));
  float c1 = 1.0f + threadIdx.x * 1e-6f, c2 = threadIdx.x * 1e-3f;
  int addr = 4 * ((threadIdx.x * 7) & 63);
  asm volatile("" : "+v"(c1), "+v"(c2), "+v"(addr));
  const int w = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || w < 4, do_v = MODE == 0 || w >= 4;
  if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if constexpr (NM > 0) mfma<MF>(acc[m], c1, c2, pa, pb);
#pragma unroll
        for (int k = 0; k < NV / 8; ++k) filler<OP>(v[(m * (NV / 8) + k) & 7], v2[(m * (NV / 8) + k) & 7], c1, c2, addr);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if constexpr (NM > 0) mfma<MF>(acc[m], c1, c2, pa, pb);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < NV; ++k) filler<OP>(v[k & 7], v2[k & 7], c1, c2, addr);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + v2[i][0] + v2[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MF, int NM, int NV, int OP, int MODE>
static float run(int waves_per_simd, float* d) {
  const int iters = 2000, grid = 256, block = MODE == 1 ? 512 : 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MF, NM, NV, OP, MODE><<<grid, block>>>(d, 10);
  hipEventRecord(e0);
  probe<MF, NM, NV, OP, MODE><<<grid, block>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.4e9f / iters;  // cycles per loop iteration per SIMD at 2.4 GHz (relative numbers matter)
}

template <int MF, int OP>
static void row(float* d) {
  const float m1 = run<MF, 8, 0, kNone, 0>(1, d), m2 = run<MF, 8, 0, kNone, 0>(2, d);
  const float v16 = run<MF, 0, 16, OP, 0>(1, d), v48 = run<MF, 0, 48, OP, 0>(1, d);
  const float s16 = run<MF, 8, 16, OP, 0>(1, d), s48 = run<MF, 8, 48, OP, 0>(1, d);
  const float t16 = run<MF, 8, 16, OP, 0>(2, d), t48 = run<MF, 8, 48, OP, 0>(2, d);
  const float x16 = run<MF, 8, 16, OP, 1>(2, d), x48 = run<MF, 8, 48, OP, 1>(2, d);
  printf("%-26s + %-16s | 8 mfma: 1 wave %5.0f, 2 waves %5.0f | fillers alone (1 wave): 16 %5.0f, 48 %5.0f | same wave, 1 wave/SIMD: +16 %5.0f, +48 %5.0f | "
         "same wave, 2 waves/SIMD: +16 %5.0f, +48 %5.0f | split (MFMA wave + filler wave): 16 %5.0f, 48 %5.0f\n",
         kMf[MF], kOp[OP], m1, m2, v16, v48, s16, s48, t16, t48, x16, x48);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
  row<kF32x4, kFma>(d);
  row<kF32x4, kPkFma>(d);
  row<kF32x4, kCndmask>(d);
  row<kF32x4, kBperm>(d);
  row<kF32x4, kReadlane>(d);
  row<kF32x4, kRcp>(d);
  row<kF32x4, kDpp>(d);
  row<kBf16K32, kFma>(d);
  row<kBf16K32, kBperm>(d);
  row<kBf16K16, kFma>(d);
  return 0;
}
