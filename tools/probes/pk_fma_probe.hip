// Is v_pk_fma_f32 trustworthy beside MFMA work on gfx950?  (VERDICT r03 missing 5: round 3 saw wrong CG mat-vecs with the
// packed form on 1-4 % of the long Netflix X rows, a different set each run, and left it "unexplained".)
// Every 512-thread block puts two waves on each SIMD: waves 0..3 run MFMA bursts (bf16 or fp32 rank-4), waves 4..7 run the
// shape of the CG mat-vec -- chains of packed FMAs on register pairs whose results feed DPP row reductions and a
// ds_bpermute -- and, beside them, the same arithmetic with scalar v_fma_f32 on separate registers.  Both are IEEE FMAs per
// component: the results must agree bit for bit.  Mismatching lanes are counted per launch.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_fma_probe.hip -o tools/_bin/pk_fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float sfma(float a, float b, float c) {  // a scalar v_fma_f32, whatever the vectoriser thinks
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int MF>
__global__ __launch_bounds__(512) void probe(unsigned* mismatches, float* sink, int iters, int seed) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (w < 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) {
      pa[i] = (short)(0x3f80 + ((lane + i) & 7));
      pb[i] = (short)(0x3f00 + ((lane * 3 + i) & 7));
    }
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if constexpr (MF == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(pa), "v"(pb));
        if constexpr (MF == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
      }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    return;
  }
  // the CG mat-vec's shape: 7 tiles of 4 registers as two pairs, times a vector element, accumulated in pairs; then a
  // 16-lane DPP reduction and a ds_bpermute of the result
  unsigned bad = 0, mask = 0;
  float s0_last[12] = {0};
  float t[7][4], v[7];
  unsigned h = (unsigned)(seed * 2654435761u) ^ (blockIdx.x * 97u + threadIdx.x * 131u);
  auto rnd = [&]() {
    h = h * 1664525u + 1013904223u;
    return ((int)(h >> 9) - (1 << 22)) * (1.0f / (1 << 22));
  };
  for (int j = 0; j < 7; ++j) {
    v[j] = rnd();
    for (int r = 0; r < 4; ++r) t[j][r] = rnd();
  }
  for (int it = 0; it < iters; ++it) {
    f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const f32x2 vj = {v[j], v[j]};
      p01 = pk_fma(f32x2{t[j][0], t[j][1]}, vj, p01);
      p23 = pk_fma(f32x2{t[j][2], t[j][3]}, vj, p23);
      s0 = sfma(t[j][0], v[j], s0);
      s1 = sfma(t[j][1], v[j], s1);
      s2 = sfma(t[j][2], v[j], s2);
      s3 = sfma(t[j][3], v[j], s3);
    }
    const float rp = row16_sum(p01[0]) + row16_sum(p01[1]) + row16_sum(p23[0]) + row16_sum(p23[1]);
    const float rs = row16_sum(s0) + row16_sum(s1) + row16_sum(s2) + row16_sum(s3);
    const float bp = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 16), __builtin_bit_cast(int, rp)));
    const float bs = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 16), __builtin_bit_cast(int, rs)));
    // (components go through floats first: __builtin_bit_cast on a vector-element lvalue reads element 0)
    const float q0 = p01[0], q1 = p01[1], q2 = p23[0], q3 = p23[1];
    const unsigned m = (unsigned)(__builtin_bit_cast(unsigned, q0) != __builtin_bit_cast(unsigned, s0)) |
                       (unsigned)(__builtin_bit_cast(unsigned, q1) != __builtin_bit_cast(unsigned, s1)) << 1 |
                       (unsigned)(__builtin_bit_cast(unsigned, q2) != __builtin_bit_cast(unsigned, s2)) << 2 |
                       (unsigned)(__builtin_bit_cast(unsigned, q3) != __builtin_bit_cast(unsigned, s3)) << 3 |
                       (unsigned)(__builtin_bit_cast(unsigned, rp) != __builtin_bit_cast(unsigned, rs)) << 4 |
                       (unsigned)(__builtin_bit_cast(unsigned, bp) != __builtin_bit_cast(unsigned, bs)) << 5;
    bad += m != 0;
    mask |= m;
    if (bad != 0 && s0_last[11] == 0.f) {
      s0_last[0] = p01[0], s0_last[1] = s0, s0_last[2] = p01[1], s0_last[3] = s1, s0_last[4] = p23[0], s0_last[5] = s2;
      s0_last[6] = p23[1], s0_last[7] = s3, s0_last[8] = rp, s0_last[9] = rs, s0_last[10] = bp, s0_last[11] = bs;
    }
    // new data for the next round, derived from the results (keeps everything live and varying)
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = v[j] * 0.75f + bs * 1e-3f + 0.01f * (float)(j - 3);
  }
  if (bad) {
    if (atomicAdd(mismatches, bad) == 0) {  // the first reporter leaves its last values for inspection
      float* dbg = sink + 1024 * 512;
      dbg[0] = s0_last[0], dbg[1] = s0_last[1], dbg[2] = s0_last[2], dbg[3] = s0_last[3];
      dbg[4] = s0_last[4], dbg[5] = s0_last[5], dbg[6] = s0_last[6], dbg[7] = s0_last[7];
      dbg[8] = s0_last[8], dbg[9] = s0_last[9], dbg[10] = s0_last[10], dbg[11] = s0_last[11];
      dbg[12] = (float)mask, dbg[13] = (float)(threadIdx.x), dbg[14] = (float)bad;
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = v[0];
}

int main() {
  unsigned* d_bad;
  float* d_sink;
  hipMalloc(&d_bad, sizeof(unsigned));
  hipMalloc(&d_sink, (1024 * 512 + 16) * sizeof(float));
  const char* names[2] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32"};
  for (int mf = 0; mf < 2; ++mf) {
    unsigned long long total = 0, checks = 0;
    for (int rep = 0; rep < 20; ++rep) {
      hipMemset(d_bad, 0, sizeof(unsigned));
      const int iters = 20000;
      if (mf == 0) probe<0><<<1024, 512>>>(d_bad, d_sink, iters, rep);
      else probe<1><<<1024, 512>>>(d_bad, d_sink, iters, rep);
      unsigned bad = 0;
      hipMemcpy(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost);
      if (bad && total == 0) {
        float dbg[16];
        hipMemcpy(dbg, d_sink + 1024 * 512, sizeof(dbg), hipMemcpyDeviceToHost);
        printf("  first reporter, its first mismatching round (packed, scalar): p0 %.9g %.9g  p1 %.9g %.9g  p2 %.9g %.9g  p3 %.9g %.9g  row sum %.9g %.9g  "
               "bpermute %.9g %.9g\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10], dbg[11]);
        printf("  mask of failing compares %d (1,2,4,8: the four components, 16: row sum, 32: bpermute), thread %d, rounds %d\n", (int)dbg[12], (int)dbg[13], (int)dbg[14]);
      }
      total += bad;
      checks += 1024ull * 256 * iters;
    }
    printf("packed FMAs beside %-26s: %llu mismatching lane-rounds out of %.3g (28 v_pk_fma_f32 each)\n", names[mf], total,
           (double)checks);
  }
  return 0;
}
