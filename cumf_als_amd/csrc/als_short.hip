// als_short.hip -- CG half-iteration of SHORT rows without forming their Gram matrix (round 6).
//
// The reference forms A = sum_v theta_v theta_v^T + lambda n I per row (get_hermitian*, als.cu:443-659) and runs its CG on A
// (cg.cu:36-231): f^2 work per mat-vec whatever the row holds.  A row of n <= kShortRow ratings has A = T^T T + lambda n I
// with T its n x f block of gathered factor rows -- rank n -- and A p = T^T (T p) + lambda n p costs 2 n f: on the hugewiki
// X side (6.26 M rows per GPU, three quarters of them at or below 32 ratings, half of them below 16) the wave kernels spend
// a full 32-rating MFMA stage and six 112 x 112 mat-vecs (~3 000 VALU instructions) on every such row.  Here ONE wave holds T
// in the FEATURE layout only -- lane k = features k and 64 + k, T0[r] / T1[r] = theta_r[k] / theta_r[64 + k], 2 n registers:
//   u = T p      every lane forms its share T0[r] p0 + T1[r] p1 of all N (8 / 16 / 32 >= n) dot products, and ONE transposing
//                wave reduction sums them: each butterfly step halves the registers (the lane's bit picks which of a pair it
//                keeps and which it hands over), so N registers cost ~3 N cross-lane instructions instead of 7 N, and lane r
//                ends with u_r;
//   y = T^T u    u_r by v_readlane into an SGPR, two FMAs per rating, the loop leaves after the row's last rating;
// the vectors of the CG live in the feature layout (two registers each), dot products are one DPP wave reduction; no LDS
// memory, no MFMA, ~125 registers = four waves per SIMD (the kernel is a chain of dependent reductions: occupancy is what
// hides them).  Same recurrence, same warm start, same exit test as cg_wave_core / cg.cu; the rounding differs from the Gram
// route (the reference rounds the entries of A first), parity is by the CG tolerance of the tests.  The right-hand side is
// T^T r, the fused train SSE is S - x.b - x.r - lambda n |x|^2 with S = sum r^2 as in cg_wave_core.
// Items: the plan lists its short whole rows last (als_plan.cpp); launch_half_iteration hands them to this kernel when the
// solver is CG and 32 <= f <= 128 (of the wave kernels' range f <= 111; BASELINE configs[3] is f = 100).
#include <hip/hip_runtime.h>

#include "als_device.h"
#include "als_internal.h"

namespace cumf {

namespace {

// zeros that stand in for the factor row of a lane without a rating
__device__ __attribute__((aligned(16))) float g_short_zeros[256];

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               0xf, BANK_MASK, false));
}
// the value of lane L ^ X (X = 1, 2, 4, 8, 16, 32)
template <int X>
__device__ __forceinline__ float lane_xor(float v, int lane) {
  if constexpr (X == 1) return dpp_move<0xB1, 0xf>(0.f, v);   // quad_perm [1,0,3,2]
  if constexpr (X == 2) return dpp_move<0x4E, 0xf>(0.f, v);   // quad_perm [2,3,0,1]
  if constexpr (X == 4)                                        // row_shr:4 into the banks with bit 2 set, row_shl:4 into the others
    return dpp_move<0x104, 0x5>(dpp_move<0x114, 0xa>(0.f, v), v);
  if constexpr (X == 8) return dpp_move<0x128, 0xf>(0.f, v);  // row_ror:8
  if constexpr (X == 16) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
  if constexpr (X == 32)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __builtin_bit_cast(int, v)));
}
// Sums of N registers over the wave, transposed: lane L ends with the full sum of register L mod N.
template <int N>
__device__ __forceinline__ float reduce_transposed(float (&P)[N], int lane) {
  static_assert(N == 8 || N == 16 || N == 32, "register count");
  auto step = [&](auto xc, auto mc) {  // M registers -> M / 2: the lane keeps register 2 m + bit, hands 2 m + !bit to lane ^ X
    constexpr int X = decltype(xc)::value, M = decltype(mc)::value;
    const bool bit = (lane & X) != 0;
#pragma unroll
    for (int m = 0; m < M / 2; ++m) {
      const float keep = bit ? P[2 * m + 1] : P[2 * m], send = bit ? P[2 * m] : P[2 * m + 1];
      P[m] = keep + lane_xor<X>(send, lane);
    }
  };
  auto ic = [](auto v) { return v; };
  (void)ic;
  step(std::integral_constant<int, 1>{}, std::integral_constant<int, N>{});
  step(std::integral_constant<int, 2>{}, std::integral_constant<int, N / 2>{});
  step(std::integral_constant<int, 4>{}, std::integral_constant<int, N / 4>{});
  float v;
  if constexpr (N == 8) {
    v = P[0];
    v += lane_xor<8>(v, lane);
    v += lane_xor<16>(v, lane);
  } else if constexpr (N == 16) {
    step(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
    v = P[0];
    v += lane_xor<16>(v, lane);
  } else {
    step(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
    v = P[0];
  }
  return v + lane_xor<32>(v, lane);
}

// the whole row with at most N ratings in flight (N = 8, 16, 32 >= n); TWO: f > 64, the lanes hold two features each
template <bool TWO, int N>
__device__ __forceinline__ void short_cg_row(const KernelArgs& a, int row, int n, float rv, const float* grow, int lane) {
  const int f = a.f;
  const bool f0 = lane < f, f1 = TWO && lane + 64 < f;  // this lane's features exist
  float* xg = a.update + (size_t)row * f;
  // warm start (cg.cu:48)
  float x0 = f0 ? xg[lane] : 0.f, x1 = f1 ? xg[64 + lane] : 0.f;
  // feature layout: one rating at a time, coalesced; ratings past n read the zero row
  float T0[N], T1[N];
  const unsigned long long gaddr = reinterpret_cast<unsigned long long>(grow);
  const int glo = (int)(unsigned)gaddr, ghi = (int)(unsigned)(gaddr >> 32);
  static_for<N>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane(glo, r), hi = (unsigned)__builtin_amdgcn_readlane(ghi, r);
    typedef __attribute__((address_space(1))) const float gfloat;  // global loads, not flat ones
    gfloat* base = reinterpret_cast<gfloat*>(((unsigned long long)hi << 32) | lo);
    T0[r] = f0 ? base[lane] : 0.f;
    T1[r] = f1 ? base[64 + lane] : 0.f;
  });
  const float reg = (float)n * a.lambda;  // lambda * n_u on the diagonal (als.cu:545-557)
  // y += T^T w for a rating-layout vector w (lane r: w_r)
  auto tt_product = [&](float w, float& y0, float& y1) {
    static_for<N / 4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if (4 * q < n) {  // uniform
        static_for<4>([&](auto ic) {
          constexpr int r = 4 * q + decltype(ic)::value;
          const float wr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), r));
          y0 = fmaf(T0[r], wr, y0);
          if constexpr (TWO) y1 = fmaf(T1[r], wr, y1);
        });
      }
    });
  };
  // y = A v = T^T (T v) + reg v
  auto matvec = [&](float v0, float v1, float& y0, float& y1) {
    float P[N];
    static_for<N>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      P[r] = TWO ? fmaf(T1[r], v1, T0[r] * v0) : T0[r] * v0;
    });
    const float u = reduce_transposed<N>(P, lane);
    y0 = reg * v0, y1 = reg * v1;
    tt_product(u, y0, y1);
  };
  auto dot = [&](float a0, float a1, float c0, float c1) { return wave_sum_uniform(TWO ? fmaf(a1, c1, a0 * c0) : a0 * c0); };
  // right-hand side b = T^T r (als.cu:750-757)
  float b0 = 0.f, b1 = 0.f;
  tt_product(rv, b0, b1);
  // ---- CG (cg.cu:36-231)
  float ap0, ap1;
  matvec(x0, x1, ap0, ap1);
  float r0 = b0 - ap0, r1 = b1 - ap1;
  float p0 = r0, p1 = r1;
  float rsold = dot(r0, r1, r0, r1);
  for (int iter = 0; iter < a.cg_iters; ++iter) {
    matvec(p0, p1, ap0, ap1);
    const float pap = dot(p0, p1, ap0, ap1);
    const float alpha = rsold / pap;
    x0 = fmaf(alpha, p0, x0), x1 = fmaf(alpha, p1, x1);
    r0 = fmaf(-alpha, ap0, r0), r1 = fmaf(-alpha, ap1, r1);
    const float rsnew = dot(r0, r1, r0, r1);
    if ((double)rsnew < 1e-4) break;  // CG_ERROR (cg.cu:31,195)
    const float beta = rsnew / rsold;
    rsold = rsnew;
    p0 = fmaf(beta, p0, r0), p1 = fmaf(beta, p1, r1);
  }
  if (f0) xg[lane] = x0;
  if (f1) xg[64 + lane] = x1;
  if (a.sse_bins != nullptr) {  // fused train SSE, as cg_wave_core: S - x.b - x.r - reg |x|^2
    const float S = wave_sum_uniform(rv * rv);
    const float xb = dot(x0, x1, b0, b1), xr = dot(x0, x1, r0, r1), xx = dot(x0, x1, x0, x1);
    if (lane == 0 && n > 0)
      atomicAdd(a.sse_bins + (blockIdx.x & (kSseBins - 1)), (double)S - (double)xb - (double)xr - (double)reg * (double)xx);
  }
}

template <bool TWO>
__global__ __launch_bounds__(64, 4) void als_short_cg_kernel(const KernelArgs a) {
  static_assert(kShortRow == 32, "ratings per row at most");
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int row = a.item_row[item];
  const long long begin = a.item_begin[item];
  const int n = a.item_len[item];  // the whole row (uniform)
  const bool live = lane < n;
  const int j = live ? a.colidx[begin + lane] : 0;
  const float rv = live ? a.val[begin + lane] : 0.f;
  const float* grow = live ? a.gather_f32 + (size_t)j * a.f : g_short_zeros;
  if (n <= 8)
    short_cg_row<TWO, 8>(a, row, n, rv, grow, lane);
  else if (n <= 16)
    short_cg_row<TWO, 16>(a, row, n, rv, grow, lane);
  else
    short_cg_row<TWO, 32>(a, row, n, rv, grow, lane);
}

}  // namespace

// any f of the wave kernels' range up to two features per lane (the factored mat-vec costs 2 n f against f^2: it pays from
// f = 32 on for every n <= 32)
bool short_cg_available(int f) { return f >= 32 && f <= 128; }

// items [0, n_items) of the lists in `a` (the caller has advanced the list pointers to the plan's short rows)
hipError_t launch_short_cg(const KernelArgs& a, long n_items, hipStream_t stream) {
  if (n_items <= 0) return hipSuccess;
  if (!short_cg_available(a.f)) return hipErrorInvalidValue;
  if (a.f > 64)
    hipLaunchKernelGGL((als_short_cg_kernel<true>), dim3((unsigned)n_items), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((als_short_cg_kernel<false>), dim3((unsigned)n_items), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace cumf
