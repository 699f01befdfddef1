// als_device.h -- device helpers shared by als_kernels.hip (workgroup-per-item kernels) and
// als_wave.hip (wave-per-item kernels): tile enumeration, compile-time loops, DPP reductions, the
// packed upper-triangular row store of the LU paths and its back substitution.
#ifndef CUMF_ALS_DEVICE_H_
#define CUMF_ALS_DEVICE_H_

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "als_internal.h"

namespace cumf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Row-major enumeration of the upper triangle: t -> (I, J), I <= J < NB.
template <int NB>
__host__ __device__ constexpr int tile_I(int t) {
  int I = 0, rem = t;
  while (rem >= NB - I) {
    rem -= NB - I;
    ++I;
  }
  return I;
}
template <int NB>
__host__ __device__ constexpr int tile_J(int t) {
  int I = 0, rem = t;
  while (rem >= NB - I) {
    rem -= NB - I;
    ++I;
  }
  return I + rem;
}

// Compile-time loop: body(std::integral_constant<int, i>) for i in [0, N).
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& body, std::integer_sequence<int, Is...>) {
  (body(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& body) {
  static_for_impl(body, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Packed row store of the register LU: block row kb (rows 16 kb .. 16 kb + 15) keeps columns
// 16 kb .. 16 NB - 1 only, at an odd pitch (the back substitution walks columns).  For NB = 7
// that is 7 280 floats instead of the 10 100 of a full f x (f + 1) matrix, which is what lets
// a fifth workgroup share the CU.
template <int NB>
__host__ __device__ constexpr int lu_row_pitch(int kb) { return 16 * (NB - kb) + 1; }
template <int NB>
__host__ __device__ constexpr int lu_block_off(int kb) { return 256 * (kb * NB - kb * (kb - 1) / 2) + 16 * kb; }
// offset of the (virtual) element (k, 0); valid for columns j >= 16 * (k >> 4)
template <int NB>
__device__ __forceinline__ int lu_row_off(int k) {
  const int kb = k >> 4, kk = k & 15;
  return 256 * (kb * NB - ((kb * (kb - 1)) >> 1)) + kk * (16 * (NB - kb) + 1);
}
template <int NB>
__host__ __device__ constexpr int tile_of(int I, int J) { return I * NB - I * (I - 1) / 2 + (J - I); }
// row m of a 16 x 16 tile parked in LDS sits at slot 4 * (m & 3) + (m >> 2): the accumulator
// store (lane group kk writes rows 4 kk + r) then spreads over all 32 banks.
__device__ __forceinline__ int tiled_row(int m) { return 4 * (m & 3) + (m >> 2); }

// Deterministic wave64 sum on the DPP cross-lane network (no LDS traffic): xor-1, xor-2,
// half-mirror, mirror inside each row of 16, then row_bcast15 / row_bcast31 across rows;
// lane 63 ends with the total, returned wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
  v += dpp_term<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_term<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_term<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_term<0x140, 0xf>(v);  // row_mirror  -> every lane holds its row's sum
  v += dpp_term<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3 += rows 0,2
  v += dpp_term<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3 += row 1
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Back substitution of the fast LU paths, U x = y with y = column f of the packed row store and the
// reciprocals of the diagonal in rdiag (one wave; lane i holds rows i, i + 64, ...).  The
// recurrence is a chain of f dependent steps, so everything that does not depend on the running
// vector is moved off it: row i is scaled by 1/u_ii (z_i = y_i / u_ii, v_ik = u_ik / u_ii: unit
// diagonal, x_k = z_k needs no multiply), and the work is organised by 16-pivot blocks on a row
// store whose entries at or left of the diagonal inside the diagonal blocks are zero (the
// publishers write zeros there): the lanes of the pivot block then need no triangle mask, the
// lanes of later blocks are pointed at 16 zeros (zpad) once per block, the 16 columns of a block
// are read with immediate offsets from one base per lane, one block ahead of their use.  Per
// step that leaves v_readlane, one packed multiply (the 1/u_ii scaling) and one packed FMA.
template <int NB, int NQ>
__device__ __forceinline__ void back_substitute_zeroed(const float* __restrict__ U, int f,
                                                       const float* __restrict__ rdiag,
                                                       const float* __restrict__ zpad,
                                                       float* __restrict__ x_global, int lane) {
  float z[NQ], rdl[NQ];
  const float* rowp[NQ];
  int ib[NQ];  // block of this lane's row
  static_for<NQ>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int i = lane + 64 * q;
    const int ic = i < f ? i : f - 1;
    ib[q] = i < f ? (i >> 4) : 1 << 20;  // rows past f never take part
    rowp[q] = U + lu_row_off<NB>(ic);
    rdl[q] = i < f ? rdiag[ic] : 0.f;
    z[q] = rowp[q][f] * rdl[q];
  });
  const int top = f - 1;
  constexpr int NBLK = NB;  // pivot blocks 0 .. (f-1)>>4 <= NB-1
  // two column buffers: the 16 columns of the next block are in flight while this one is consumed
  float col[2][16][NQ];
  auto issue = [&](auto kbc, auto bufc) {
    constexpr int kb = decltype(kbc)::value, buf = decltype(bufc)::value, Q = kb >> 2;
    // per lane: the 16 entries of its row in the block's columns (zeros for rows of later blocks)
    const float* base[Q + 1];
    static_for<Q + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      base[q] = (ib[q] > kb) ? zpad : rowp[q] + 16 * kb;
    });
    static_for<16>([&](auto jc) {  // issued in the order they are consumed (LDS returns in order)
      constexpr int j = 15 - decltype(jc)::value;
      static_for<Q + 1>([&](auto qc) { col[buf][j][decltype(qc)::value] = base[decltype(qc)::value][j]; });
    });
  };
  static_for<NBLK>([&](auto bc) {
    constexpr int n = decltype(bc)::value;
    constexpr int kb = NBLK - 1 - n;
    constexpr int buf = n & 1;
    constexpr int Q = kb >> 2;  // pivots of this block live in z[Q]
    if constexpr (Q < NQ) {
      if (kb == (top >> 4)) issue(std::integral_constant<int, kb>{}, std::integral_constant<int, buf>{});
      if (16 * kb <= top) {
        if constexpr (kb > 0) issue(std::integral_constant<int, kb - 1>{}, std::integral_constant<int, buf ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto jc) {
          constexpr int j = 15 - decltype(jc)::value;
          const int k = 16 * kb + j;
          if (k <= top) {  // uniform; only the last block can be short
            const float xk = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[Q]), k & 63));
            static_for<Q + 1>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              z[q] = fmaf(-(col[buf][j][q] * rdl[q]), xk, z[q]);
            });
          }
        });
      }
    }
  });
  static_for<NQ>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    if (lane + 64 * q < f) x_global[lane + 64 * q] = z[q];
  });
}

// Entry (f, f) of the augmented system -- the rating rides in slot f of the gathered rows, so the Gram pass accumulates
// sum r^2 there (the fused train SSE starts from it, als_wave.hip) -- out of the last diagonal tile, wave-uniform.
template <int NB>
__device__ __forceinline__ float wave_tile_ff(const f32x4& last_diag, int f) {
  const int cf = f - 16 * (NB - 1);  // slot f inside the last block: lane (cf >> 2, cf), register cf & 3
  float v = last_diag[0];
  v = (cf & 3) == 1 ? last_diag[1] : v;
  v = (cf & 3) == 2 ? last_diag[2] : v;
  v = (cf & 3) == 3 ? last_diag[3] : v;
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * (cf >> 2) + cf));
}
}  // namespace cumf

#endif  // CUMF_ALS_DEVICE_H_
