// als_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ALS solve path.
//
// What the reference does per half-iteration (als.cu:727-964):
//   b   = R * Theta            cusparseScsrmm2 + cublasSgeam   (als.cu:750-757)
//   A_u = sum theta theta^T + lambda n_u I   get_hermitian100 / get_hermitianT10
//                                            (als.cu:443-569 / 575-659), one CUDA
//                                            block per row, 10x10 register tiles
//   x_u = A_u^-1 b_u           updateXWithCGKernel (cg.cu:36-231) or cuBLAS batched LU
// with the f x f Gram batch written to and re-read from device memory.
//
// What this file does instead (MI355X-first):
//   * one pass gathers each factor row ONCE into LDS (16-byte loads, zero padded
//     to 16-wide feature blocks, the rating value parked in feature slot f);
//   * the rank-k update theta theta^T is a SYRK on the fp32 matrix cores:
//     v_mfma_f32_16x16x4_f32 over the upper-triangular 16x16 tiles only.  One VGPR
//     per (feature block, 4 ratings) is both the A and the B operand.  The RHS
//     b = sum r theta falls out of column f of the last tile column for free;
//   * rows are cut into chunks (plan, als_plan.cpp) so heavy rows spread over
//     many workgroups; whole rows are solved in the same workgroup straight out of
//     LDS (CG or unpivoted LU) -- the Gram never touches HBM; chunked rows go
//     through a deterministic partial-tile reduction kernel;
//   * fp32 MFMA is an exact k-ordered fmaf chain, so a whole-row Gram entry is the
//     same sequential FMA chain one reference thread computes (als.h:39-143).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>
#include <utility>

#include "als_internal.h"

namespace cumf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ----------------------------------------------------------------------------------
// Geometry of one workgroup (256 threads = 4 waves) for NB 16-wide feature blocks.
// ----------------------------------------------------------------------------------
template <int NB>
struct Geo {
  static constexpr int NT = NB * (NB + 1) / 2;  // upper-triangular tiles
  static constexpr int TPW = (NT + 3) / 4;      // tiles per wave (T-split over the 4 waves)
  // Stage row pitch in floats.  LD % 32 == 16 makes the MFMA operand read
  // (lane = 16*kk + c reads stage[4g+kk][16B+c]) conflict-free for ds_read_b32,
  // whose lane groups are {0-31},{32-63} over 32 banks.
  static constexpr int LD = 16 * NB + ((NB % 2 == 0) ? 16 : 0);
};

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

// ----------------------------------------------------------------------------------
// Global -> register -> LDS staging of kStage gathered factor rows.
// VT is float4 when f % 4 == 0 (16-byte loads; f = 100: 25 loads per row) else float2
// (f % 10 == 0 guarantees f even, main.cpp:33).
// ----------------------------------------------------------------------------------
template <int NB, typename VT>
struct Stager {
  static constexpr int VW = sizeof(VT) / 4;
  static constexpr int PPR = Geo<NB>::LD / VW;  // vector pieces per stage row
  static constexpr int TOTAL = kStage * PPR;
  static constexpr int PASSES = (TOTAL + kThreads - 1) / kThreads;
  VT v[PASSES];

  // Rows [0, nvalid) of the stage come from ratings [begin, begin + nvalid).
  __device__ __forceinline__ void load(const int* __restrict__ colidx, const float* __restrict__ val,
                                       const float* __restrict__ gather, int f, long long begin,
                                       int nvalid, int tid) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int q = tid + p * kThreads;
      const int r = q / PPR;
      const int col0 = (q - r * PPR) * VW;
      VT x = {};
      if (q < TOTAL && r < nvalid) {
        if (col0 < f) {
          const int c = colidx[begin + r];
          x = *reinterpret_cast<const VT*>(gather + (size_t)c * f + col0);
        } else if (col0 == f) {
          x[0] = val[begin + r];  // rating value rides in feature slot f -> RHS from the MFMA
        }
      }
      v[p] = x;
    }
  }

  // Rows [nvalid, nwrite) are written as zeros (nwrite = nvalid rounded up to 4).
  __device__ __forceinline__ void store(float* __restrict__ stage, int nwrite, int tid) const {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int q = tid + p * kThreads;
      const int r = q / PPR;
      const int col0 = (q - r * PPR) * VW;
      if (q < TOTAL && r < nwrite) *reinterpret_cast<VT*>(stage + r * Geo<NB>::LD + col0) = v[p];
    }
  }
};

// ----------------------------------------------------------------------------------
// SYRK of one stage on the matrix cores.  Wave W owns tiles [W*TPW, (W+1)*TPW).
//   D[i][j] += sum_k A[i][k] B[k][j],  A[i][k] = theta_k[16I+i], B[k][j] = theta_k[16J+j]
// v_mfma_f32_16x16x4_f32 operand layout: lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; both are stage[4g + (l>>4)][16*blk + (l&15)], so a feature
// block's register serves as the A operand of its tile row and the B operand of its
// tile column.  Accumulation is the exact k-ordered fmaf chain (rating order).
// ----------------------------------------------------------------------------------
template <int NB, int W>
__device__ __forceinline__ void mma_stage(const float* __restrict__ stage, f32x4 (&acc)[Geo<NB>::TPW],
                                          int ngroups, int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW, LD = Geo<NB>::LD;
  const float* rowp = stage + (lane >> 4) * LD + (lane & 15);
  for (int g = 0; g < ngroups; ++g) {
    float blk[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) blk[b] = rowp[16 * b];  // blocks this wave never uses are dead code
    static_for<TPW>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int t = W * TPW + s;
      if constexpr (t < NT) {
        constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(blk[I], blk[J], acc[s], 0, 0, 0);
      }
    });
    rowp += 4 * LD;
  }
}

// Accumulator tile -> LDS system matrix G (f x ldg, column f = RHS).  C/D layout of
// the 16x16 MFMA: lane l, register r holds D[4*(l>>4) + r][l & 15].
template <int NB, int W>
__device__ __forceinline__ void tiles_to_lds(const f32x4 (&acc)[Geo<NB>::TPW], float* __restrict__ G,
                                             int ldg, int f, int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  const int c = lane & 15, kk = lane >> 4;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = W * TPW + s;
    if constexpr (t < NT) {
      constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * kk + r, j = 16 * J + c;
        const float v = acc[s][r];
        if (i < f && j <= f) G[i * ldg + j] = v;           // j == f: b_i = sum r * theta[i]
        if (I != J && i < f && j < f) G[j * ldg + i] = v;  // mirror
      }
    }
  });
}

// Accumulator tile -> row-major f x f Gram in global memory (both triangles,
// lambda * n on the diagonal: als.cu:545-566) + RHS.
template <int NB, int W>
__device__ __forceinline__ void tiles_to_global(const f32x4 (&acc)[Geo<NB>::TPW], float* __restrict__ tt,
                                                float* __restrict__ rhs, int f, float reg, int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  const int c = lane & 15, kk = lane >> 4;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = W * TPW + s;
    if constexpr (t < NT) {
      constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * kk + r, j = 16 * J + c;
        float v = acc[s][r];
        if (i < f && j < f) {
          if (i == j) v += reg;
          tt[(size_t)i * f + j] = v;
          if (I != J) tt[(size_t)j * f + i] = v;
        } else if (i < f && j == f && rhs != nullptr) {
          rhs[i] = v;
        }
      }
    }
  });
}

// Partial tiles <-> global scratch, in accumulator layout ([slot][tile][reg][lane]:
// every store/load is one coalesced 256-byte wave access).
template <int NB, int W>
__device__ __forceinline__ void tiles_to_partial(const f32x4 (&acc)[Geo<NB>::TPW], float* __restrict__ part,
                                                 int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = W * TPW + s;
    if constexpr (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[((size_t)t * 4 + r) * 64 + lane] = acc[s][r];
    }
  });
}
template <int NB, int W>
__device__ __forceinline__ void partial_accumulate(f32x4 (&acc)[Geo<NB>::TPW], const float* __restrict__ part,
                                                   int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = W * TPW + s;
    if constexpr (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[s][r] += part[((size_t)t * 4 + r) * 64 + lane];
    }
  });
}

// ----------------------------------------------------------------------------------
// In-LDS solvers.  G is f x ldg (ldg = f + 1, odd => column walks are conflict-free),
// column f holds b.  256 threads.
// ----------------------------------------------------------------------------------

// Conjugate gradient exactly as cg.cu:36-231: warm start, r = b - A x, <= cg_iters
// iterations, stop when ||r||^2 < 1e-4 (CG_ERROR, cg.cu:31,195; the float is compared
// against the double literal).  Dot products are deterministic wave butterflies that
// every wave evaluates redundantly (same bits in every wave => uniform branch), in
// place of the reference's order-dependent smem atomics (device_utilities.h:36-48).
// vec: 6 * kVecLd floats of LDS.  Requires f <= 128.
__device__ __forceinline__ void cg_solve_lds(const float* __restrict__ G, int ldg, int f,
                                             float* __restrict__ vec, float* __restrict__ x_global,
                                             int cg_iters, int tid) {
  float* xs = vec;
  float* rs = vec + kVecLd;
  float* ps = vec + 2 * kVecLd;
  float* aps = vec + 3 * kVecLd;
  float* part = vec + 4 * kVecLd;  // [2][kVecLd]
  const int i = tid & 127, h = tid >> 7, lane = tid & 63;
  const int jh = (f + 1) >> 1;
  const int j0 = h ? jh : 0, j1 = h ? f : jh;

  auto matvec = [&](const float* __restrict__ v) {
    float s = 0.f;
    if (i < f) {
      for (int j = j0; j < j1; ++j) s = fmaf(G[j * ldg + i], v[j], s);  // A symmetric: column i == row i (cg.cu:55)
    }
    part[h * kVecLd + i] = s;
  };
  auto dot = [&](const float* __restrict__ a, const float* __restrict__ b) {
    float s = 0.f;
    for (int j = lane; j < f; j += 64) s = fmaf(a[j], b[j], s);
    return wave_sum(s);
  };

  if (tid < f) xs[tid] = x_global[tid];
  __syncthreads();
  matvec(xs);
  __syncthreads();
  if (tid < f) {
    const float r = G[tid * ldg + f] - (part[tid] + part[kVecLd + tid]);
    rs[tid] = r;
    ps[tid] = r;
  }
  __syncthreads();
  float rsold = dot(rs, rs);
  for (int iter = 0; iter < cg_iters; ++iter) {
    matvec(ps);
    __syncthreads();
    if (tid < f) aps[tid] = part[tid] + part[kVecLd + tid];
    __syncthreads();
    const float pap = dot(ps, aps);
    const float alpha = rsold / pap;
    if (tid < f) {
      xs[tid] = fmaf(alpha, ps[tid], xs[tid]);
      rs[tid] = fmaf(-alpha, aps[tid], rs[tid]);
    }
    __syncthreads();
    const float rsnew = dot(rs, rs);
    if ((double)rsnew < 1e-4) break;
    const float beta = rsnew / rsold;
    rsold = rsnew;
    if (tid < f) ps[tid] = fmaf(beta, ps[tid], rs[tid]);
    __syncthreads();
  }
  if (tid < f) x_global[tid] = xs[tid];
}

// Unpivoted Gaussian elimination on the augmented system [A | b] followed by back
// substitution: the mathematical content of cublasSgetrfBatched(PivotArray = NULL) +
// cublasSgetrsBatched (als.cu:77,98).  Same operation order as oracle_lu
// (right-looking, IEEE division by the pivot, fmaf updates, descending back
// substitution), so the result is bit-identical to the oracle on identical A, b.
__device__ __forceinline__ void lu_solve_lds(float* __restrict__ G, int ldg, int f,
                                             float* __restrict__ x_global, int tid) {
  const int ti = tid >> 4, tj = tid & 15;
  for (int k = 0; k < f; ++k) {
    const float piv = G[k * ldg + k];
    for (int i = k + 1 + tid; i < f; i += kThreads) G[i * ldg + k] = G[i * ldg + k] / piv;
    __syncthreads();
    for (int i = k + 1 + ti; i < f; i += 16) {
      const float l = G[i * ldg + k];
      for (int j = k + 1 + tj; j <= f; j += 16) G[i * ldg + j] = fmaf(-l, G[k * ldg + j], G[i * ldg + j]);
    }
    __syncthreads();
  }
  if (tid < 64) {
    const int lane = tid;
    float y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = (lane + 64 * q < f) ? G[(lane + 64 * q) * ldg + f] : 0.f;
    for (int k = f - 1; k >= 0; --k) {
      const int kq = k >> 6, kl = k & 63;
      float yk = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q == kq) yk = __shfl(y[q], kl);
      const float xk = yk / G[k * ldg + k];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        if (i == k)
          y[q] = xk;
        else if (i < k)
          y[q] = fmaf(-G[i * ldg + k], xk, y[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < f) x_global[lane + 64 * q] = y[q];
  }
}

// ----------------------------------------------------------------------------------
// Finish one row whose complete accumulator tiles sit in `acc`.
// ----------------------------------------------------------------------------------
template <int NB, int MODE>
__device__ __forceinline__ void finish_row(f32x4 (&acc)[Geo<NB>::TPW], float* smem, const KernelArgs& a,
                                           int row, int rowlen, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int f = a.f;
  // als.cu:547: float temp = (end - start) * lambda;
  const float reg = (float)rowlen * a.lambda;
  if constexpr (MODE == kModeMaterialize) {
    float* tt = a.tt + (size_t)(row - a.row_begin) * f * f;
    float* rhs = a.rhs ? a.rhs + (size_t)(row - a.row_begin) * f : nullptr;
    switch (wave) {
      case 0: tiles_to_global<NB, 0>(acc, tt, rhs, f, reg, lane); break;
      case 1: tiles_to_global<NB, 1>(acc, tt, rhs, f, reg, lane); break;
      case 2: tiles_to_global<NB, 2>(acc, tt, rhs, f, reg, lane); break;
      default: tiles_to_global<NB, 3>(acc, tt, rhs, f, reg, lane); break;
    }
  } else {
    const int ldg = f + 1;
    float* G = smem;                         // aliases the stage buffers (all MFMA reads are done)
    float* vec = smem + solve_g_floats(f);   // CG vectors behind G
    switch (wave) {
      case 0: tiles_to_lds<NB, 0>(acc, G, ldg, f, lane); break;
      case 1: tiles_to_lds<NB, 1>(acc, G, ldg, f, lane); break;
      case 2: tiles_to_lds<NB, 2>(acc, G, ldg, f, lane); break;
      default: tiles_to_lds<NB, 3>(acc, G, ldg, f, lane); break;
    }
    __syncthreads();
    if (tid < f) G[tid * ldg + tid] += reg;
    __syncthreads();
    float* x = a.update + (size_t)row * f;
    if constexpr (MODE == kModeCG)
      cg_solve_lds(G, ldg, f, vec, x, a.cg_iters, tid);
    else
      lu_solve_lds(G, ldg, f, x, tid);
  }
}

// ----------------------------------------------------------------------------------
// Kernel 1: one workgroup per plan item (a whole row, or one chunk of a heavy row).
// ----------------------------------------------------------------------------------
template <int NB, typename VT, int MODE>
__global__ __launch_bounds__(kThreads) void als_item_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = Geo<NB>::LD, TPW = Geo<NB>::TPW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x;
  const int row = a.item_row[item];
  const long long begin = a.item_begin[item];
  const int len = a.item_len[item];
  const int slot = a.item_slot[item];

  f32x4 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int kStageFloats = kStage * LD;
  const int nstages = (len + kStage - 1) / kStage;
  Stager<NB, VT> st;
  if (nstages > 0) {
    const int nv = len < kStage ? len : kStage;
    st.load(a.colidx, a.val, a.gather, a.f, begin, nv, tid);
    st.store(smem, (nv + 3) & ~3, tid);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const int nv = (len - s * kStage) < kStage ? (len - s * kStage) : kStage;
    const bool more = (s + 1 < nstages);
    int nv_next = 0;
    if (more) {
      nv_next = (len - (s + 1) * kStage) < kStage ? (len - (s + 1) * kStage) : kStage;
      st.load(a.colidx, a.val, a.gather, a.f, begin + (long long)(s + 1) * kStage, nv_next, tid);
    }
    const float* cur = smem + (s & 1) * kStageFloats;
    const int ngroups = (nv + 3) >> 2;
    switch (wave) {
      case 0: mma_stage<NB, 0>(cur, acc, ngroups, lane); break;
      case 1: mma_stage<NB, 1>(cur, acc, ngroups, lane); break;
      case 2: mma_stage<NB, 2>(cur, acc, ngroups, lane); break;
      default: mma_stage<NB, 3>(cur, acc, ngroups, lane); break;
    }
    if (more) st.store(smem + ((s + 1) & 1) * kStageFloats, (nv_next + 3) & ~3, tid);
    __syncthreads();
  }

  if (slot >= 0) {
    float* part = a.part + (size_t)slot * Geo<NB>::NT * 256;
    switch (wave) {
      case 0: tiles_to_partial<NB, 0>(acc, part, lane); break;
      case 1: tiles_to_partial<NB, 1>(acc, part, lane); break;
      case 2: tiles_to_partial<NB, 2>(acc, part, lane); break;
      default: tiles_to_partial<NB, 3>(acc, part, lane); break;
    }
    return;
  }
  finish_row<NB, MODE>(acc, smem, a, row, a.item_rowlen[item], tid);
}

// ----------------------------------------------------------------------------------
// Kernel 2: one workgroup per chunked row: sum the partial tiles in slot order (a
// fixed, deterministic order) and finish the row.
// ----------------------------------------------------------------------------------
template <int NB, int MODE>
__global__ __launch_bounds__(kThreads) void als_reduce_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TPW = Geo<NB>::TPW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int mr = blockIdx.x;
  const int row = a.mrow_row[mr];
  const int slot0 = a.mrow_slot0[mr];
  const int nslots = a.mrow_nslots[mr];

  f32x4 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < nslots; ++sl) {
    const float* part = a.part + (size_t)(slot0 + sl) * Geo<NB>::NT * 256;
    switch (wave) {
      case 0: partial_accumulate<NB, 0>(acc, part, lane); break;
      case 1: partial_accumulate<NB, 1>(acc, part, lane); break;
      case 2: partial_accumulate<NB, 2>(acc, part, lane); break;
      default: partial_accumulate<NB, 3>(acc, part, lane); break;
    }
  }
  finish_row<NB, MODE>(acc, smem, a, row, a.mrow_rowlen[mr], tid);
}

// ----------------------------------------------------------------------------------
// Standalone batched solvers on materialised systems (the reference's data flow).
// ----------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kThreads) void solve_lds_kernel(const float* __restrict__ A, const float* __restrict__ b,
                                                             float* __restrict__ x, int f, int cg_iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const size_t sys = blockIdx.x;
  const int ldg = f + 1;
  float* G = smem;
  const float* As = A + sys * (size_t)f * f;
  for (int e = tid; e < f * f; e += kThreads) {
    const int i = e / f, j = e - i * f;
    G[i * ldg + j] = As[e];
  }
  if (tid < f) G[tid * ldg + f] = b[sys * f + tid];
  __syncthreads();
  if constexpr (MODE == kModeCG)
    cg_solve_lds(G, ldg, f, smem + solve_g_floats(f), x + sys * f, cg_iters, tid);
  else
    lu_solve_lds(G, ldg, f, x + sys * f, tid);
}

// CG with A streamed from global memory every mat-vec, for f too large for an
// LDS-resident system (f > 128).  One workgroup per system, thread t owns row t
// (blockDim = f rounded up to 64; same shape as cg.cu:36-231, wave64 reductions).
__global__ void cg_global_kernel(const float* __restrict__ A, float* __restrict__ x, const float* __restrict__ b,
                                 int f, int cg_iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  float* ps = smem;             // f
  float* red = smem + blockDim.x;  // nwaves
  const float* As = A + (size_t)blockIdx.x * f * f;
  float* xs = x + (size_t)blockIdx.x * f;
  const bool own = tid < f;

  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < nwaves; ++w) s += red[w];
    return s;
  };
  auto matvec = [&]() {
    float s = 0.f;
    if (own)
      for (int j = 0; j < f; ++j) s = fmaf(As[(size_t)j * f + tid], ps[j], s);
    return s;
  };

  float xv = own ? xs[tid] : 0.f;
  if (own) ps[tid] = xv;
  __syncthreads();
  float r = own ? (b[(size_t)blockIdx.x * f + tid] - matvec()) : 0.f;
  __syncthreads();
  float p = r;
  if (own) ps[tid] = p;
  float rsold = block_sum(r * r);  // its barriers also publish ps
  for (int iter = 0; iter < cg_iters; ++iter) {
    const float ap = matvec();
    const float pap = block_sum(own ? p * ap : 0.f);
    const float alpha = rsold / pap;
    xv = fmaf(alpha, p, xv);
    r = fmaf(-alpha, ap, r);
    const float rsnew = block_sum(own ? r * r : 0.f);
    if ((double)rsnew < 1e-4) break;
    const float beta = rsnew / rsold;
    rsold = rsnew;
    p = fmaf(beta, p, r);
    __syncthreads();
    if (own) ps[tid] = p;
    __syncthreads();
  }
  if (own) xs[tid] = xv;
}

// ----------------------------------------------------------------------------------
// Sum of squared errors (RMSE kernel + Sasum, als.cu:191-219, 979-991): 16 lanes per
// rating, 8/16-byte gathers of both factor rows, fp64 accumulation across ratings.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void sse_kernel(const float* __restrict__ val, const int* __restrict__ row,
                                                       const int* __restrict__ col, const float* __restrict__ thetaT,
                                                       const float* __restrict__ XT, long long count, int f,
                                                       int surpass_nan, double* __restrict__ out) {
  __shared__ double red[kThreads / 64];
  const int tid = threadIdx.x, sub = tid & 15;
  const long long per_block = kThreads / 16;
  double local = 0.0;
  for (long long base = (long long)blockIdx.x * per_block; base < count; base += (long long)gridDim.x * per_block) {
    const long long i = base + (tid >> 4);
    float e = 0.f;
    if (i < count) {
      const float* th = thetaT + (size_t)col[i] * f;
      const float* xr = XT + (size_t)row[i] * f;
      float s = 0.f;
      int first_nan = f;
      if (surpass_nan) {  // SURPASS_NAN (als.cu:201-211): stop at the first NaN factor entry
        for (int k = sub * 2; k < f; k += 32) {
          const f32x2 a = *reinterpret_cast<const f32x2*>(th + k);
          const f32x2 b = *reinterpret_cast<const f32x2*>(xr + k);
          if ((a[0] != a[0] || b[0] != b[0]) && k < first_nan) first_nan = k;
          if ((a[1] != a[1] || b[1] != b[1]) && k + 1 < first_nan) first_nan = k + 1;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          const int other = __shfl_xor(first_nan, o);
          first_nan = other < first_nan ? other : first_nan;
        }
      }
      for (int k = sub * 2; k < f; k += 32) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(th + k);
        const f32x2 b = *reinterpret_cast<const f32x2*>(xr + k);
        if (k < first_nan) s = fmaf(a[0], b[0], s);
        if (k + 1 < first_nan) s = fmaf(a[1], b[1], s);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
      e = val[i] - s;
    }
    if (sub == 0 && i < count) local += (double)e * (double)e;
  }
  // block reduction in fp64
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((tid & 63) == 0) red[tid >> 6] = local;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
    atomicAdd(out, s);
  }
}

// ----------------------------------------------------------------------------------
// Launchers
// ----------------------------------------------------------------------------------

// Optional per-kernel HIP-event timing of the last half-iteration (bench.py's roofline
// leg): events are recorded on the SAME stream the kernels are launched on.
static bool g_timing = false;
static hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
static bool g_timed_item = false, g_timed_reduce = false;

void set_kernel_timing(bool on) {
  g_timing = on;
  if (on && g_ev[0] == nullptr)
    for (auto& e : g_ev) (void)hipEventCreate(&e);
}
hipError_t last_kernel_ms(float* item_ms, float* reduce_ms) {
  *item_ms = 0.f;
  *reduce_ms = 0.f;
  if (!g_ev[0]) return hipSuccess;
  hipError_t e = hipEventSynchronize(g_ev[2]);
  if (e != hipSuccess) return e;
  if (g_timed_item) (void)hipEventElapsedTime(item_ms, g_ev[0], g_ev[1]);
  if (g_timed_reduce) (void)hipEventElapsedTime(reduce_ms, g_ev[1], g_ev[2]);
  return hipSuccess;
}
template <int NB, typename VT, int MODE>
static hipError_t launch_nb(const KernelArgs& a, long n_items, long n_mrows, hipStream_t stream) {
  const size_t stage_floats = 2 * (size_t)kStage * Geo<NB>::LD;
  size_t floats = stage_floats;
  if (MODE != kModeMaterialize) {
    const size_t solve = solve_g_floats(a.f) + (MODE == kModeCG ? 6 * kVecLd : 0);
    floats = floats > solve ? floats : solve;
  }
  const size_t lds = floats * sizeof(float);
  hipError_t e;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_item_kernel<NB, VT, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_reduce_kernel<NB, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (g_timing) (void)hipEventRecord(g_ev[0], stream);
  g_timed_item = n_items > 0;
  g_timed_reduce = n_mrows > 0;
  if (n_items > 0) {
    hipLaunchKernelGGL((als_item_kernel<NB, VT, MODE>), dim3((unsigned)n_items), dim3(kThreads), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (g_timing) (void)hipEventRecord(g_ev[1], stream);
  if (n_mrows > 0) {
    const size_t lds2 = (MODE == kModeMaterialize) ? 0 : lds;
    hipLaunchKernelGGL((als_reduce_kernel<NB, MODE>), dim3((unsigned)n_mrows), dim3(kThreads), lds2, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (g_timing) (void)hipEventRecord(g_ev[2], stream);
  return hipSuccess;
}

template <int NB>
static hipError_t launch_mode(const KernelArgs& a, int mode, long n_items, long n_mrows, hipStream_t stream) {
  const bool v4 = (a.f % 4 == 0);
  if (mode == kModeMaterialize)
    return v4 ? launch_nb<NB, f32x4, kModeMaterialize>(a, n_items, n_mrows, stream)
              : launch_nb<NB, f32x2, kModeMaterialize>(a, n_items, n_mrows, stream);
  if constexpr (NB <= kMaxFusedNB) {
    if (mode == kModeCG)
      return v4 ? launch_nb<NB, f32x4, kModeCG>(a, n_items, n_mrows, stream)
                : launch_nb<NB, f32x2, kModeCG>(a, n_items, n_mrows, stream);
    return v4 ? launch_nb<NB, f32x4, kModeLU>(a, n_items, n_mrows, stream)
              : launch_nb<NB, f32x2, kModeLU>(a, n_items, n_mrows, stream);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_half_iteration(const KernelArgs& a, int mode, long n_items, long n_mrows, hipStream_t stream) {
  const int nb = nb_for_f(a.f);
  switch (nb) {
    case 1: return launch_mode<1>(a, mode, n_items, n_mrows, stream);
    case 2: return launch_mode<2>(a, mode, n_items, n_mrows, stream);
    case 3: return launch_mode<3>(a, mode, n_items, n_mrows, stream);
    case 4: return launch_mode<4>(a, mode, n_items, n_mrows, stream);
    case 5: return launch_mode<5>(a, mode, n_items, n_mrows, stream);
    case 6: return launch_mode<6>(a, mode, n_items, n_mrows, stream);
    case 7: return launch_mode<7>(a, mode, n_items, n_mrows, stream);
    case 8: return launch_mode<8>(a, mode, n_items, n_mrows, stream);
    case 9: return launch_mode<9>(a, mode, n_items, n_mrows, stream);
    case 10: return launch_mode<10>(a, mode, n_items, n_mrows, stream);
    case 11: return launch_mode<11>(a, mode, n_items, n_mrows, stream);
    case 12: return launch_mode<12>(a, mode, n_items, n_mrows, stream);
    case 13: return launch_mode<13>(a, mode, n_items, n_mrows, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_solve_batched(const float* A, const float* b, float* x, long batch, int f, int mode, int cg_iters,
                                hipStream_t stream) {
  if (batch <= 0) return hipSuccess;
  if (mode == kModeCG && f > 128) {
    const int threads = ((f + 63) / 64) * 64;
    const size_t lds = (threads + 16) * sizeof(float);
    hipLaunchKernelGGL(cg_global_kernel, dim3((unsigned)batch), dim3(threads), lds, stream, A, x, b, f, cg_iters);
    return hipGetLastError();
  }
  const size_t lds = (solve_g_floats(f) + (mode == kModeCG ? 6 * kVecLd : 0)) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e;
  if (mode == kModeCG) {
    if (lds > 64 * 1024) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(solve_lds_kernel<kModeCG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(solve_lds_kernel<kModeCG>, dim3((unsigned)batch), dim3(kThreads), lds, stream, A, b, x, f,
                       cg_iters);
  } else {
    if (lds > 64 * 1024) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(solve_lds_kernel<kModeLU>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(solve_lds_kernel<kModeLU>, dim3((unsigned)batch), dim3(kThreads), lds, stream, A, b, x, f,
                       cg_iters);
  }
  return hipGetLastError();
}

hipError_t launch_sse(const float* val, const int* row, const int* col, const float* thetaT, const float* XT,
                      long count, int f, int surpass_nan, double* out, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double), stream);
  if (e != hipSuccess) return e;
  if (count <= 0) return hipSuccess;
  long blocks = (count + 15) / 16;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(sse_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, val, row, col, thetaT, XT,
                     (long long)count, f, surpass_nan, out);
  return hipGetLastError();
}

}  // namespace cumf
