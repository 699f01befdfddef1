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

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "als_internal.h"
#include "als_device.h"
#include "als_lu_wg.h"

namespace cumf {


// The file is compiled once per NB (Makefile: -DCUMF_NB_SLICE=1..13, the kernels of that
// feature-block count) plus once with -DCUMF_NB_SLICE=0 (dispatch, NB-independent kernels,
// shared state), so that the build runs in parallel; without the macro everything lands in
// one translation unit.
#ifndef CUMF_ONLY_NB
#define CUMF_ONLY_NB 0  // experiments (tools/lu_variants.sh): build the kernels of one NB only
#endif
#if !defined(CUMF_NB_SLICE)
#define CUMF_SLICE_COMMON 1
#define CUMF_SLICE_HAS(n) (CUMF_ONLY_NB == 0 || CUMF_ONLY_NB == (n))
#elif CUMF_NB_SLICE == 0
#define CUMF_SLICE_COMMON 1
#define CUMF_SLICE_HAS(n) 0
#else
#define CUMF_SLICE_COMMON 0
#define CUMF_SLICE_HAS(n) (CUMF_NB_SLICE == (n))
#endif


#ifndef CUMF_LU_MFMA
#define CUMF_LU_MFMA 1  // 0: fused LU through the LDS hand-over + lu_solve_reg (the previous path)
#endif
// The accumulator LU pays off from f = 96 on (measured: f = 64 18.8 vs 18.0 ms, f = 10 0.67 vs 0.56 ms with
// the thread-grid LU; f = 100 35.7 vs 36.8, f = 128 62.2 vs 63.8, f = 200 200 vs 224).
constexpr bool lu_on_accumulators(int nb) { return CUMF_LU_MFMA != 0 && nb >= 7; }
#ifndef CUMF_RR_TILES
#define CUMF_RR_TILES 1
#endif
constexpr bool kRoundRobinTiles = CUMF_RR_TILES != 0;
static_assert(kRoundRobinTiles || !CUMF_LU_MFMA, "lu_solve_mfma (als_lu_wg.h) assumes round-robin tiles");



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
  // Tile held in accumulator slot s of wave role W.  Round-robin: every role keeps a similar
  // share of live tiles all through the elimination of lu_solve_mfma (with contiguous ranges the
  // last role owns the tiles that stay live to the end); the price is that every role reads all
  // NB feature blocks in the Gram pass.
  __host__ __device__ static constexpr int tile(int W, int s) {
    return (kRoundRobinTiles && NB >= 7) ? W + 4 * s : W * TPW + s;
  }
};

// ----------------------------------------------------------------------------------
// Global -> register -> LDS staging of kStage gathered factor rows.
// VT is float4 when f % 4 == 0 (16-byte loads; f = 100: 25 loads per row) else float2
// (f % 10 == 0 guarantees f even, main.cpp:33).
// ----------------------------------------------------------------------------------
template <int NB, typename VT>
struct Stager {
  static constexpr int VW = sizeof(VT) / 4;
  static constexpr int LD = Geo<NB>::LD;
  static constexpr int PPR = LD / VW;                 // vector pieces per stage row
  static constexpr int LPR = PPR <= 32 ? 32 : (PPR <= 64 ? 64 : 128);  // lanes covering one row (power of two)
  static constexpr int RPP = kThreads / LPR;          // rows per pass
  static constexpr int PASSES = kStage / RPP;
  static_assert(PPR <= 128, "stage row too wide");
  VT v[PASSES];        // gathered factor-row pieces of one stage
  float rvv;           // rating of row (tid & 31) of that stage
  int cols[PASSES];    // column indices feeding the next gather
  int cols_nx[PASSES]; // column indices one stage further ahead
  // loop-invariant per-thread state
  unsigned goff;       // byte offset of this lane's piece inside a factor row (clamped)
  int lds_row0;        // float offset of (row rsub, this piece) inside a stage buffer
  bool feat;           // this lane's piece holds features (col0 < f)
  int rsub, col0;

  // The steady-state stage loop must stay ONE basic block with as few VALU instructions as
  // possible: measured with tools/probes/mfma_ladder.hip, every VALU instruction issued next to the
  // MFMAs costs matrix-pipe time, and a load under a branch degrades every s_waitcnt to
  // vmcnt(0).  So: full stages take a select-free path (feature lanes store what they loaded,
  // the zero padding of the stage rows is written once per item, the rating goes through its
  // own 4-byte store), addresses are 32-bit offsets from wave-uniform bases (factor tables
  // are < 4 GiB), and only the last -- possibly ragged -- stage of an item uses masked stores.
  __device__ __forceinline__ void init(int f, int tid) {
    const int pc = tid % LPR;
    rsub = tid / LPR;
    col0 = pc * VW;
    feat = col0 < f;
    goff = feat ? (unsigned)col0 * 4u : 0u;
    lds_row0 = rsub * LD + col0;
  }

  // Zero the padding of both stage buffers: columns [f + VW, LD) of every row (the piece at
  // column f carries the rating and is rewritten whole per stage).  Needed once per item (the
  // solvers' G aliases the buffers).
  __device__ __forceinline__ void zero_padding(float* __restrict__ smem, int f, int tid) const {
    const int pieces = (LD - f) / VW - 1;  // per row
    for (int e = tid; e < 2 * kStage * pieces; e += kThreads) {
      const int row = e / pieces, k = e - row * pieces;
      VT z = {};
      *reinterpret_cast<VT*>(smem + row * LD + f + (k + 1) * VW) = z;
    }
  }

  template <bool FULL>
  __device__ __forceinline__ void load_cols_into(int (&dst)[PASSES], const int* __restrict__ colidx, long long begin,
                                                 int nvalid) {
    const int* base = colidx + begin;  // wave-uniform
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int r = rsub + p * RPP;
      dst[p] = base[(unsigned)(FULL ? r : (r < nvalid ? r : nvalid - 1))];
    }
  }

  // Gather of one stage: pass p loads VW consecutive features of factor row cols[p]; the
  // rating of row (tid & 31) rides along.  Nothing here consumes a loaded value.
  template <int P>
  __device__ __forceinline__ void gather_pass(const float* __restrict__ gat, unsigned f4) {
    const unsigned off = (unsigned)cols[P] * f4 + goff;  // bytes, < 4 GiB
    v[P] = *reinterpret_cast<const VT*>(reinterpret_cast<const char*>(gat) + off);
  }
  template <bool FULL>
  __device__ __forceinline__ void gather_val(const float* __restrict__ val, long long begin, int nvalid, int tid) {
    const int r = tid & (kStage - 1);
    if (val == nullptr) {  // no ratings given (alsUpdateFeature100Host: the right-hand side comes precomputed): zeros
      rvv = 0.f;
      return;
    }
    const float* vbase = val + begin;  // wave-uniform
    rvv = vbase[(unsigned)(FULL ? r : (r < nvalid ? r : nvalid - 1))];
  }
  template <bool FULL>
  __device__ __forceinline__ void gather(const float* __restrict__ val, const float* __restrict__ gat, unsigned f4,
                                         long long begin, int nvalid, int tid) {
    static_for<PASSES>([&](auto pc) { gather_pass<decltype(pc)::value>(gat, f4); });
    gather_val<FULL>(val, begin, nvalid, tid);
  }

  // Full stage: feature lanes store their piece as loaded; other lanes hit the dummy slot.
  template <int P>
  __device__ __forceinline__ void store_pass_full(float* __restrict__ stage, float* __restrict__ dummy) const {
    float* dst = feat ? stage + lds_row0 + P * RPP * LD : dummy;
    *reinterpret_cast<VT*>(dst) = v[P];
  }
  __device__ __forceinline__ void store_val_full(float* __restrict__ stage, int f, int tid) const {
    VT x = {};
    x[0] = rvv;  // the whole piece {rating, 0, ...}; 8 threads per row write the same value
    *reinterpret_cast<VT*>(stage + (tid & (kStage - 1)) * LD + f) = x;
  }
  // Ragged stage: rows [nvalid, nwrite) are written as zeros (nwrite = nvalid rounded up to 4).
  template <int P>
  __device__ __forceinline__ void store_pass_masked(float* __restrict__ stage, float* __restrict__ dummy, int nvalid,
                                                    int nwrite) const {
    const int r = rsub + P * RPP;
    VT x = v[P];
#pragma unroll
    for (int e = 0; e < VW; ++e) x[e] = (r < nvalid) ? x[e] : 0.f;
    float* dst = (feat && r < nwrite) ? stage + lds_row0 + P * RPP * LD : dummy;
    *reinterpret_cast<VT*>(dst) = x;
  }
  __device__ __forceinline__ void store_val_masked(float* __restrict__ stage, float* __restrict__ dummy, int f,
                                                   int nvalid, int nwrite, int tid) const {
    const int r = tid & (kStage - 1);
    float* dst = (r < nwrite) ? stage + r * LD + f : dummy;
    VT x = {};
    x[0] = (r < nvalid) ? rvv : 0.f;
    *reinterpret_cast<VT*>(dst) = x;
  }
  __device__ __forceinline__ void store_masked(float* __restrict__ stage, float* __restrict__ dummy, int f, int nvalid,
                                               int nwrite, int tid) const {
    static_for<PASSES>([&](auto pc) { store_pass_masked<decltype(pc)::value>(stage, dummy, nvalid, nwrite); });
    store_val_masked(stage, dummy, f, nvalid, nwrite, tid);
  }
  __device__ __forceinline__ void rotate_cols() {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) cols[p] = cols_nx[p];
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
__device__ __forceinline__ void mma_group(const float (&blk)[NB], f32x4 (&acc)[Geo<NB>::TPW]) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = Geo<NB>::tile(W, s);
    if constexpr (t < NT) {
      constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
      acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(blk[I], blk[J], acc[s], 0, 0, 0);
    }
  });
}

template <int NB, int W>
__device__ __forceinline__ void mma_stage(const float* __restrict__ stage, f32x4 (&acc)[Geo<NB>::TPW],
                                          int ngroups, int lane) {
  constexpr int LD = Geo<NB>::LD;
  const float* rowp = stage + (lane >> 4) * LD + (lane & 15);
  // Software-pipelined by hand: the operand reads of group g+1 are issued before the MFMAs
  // of group g so that LDS latency hides behind the matrix pipe (the scheduler sinks the
  // reads next to their use otherwise, hence the sched_barriers).  Blocks this wave never
  // uses are dead code.  The read-ahead may run up to two groups past `ngroups`: it stays
  // inside the LDS allocation (launch_nb pads it) and the values are never used.
  auto load_blk = [&](float (&blk)[NB], const float* p) {
#pragma unroll
    for (int b = 0; b < NB; ++b) blk[b] = p[16 * b];
  };
  float blk_a[NB], blk_b[NB];
  load_blk(blk_a, rowp);
  int g = 0;
  for (; g + 1 < ngroups; g += 2) {
    load_blk(blk_b, rowp + 4 * LD);
    __builtin_amdgcn_sched_barrier(0);
    mma_group<NB, W>(blk_a, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_blk(blk_a, rowp + 8 * LD);
    __builtin_amdgcn_sched_barrier(0);
    mma_group<NB, W>(blk_b, acc);
    __builtin_amdgcn_sched_barrier(0);
    rowp += 8 * LD;
  }
  if (g < ngroups) mma_group<NB, W>(blk_a, acc);
}

// Accumulator tiles -> LDS, tile-major ([tile][16][16], rows permuted by tiled_row): the
// hand-over to lu_solve_reg, whose threads pick their elements up with TileLoad.
template <int NB, int W>
__device__ __forceinline__ void tiles_to_tiled(const f32x4 (&acc)[Geo<NB>::TPW], float* __restrict__ T,
                                               float reg, int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  const int c = lane & 15, kk = lane >> 4;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = Geo<NB>::tile(W, s);
    if constexpr (t < NT) {
      constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[s][r];
        if (I == J && 4 * kk + r == c) v += reg;  // lambda * n_u on the diagonal (als.cu:545-557)
        T[256 * t + (4 * r + kk) * 16 + c] = v;
      }
    }
  });
}

// Accumulator tile -> LDS system matrix G (f x ldg, column f = RHS).  C/D layout of
// the 16x16 MFMA: lane l, register r holds D[4*(l>>4) + r][l & 15].
template <int NB, int W>
__device__ __forceinline__ void tiles_to_lds(const f32x4 (&acc)[Geo<NB>::TPW], float* __restrict__ G,
                                             int ldg, int f, float reg, int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  const int c = lane & 15, kk = lane >> 4;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = Geo<NB>::tile(W, s);
    if constexpr (t < NT) {
      constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * kk + r, j = 16 * J + c;
        float v = acc[s][r];
        if (I == J && i == j) v += reg;                    // lambda * n_u on the diagonal (als.cu:545-557)
        if (i < f && j <= f) G[i * ldg + j] = v;           // j == f: b_i = sum r * theta[i]
        if (I != J && i < f && j < f) G[j * ldg + i] = v;  // mirror
      }
    }
  });
}

// Accumulator tile -> row-major f x f Gram in global memory (both triangles,
// lambda * n on the diagonal: als.cu:545-566) + RHS.
template <int NB, int W, typename T>
__device__ __forceinline__ void tiles_to_global(const f32x4 (&acc)[Geo<NB>::TPW], T* __restrict__ tt,
                                                float* __restrict__ rhs, int f, float reg, int lane,
                                                bool packed = false) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  const int c = lane & 15, kk = lane >> 4;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = Geo<NB>::tile(W, s);
    if constexpr (t < NT) {
      constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * kk + r, j = 16 * J + c;
        float v = acc[s][r];
        if (i < f && j < f) {
          if (i == j) v += reg;
          // both triangles from one accumulator entry (tiles summed from the split path are not bit-symmetric
          // inside a diagonal tile; als.h:39-143 mirrors one temp as well)
          if (I != J || i <= j) {
            if (packed) {  // row i keeps columns i .. f - 1 (cumf_get_hermitian_packed)
              tt[(size_t)i * f - (size_t)(i * (i - 1) / 2) + (j - i)] = (T)v;
            } else {
              tt[(size_t)i * f + j] = (T)v;  // T = _Float16: round to nearest even, as __float2half_rn (als.h:373-499)
              if (i != j) tt[(size_t)j * f + i] = (T)v;
            }
          }
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
    constexpr int t = Geo<NB>::tile(W, s);
    if constexpr (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[((size_t)t * 4 + r) * 64 + lane] = acc[s][r];
    }
  });
}
// NEG: the sum of the partial tiles NEGATED (what the blocked workgroup LU eliminates: lu_solve_blocked_wg)
template <int NB, int W, bool NEG = false>
__device__ __forceinline__ void partial_accumulate(f32x4 (&acc)[Geo<NB>::TPW], const float* __restrict__ part,
                                                   int lane) {
  constexpr int NT = Geo<NB>::NT, TPW = Geo<NB>::TPW;
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = Geo<NB>::tile(W, s);
    if constexpr (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = part[((size_t)t * 4 + r) * 64 + lane];
        acc[s][r] = NEG ? acc[s][r] - v : acc[s][r] + v;
      }
    }
  });
}

// ----------------------------------------------------------------------------------
// In-LDS solvers.  G is f x ldg (ldg = solve_ldg(f): f + 1 rounded up to 4, so rows
// are 16-byte aligned), column f holds b.  256 threads.
// ----------------------------------------------------------------------------------

// Conjugate gradient exactly as cg.cu:36-231: warm start, r = b - A x, <= cg_iters
// iterations, stop when ||r||^2 < 1e-4 (CG_ERROR, cg.cu:31,195; the float is compared
// against the double literal).
//
// Layout: every wave keeps ALL four vectors (x, r, p, ap) in registers, element i in lane
// i & 63, slot i >> 6, and performs the vector updates and the dot products redundantly;
// identical instruction sequences on identical data give identical bits in the four
// waves, so alpha / beta / the exit test are workgroup-uniform without communication.
// Only the mat-vec is shared: wave w multiplies rows [w*JW, (w+1)*JW) of the symmetric G
// (16-byte LDS reads, both half-waves on different rows), the four partial vectors go
// through LDS and ONE barrier per iteration.  Dot products are fixed-order DPP
// reductions in place of the reference's order-dependent smem atomics
// (device_utilities.h:36-48).  Requires f <= 128.
template <int NB>
__device__ __forceinline__ void cg_solve_lds(const float* __restrict__ G, int ldg, int f,
                                             float* __restrict__ vec, float* __restrict__ x_global,
                                             int cg_iters, int tid) {
  constexpr int MAXIT = 2 * NB;  // rows per half-wave: ceil(ceil(16*NB / 4) / 2)
  const int wave = tid >> 6, lane = tid & 63;
  const int c = lane & 31, h = lane >> 5;
  float* pw = vec + wave * kVecLd;      // this wave's private copy of the mat-vec operand
  float* part = vec + 4 * kVecLd;       // [2][4][kVecLd] partial mat-vecs, double-buffered
  const int jw = (f + 3) >> 2;          // rows of G per wave
  const int jbeg = wave * jw;
  const int jend = (jbeg + jw) < f ? (jbeg + jw) : f;
  const bool colok = 4 * c < ldg;
  const int i0 = lane, i1 = lane + 64;
  const bool ok0 = i0 < f, ok1 = i1 < f;

  int buf = 0;
  // y = G * v for the vector held as (v0, v1); result replicated in every wave
  auto matvec = [&](float v0, float v1, float& y0, float& y1) {
    pw[i0] = v0;
    pw[i1] = v1;
    __builtin_amdgcn_wave_barrier();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int j = jbeg + h + 2 * it;
      const bool on = (j < jend) && colok;
      const int jc = on ? j : 0;
      const f32x4 g = *reinterpret_cast<const f32x4*>(G + jc * ldg + (on ? 4 * c : 0));
      const float pj = on ? pw[jc] : 0.f;
      acc[0] = fmaf(g[0], pj, acc[0]);
      acc[1] = fmaf(g[1], pj, acc[1]);
      acc[2] = fmaf(g[2], pj, acc[2]);
      acc[3] = fmaf(g[3], pj, acc[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], 32);
    float* pb = part + (buf * 4 + wave) * kVecLd;
    if (h == 0) *reinterpret_cast<f32x4*>(pb + 4 * c) = acc;
    __syncthreads();
    const float* pr = part + buf * 4 * kVecLd;
    y0 = ((pr[i0] + pr[kVecLd + i0]) + pr[2 * kVecLd + i0]) + pr[3 * kVecLd + i0];
    y1 = ((pr[i1] + pr[kVecLd + i1]) + pr[2 * kVecLd + i1]) + pr[3 * kVecLd + i1];
    buf ^= 1;
  };

  float x0 = ok0 ? x_global[i0] : 0.f, x1 = ok1 ? x_global[i1] : 0.f;
  float ax0, ax1;
  matvec(x0, x1, ax0, ax1);
  float r0 = ok0 ? G[i0 * ldg + f] - ax0 : 0.f;
  float r1 = ok1 ? G[i1 * ldg + f] - ax1 : 0.f;
  float p0 = r0, p1 = r1;
  float rsold = wave_sum_uniform(fmaf(r1, r1, r0 * r0));
  for (int iter = 0; iter < cg_iters; ++iter) {
    float ap0, ap1;
    matvec(p0, p1, ap0, ap1);
    ap0 = ok0 ? ap0 : 0.f;
    ap1 = ok1 ? ap1 : 0.f;
    const float pap = wave_sum_uniform(fmaf(p1, ap1, p0 * ap0));
    const float alpha = rsold / pap;
    x0 = fmaf(alpha, p0, x0);
    x1 = fmaf(alpha, p1, x1);
    r0 = fmaf(-alpha, ap0, r0);
    r1 = fmaf(-alpha, ap1, r1);
    const float rsnew = wave_sum_uniform(fmaf(r1, r1, r0 * r0));
    if ((double)rsnew < 1e-4) break;
    const float beta = rsnew / rsold;
    rsold = rsnew;
    p0 = fmaf(beta, p0, r0);
    p1 = fmaf(beta, p1, r1);
  }
  if (wave == 0) {
    if (ok0) x_global[i0] = x0;
    if (ok1) x_global[i1] = x1;
  }
}

// Back substitution U x = y by one wave (lanes own rows i = lane + 64 q), column-oriented
// like BLAS strsv: x_k final, then every y_i (i < k) loses U_ik x_k.  rdiag (may be null)
// holds the reciprocals of the pivots; without it x_k = y_k / U_kk (IEEE division).
template <bool RECIP, int NQ>
__device__ __forceinline__ void back_substitute_lds(const float* __restrict__ G, int ldg, int f,
                                                    const float* __restrict__ rdiag,
                                                    float* __restrict__ x_global, int lane) {
  float y[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) y[q] = (lane + 64 * q < f) ? G[(lane + 64 * q) * ldg + f] : 0.f;
  // column k of U for this lane's rows, fetched one step ahead of its use
  float col[NQ], coln[NQ];
  float dk = 0.f, dkn = 0.f;
  const float* colp[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = lane + 64 * q;
    colp[q] = G + (i < f ? i : f - 1) * ldg;
  }
  auto fetch = [&](int k, float (&cv)[NQ], float& d) {
    const int kc = k < 0 ? 0 : k;
#pragma unroll
    for (int q = 0; q < NQ; ++q) cv[q] = colp[q][kc];
    d = RECIP ? rdiag[kc] : G[kc * ldg + kc];
  };
  fetch(f - 1, col, dk);
  for (int k = f - 1; k >= 0; --k) {
    fetch(k - 1, coln, dkn);
    const int kq = k >> 6, kl = k & 63;
    float yk = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q == kq) yk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y[q]), kl));
    const float xk = RECIP ? yk * dk : yk / dk;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = lane + 64 * q;
      const float upd = fmaf(-col[q], xk, y[q]);
      y[q] = (i == k) ? xk : ((i < k) ? upd : y[q]);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) col[q] = coln[q];
    dk = dkn;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + 64 * q < f) x_global[lane + 64 * q] = y[q];
}

// Unpivoted Gaussian elimination on the augmented system [A | b] followed by back
// substitution: the mathematical content of cublasSgetrfBatched(PivotArray = NULL) +
// cublasSgetrsBatched (als.cu:77,98), all in LDS.  Same operation order as oracle_lu
// (right-looking, IEEE division by the pivot, fmaf updates, descending back
// substitution), so the result is bit-identical to the oracle on identical A, b.  LDS
// bandwidth bound; kept for f > 128 (any ldg) and as the exact-order reference variant.
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
  if (tid < 64) back_substitute_lds<false, 4>(G, ldg, f, nullptr, x_global, tid);
}

// Register-resident symmetric elimination (the fast LU path, f <= 200).
// The upper triangle of [A | b] is spread over the 16 x 16 thread grid, element (i, j) in
// thread (i & 15, j & 15), register block (i >> 4, j >> 4); `load(bi, bj)` fetches this
// thread's element of block (bi, bj) (from the accumulator tiles parked in LDS, or from
// global memory).  Per pivot k:
//   1. the thread row owning row k publishes it (as it stands, i.e. updated by all earlier
//      pivots) to the packed row store U (lu_row_off); the thread holding u_kk computes
//      1 / u_kk meanwhile and publishes that too, so the reciprocal is off the readers'
//      critical path; ONE barrier;
//   2. every thread reads row k at its own row / column positions and applies
//      a_ij -= (u_ki / u_kk) * u_kj  to its registers (i > k).
// This is Gaussian elimination without pivoting restricted to the upper triangle (U = D L^T of
// A = L U).  Tried and measured slower or equal (tools/lu_variants.sh): panels of 2 or 4
// pivots per barrier with a redundant in-register panel elimination (half / quarter the
// barriers, same LDS reads: equal at M = 2, spills at M = 4), a rolled pivot loop (+5 %).
// U may alias the memory `load` reads from: all loads complete before the first publish.
template <int NB, typename Load>
__device__ __forceinline__ void lu_solve_reg(Load load, float* __restrict__ U, int f, float* __restrict__ rdiag,
                                             float* __restrict__ x_global, int tid) {
  const int ti = tid >> 4, tj = tid & 15;
  float* zpad = rdiag + ((f + 3) & ~3) + 32;  // 16 zeros for back_substitute_zeroed (same place as in lu_solve_mfma)
  if (tid < 16) zpad[tid] = 0.f;
  float a[NB][NB];
  static_for<NB>([&](auto bic) {
    constexpr int bi = decltype(bic)::value;
    static_for<NB>([&](auto bjc) {
      constexpr int bj = decltype(bjc)::value;
      if constexpr (bj >= bi) a[bi][bj] = load(bic, bjc, ti, tj);
    });
  });
  __syncthreads();
  // Rows >= f and columns > f of the register image are padding: never published, never read
  // back, so updates run on them unmasked (whatever lands there is dead).  Reads of padding
  // positions stay inside the row store and only feed dead registers.
  const bool last_col_ok = 16 * (NB - 1) + tj <= f;
  static_for<NB>([&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    constexpr int pitch = lu_row_pitch<NB>(kb);
    float* blk = U + lu_block_off<NB>(kb) - 16 * kb;  // element (16 kb, 0) of this block row
    for (int kk = 0; kk < 16; ++kk) {
      const int k = 16 * kb + kk;
      if (k >= f) break;
      float* urow = blk + kk * pitch;
      if (ti == kk) {
        float* w = urow + tj;
        static_for<NB>([&](auto bjc) {
          constexpr int bj = decltype(bjc)::value;
          // entries at or left of the diagonal inside the row's own block are dead for the
          // elimination: publish zeros there, the back substitution then needs no triangle mask
          const float v = (bj > kb || tj > kk) ? a[kb][bj] : 0.f;
          if constexpr (bj >= kb && bj < NB - 1) w[16 * bj] = v;
          if constexpr (bj >= kb && bj == NB - 1) {
            if (last_col_ok) w[16 * bj] = v;
          }
        });
        if (tj == kk) {
          const float piv = a[kb][kb];
          const float t = __builtin_amdgcn_rcpf(piv);
          rdiag[k] = fmaf(fmaf(-piv, t, 1.0f), t, t);  // one Newton step: 1/pivot to ~1 ulp
        }
      }
      __syncthreads();
      float ui[NB], uj[NB];
      const float nrp = -rdiag[k];
      static_for<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (b >= kb) {
          uj[b] = urow[16 * b + tj];
          ui[b] = urow[16 * b + ti];
        }
      });
      static_for<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (b > kb) ui[b] = ui[b] * nrp;
        if constexpr (b == kb) ui[b] = (ti > kk) ? ui[b] * nrp : 0.f;
      });
      static_for<NB>([&](auto bic) {
        constexpr int bi = decltype(bic)::value;
        static_for<NB>([&](auto bjc) {
          constexpr int bj = decltype(bjc)::value;
          if constexpr (bi >= kb && bj >= bi) a[bi][bj] = fmaf(ui[bi], uj[bj], a[bi][bj]);
        });
      });
    }
  });
  __syncthreads();
  if (tid < 64) back_substitute_zeroed<NB, (16 * NB + 63) / 64>(U, f, rdiag, zpad, x_global, tid);
}

// LDS floats of the fused LU of NB feature blocks: lu_solve_mfma (NB >= 7) or the thread-grid
// lu_solve_reg on the packed row store.
template <int NB>
__host__ __device__ constexpr size_t lu_fused_lds_floats(int f) {
  return lu_on_accumulators(NB) ? lu_wg_lds_floats<NB>(f) : lu_lds_floats(NB, f);
}

// Loaders of lu_solve_reg.  TileLoad: the accumulator tiles parked in LDS by tiles_to_tiled.
template <int NB>
struct TileLoad {
  const float* T;
  int f;
  template <typename BI, typename BJ>
  __device__ __forceinline__ float operator()(BI, BJ, int ti, int tj) const {
    constexpr int bi = BI::value, bj = BJ::value;
    const int i = 16 * bi + ti, j = 16 * bj + tj;
    const float v = T[256 * tile_of<NB>(bi, bj) + tiled_row(ti) * 16 + tj];
    return (i < f && j <= f) ? v : 0.f;
  }
};
// GlobalLoad: a row-major f x f matrix and its right-hand side in global memory.
template <int NB>
struct GlobalLoad {
  const float* A;
  const float* b;
  int f;
  template <typename BI, typename BJ>
  __device__ __forceinline__ float operator()(BI, BJ, int ti, int tj) const {
    constexpr int bi = BI::value, bj = BJ::value;
    const int i = 16 * bi + ti, j = 16 * bj + tj;
    const int ic = i < f ? i : f - 1;
    const float v = (j < f) ? A[(size_t)ic * f + j] : b[ic];
    return (i < f && j <= f) ? v : 0.f;
  }
};

// ----------------------------------------------------------------------------------
// Row epilogue.  dump_row<W> is per-wave (tile layout); solve_row is common to all waves.
// ----------------------------------------------------------------------------------
template <int NB, int MODE, int W>
__device__ __forceinline__ void dump_row(const f32x4 (&acc)[Geo<NB>::TPW], float* smem, const KernelArgs& a, int row,
                                         int rowlen, int lane) {
  const int f = a.f;
  if constexpr (MODE == kModeMaterialize) {
    // als.cu:547: float temp = (end - start) * lambda;
    const float reg = (float)rowlen * a.lambda;
    const size_t off = (size_t)(row - a.row_begin) * (a.tt_packed ? (size_t)f * (f + 1) / 2 : (size_t)f * f);
    float* rhs = a.rhs ? a.rhs + (size_t)(row - a.row_begin) * f : nullptr;
    if (a.tt_half)
      tiles_to_global<NB, W>(acc, reinterpret_cast<_Float16*>(a.tt) + off, rhs, f, reg, lane);
    else
      tiles_to_global<NB, W>(acc, a.tt + off, rhs, f, reg, lane, a.tt_packed != 0);
  } else {
    // G / the tile store aliases the stage buffers (all MFMA reads are done)
    if constexpr (MODE == kModeLU)
      tiles_to_tiled<NB, W>(acc, smem, (float)rowlen * a.lambda, lane);
    else
      tiles_to_lds<NB, W>(acc, smem, solve_ldg(f, MODE), f, (float)rowlen * a.lambda, lane);
  }
}

// LU (default build): the whole solve runs on the accumulators inside the wave roles.
template <int NB, int MODE, int W, bool NEG = false>
__device__ __forceinline__ void finish_row(f32x4 (&acc)[Geo<NB>::TPW], float* smem, const KernelArgs& a, int row,
                                           int rowlen, int tid) {
  if constexpr (MODE == kModeLU && lu_on_accumulators(NB)) {
    lu_solve_wg<NB, W, 4, NEG>(acc, smem, a.f, (float)rowlen * a.lambda, a.update + (size_t)row * a.f, tid, a.sse_bins, rowlen);
  } else {
    dump_row<NB, MODE, W>(acc, smem, a, row, rowlen, tid & 63);
  }
}

template <int NB, int MODE>
__device__ __forceinline__ void solve_row(float* smem, const KernelArgs& a, int row, int tid) {
  if constexpr (MODE == kModeLU && lu_on_accumulators(NB)) return;  // done in finish_row
  if constexpr (MODE != kModeMaterialize) {
    const int f = a.f, ldg = solve_ldg(f, MODE);
    float* G = smem;
    __syncthreads();  // all tiles are in G
    float* x = a.update + (size_t)row * f;
    if constexpr (MODE == kModeCG)
      cg_solve_lds<NB>(G, ldg, f, smem + solve_g_floats(f, MODE), x, a.cg_iters, tid);
    else
      lu_solve_reg<NB>(TileLoad<NB>{smem, f}, smem, f, smem + lu_packed_floats(NB), x, tid);
  }
}

// ----------------------------------------------------------------------------------
// Kernel 1: one workgroup per plan item (a whole row, or one chunk of a heavy row).
// The four waves run wave-specialised copies of the same loop (each owns a fixed set of
// tiles); every copy executes the same sequence of barriers.
// ----------------------------------------------------------------------------------
template <int NB, typename VT, int MODE, int W>
__device__ __forceinline__ void item_body(float* smem, const KernelArgs& a, int row, long long begin, int len,
                                          int slot, int rowlen, int tid) {
  constexpr int LD = Geo<NB>::LD, TPW = Geo<NB>::TPW;
  constexpr int kStageFloats = kStage * LD;
  constexpr int NG = kStage / 4;  // MFMA groups of 4 ratings per full stage
  using St = Stager<NB, VT>;
  const int lane = tid & 63;
  const int f = a.f;
  const unsigned f4 = (unsigned)f * 4u;
  f32x4 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nstages = (len + kStage - 1) / kStage;
  auto nvalid_of = [&](int s) { return (len - s * kStage) < kStage ? (len - s * kStage) : kStage; };
  auto begin_of = [&](int s) { return begin + (long long)s * kStage; };
  St st;
  st.init(f, tid);
  // landing slot for masked-off stage stores: inside the read-ahead pad behind the two stage
  // buffers (read by nobody's MFMAs; in fused modes it is overwritten by G only after the
  // last barrier of the loop)
  float* dummy = smem + 2 * kStageFloats + 4 * LD + (tid & 15) * 4;
  // Prologue: padding zeroed, stage 0 into LDS buffer 0, stage 1 gathers in flight, column
  // indices of stage 2.
  if (nstages > 0) {
    const int nv = nvalid_of(0);
    st.template load_cols_into<false>(st.cols, a.colidx, begin, nv);
    st.zero_padding(smem, f, tid);
    st.template gather<false>(a.val, a.gather, f4, begin, nv, tid);
    if (nstages > 1) st.template load_cols_into<false>(st.cols_nx, a.colidx, begin_of(1), nvalid_of(1));
    st.store_masked(smem, dummy, f, nv, (nv + 3) & ~3, tid);
    if (nstages > 1) {
      st.rotate_cols();
      st.template gather<false>(a.val, a.gather, f4, begin_of(1), nvalid_of(1), tid);
      if (nstages > 2) st.template load_cols_into<false>(st.cols, a.colidx, begin_of(2), nvalid_of(2));
    }
  }
  __syncthreads();

  // Steady state.  Every stage but the last is full (32 ratings = NG groups).  While the
  // MFMAs of group g drain through the matrix pipe the wave issues one slice of the staging
  // work: the LDS store of pass p of stage s+1 (gathered during stage s-1, so it has landed)
  // immediately followed by the gather of pass p of stage s+2 into the same registers; the
  // column indices of stage s+3 go out with the first slice.
  const float* rowbase = smem + (lane >> 4) * LD + (lane & 15);
  auto load_blk = [&](float (&blk)[NB], const float* p) {
#pragma unroll
    for (int b = 0; b < NB; ++b) blk[b] = p[16 * b];
  };
  // TARGET_FULL: stage s+1 (the one being stored) holds 32 ratings -> select-free stores.
  auto stage_body = [&](auto fullc, int s) {
    constexpr bool TARGET_FULL = decltype(fullc)::value;
    const float* cur = rowbase + (s & 1) * kStageFloats;
    float* nxt = smem + ((s + 1) & 1) * kStageFloats;
    const int nv1 = nvalid_of(s + 1), nw1 = (nv1 + 3) & ~3;
    // stages s+2 / s+3 may not exist near the end of the item: the loads are then issued
    // anyway on the last existing stage (in-bounds, never stored) to keep the loop branch-free
    const int s2 = (s + 2 < nstages) ? s + 2 : nstages - 1;
    const int s3 = (s + 3 < nstages) ? s + 3 : nstages - 1;
    const int nv2 = nvalid_of(s2), nv3 = nvalid_of(s3);
    const long long b2 = begin_of(s2), b3 = begin_of(s3);
    float blk_a[NB], blk_b[NB];
    load_blk(blk_a, cur);
    static_for<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      float (&bc)[NB] = (g & 1) ? blk_b : blk_a;
      float (&bn)[NB] = (g & 1) ? blk_a : blk_b;
      if constexpr (g + 1 < NG) load_blk(bn, cur + (g + 1) * 4 * LD);
      __builtin_amdgcn_sched_barrier(0);
      mma_group<NB, W>(bc, acc);
      __builtin_amdgcn_sched_barrier(0);
      static_for<St::PASSES>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p * NG / St::PASSES == g) {
          if constexpr (TARGET_FULL)
            st.template store_pass_full<p>(nxt, dummy);
          else
            st.template store_pass_masked<p>(nxt, dummy, nv1, nw1);
          st.template gather_pass<p>(a.gather, f4);
        }
      });
      if constexpr (g == NG - 1) {  // all passes stored: the rating register is free again
        if constexpr (TARGET_FULL)
          st.store_val_full(nxt, f, tid);
        else
          st.store_val_masked(nxt, dummy, f, nv1, nw1, tid);
        st.template gather_val<false>(a.val, b2, nv2, tid);
      }
      if constexpr (g == 0) st.template load_cols_into<false>(st.cols_nx, a.colidx, b3, nv3);  // used a stage later
      __builtin_amdgcn_sched_barrier(0);
    });
    st.rotate_cols();
    __syncthreads();
  };
  for (int s = 0; s + 2 < nstages; ++s) stage_body(std::true_type{}, s);
  if (nstages > 1) stage_body(std::false_type{}, nstages - 2);  // its target is the (ragged) last stage
  if (nstages > 0) {
    const int sl = nstages - 1;
    mma_stage<NB, W>(smem + (sl & 1) * kStageFloats, acc, (nvalid_of(sl) + 3) >> 2, lane);
    __syncthreads();  // every wave is done with the stage buffers (G aliases them)
  }
  if (slot >= 0)
    tiles_to_partial<NB, W>(acc, a.part + (size_t)slot * Geo<NB>::NT * 256, lane);
  else
    finish_row<NB, MODE, W>(acc, smem, a, row, rowlen, tid);
}

template <int NB, typename VT, int MODE>
__global__ __launch_bounds__(kThreads) void als_item_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int item = blockIdx.x;
  const int row = a.item_row[item];
  const long long begin = a.item_begin[item];
  const int len = a.item_len[item];
  const int slot = a.item_slot[item];
  const int rowlen = a.item_rowlen[item];
  // Role of this wave.  The LU on the accumulators loads the roles unevenly (the last role's tiles
  // stay live to the end), so the roles are rotated with the workgroup index: workgroups that
  // share a CU differ in item / 256 (dispatch is round-robin over 8 XCDs x 32 CUs) and their
  // heavy roles then sit on different SIMDs.
  const int role =
      (MODE == kModeLU && lu_on_accumulators(NB)) ? (((tid >> 6) + (item >> 8) + (item >> 10)) & 3) : (tid >> 6);
  switch (role) {
    case 0: item_body<NB, VT, MODE, 0>(smem, a, row, begin, len, slot, rowlen, tid); break;
    case 1: item_body<NB, VT, MODE, 1>(smem, a, row, begin, len, slot, rowlen, tid); break;
    case 2: item_body<NB, VT, MODE, 2>(smem, a, row, begin, len, slot, rowlen, tid); break;
    default: item_body<NB, VT, MODE, 3>(smem, a, row, begin, len, slot, rowlen, tid); break;
  }
  if (slot < 0) solve_row<NB, MODE>(smem, a, row, tid);
}

// ----------------------------------------------------------------------------------
// Kernel 2: one workgroup per chunked row: sum the partial tiles in slot order (a
// fixed, deterministic order) and finish the row.
// ----------------------------------------------------------------------------------
template <int NB, int MODE, int W>
__device__ __forceinline__ void reduce_body(float* smem, const KernelArgs& a, int row, int slot0, int nslots,
                                            int rowlen, int lane) {
  constexpr int TPW = Geo<NB>::TPW;
  constexpr bool NEG = MODE == kModeLU && lu_on_accumulators(NB) && lu_wg_blocked(NB);  // the blocked LU eliminates -A
  f32x4 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < nslots; ++sl)
    partial_accumulate<NB, W, NEG>(acc, a.part + (size_t)(slot0 + sl) * Geo<NB>::NT * 256, lane);
  finish_row<NB, MODE, W, NEG>(acc, smem, a, row, rowlen, lane + 64 * W);
}

// Round 6: the LU of the largest systems (NB = 12, 13: f = 176 .. 207; 164-168 registers = three workgroups per CU) is latency- and
// barrier-bound at full clock (61 % of its wave-cycles parked, profiles/r05/f200_lu/pmc_sq.txt): a fourth workgroup per CU is
// worth more than the registers it spills at 128 (118 at NB = 13, 28 at NB = 12; NB <= 11 fit anyway) -- Netflix f = 200 LU
// Theta side 74.5 -> 71.4 ms (profiles/r06/ab_r13w4.txt).
template <int NB, int MODE>
#ifndef CUMF_REDUCE_LU_WGS
#define CUMF_REDUCE_LU_WGS 4
#endif
__global__ __launch_bounds__(kThreads, (NB >= 11 && MODE == kModeLU) ? CUMF_REDUCE_LU_WGS : 1) void als_reduce_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int mr = blockIdx.x;
  const int row = a.mrow_row[mr];
  const int slot0 = a.dense_slots ? mr : a.mrow_slot0[mr];
  const int nslots = a.dense_slots ? 1 : a.mrow_nslots[mr];
  const int rowlen = a.mrow_rowlen[mr];
  // Role of this wave, rotated with the workgroup index as in als_item_kernel: the LU loads the roles unevenly (role 0 runs
  // the back substitution alone, the owner of a diagonal tile eliminates the pivot blocks), and the waves of a workgroup go
  // to fixed SIMDs -- without the rotation one SIMD of every CU carries all the heavy roles.
  const int role = (MODE == kModeLU && lu_on_accumulators(NB)) ? (((tid >> 6) + (mr >> 8) + (mr >> 10)) & 3) : (tid >> 6);
  switch (role) {
    case 0: reduce_body<NB, MODE, 0>(smem, a, row, slot0, nslots, rowlen, lane); break;
    case 1: reduce_body<NB, MODE, 1>(smem, a, row, slot0, nslots, rowlen, lane); break;
    case 2: reduce_body<NB, MODE, 2>(smem, a, row, slot0, nslots, rowlen, lane); break;
    default: reduce_body<NB, MODE, 3>(smem, a, row, slot0, nslots, rowlen, lane); break;
  }
  solve_row<NB, MODE>(smem, a, row, tid);
}

// ----------------------------------------------------------------------------------
// Standalone batched solvers on materialised systems (the reference's data flow).
// ----------------------------------------------------------------------------------
template <int NB, int MODE>
__global__ __launch_bounds__(kThreads) void solve_lds_kernel(const float* __restrict__ A, const float* __restrict__ b,
                                                             float* __restrict__ x, int f, int cg_iters, int a_half) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const size_t sys = blockIdx.x;
  const float* As = A + sys * (size_t)f * f;  // (fp32 layout; the fp16 layout is indexed below)
  if constexpr (NB != 0 && MODE == kModeLU) {
    lu_solve_reg<NB>(GlobalLoad<NB>{As, b + sys * f, f}, smem, f, smem + lu_packed_floats(NB), x + sys * f,
                               tid);
    return;
  }
  const int ldg = solve_ldg(f, MODE);
  float* G = smem;
  if (a_half) {  // updateXWithCGKernel3 (cg.cu:235-429): A stored as half, arithmetic in fp32
    const _Float16* Ah = reinterpret_cast<const _Float16*>(A) + sys * (size_t)f * f;
    for (int e = tid; e < f * f; e += kThreads) {
      const int i = e / f, j = e - i * f;
      G[i * ldg + j] = (float)Ah[e];
    }
  } else {
    for (int e = tid; e < f * f; e += kThreads) {
      const int i = e / f, j = e - i * f;
      G[i * ldg + j] = As[e];
    }
  }
  if (tid < f) G[tid * ldg + f] = b[sys * f + tid];
  __syncthreads();
  if constexpr (NB == 0)
    lu_solve_lds(G, ldg, f, x + sys * f, tid);
  else if constexpr (MODE == kModeCG)
    cg_solve_lds<NB>(G, ldg, f, smem + solve_g_floats(f, MODE), x + sys * f, cg_iters, tid);
}

#if CUMF_SLICE_COMMON
// CG with A streamed from global memory every mat-vec, for f too large for an
// LDS-resident system (f > 128).  One workgroup per system, thread t owns row t
// (blockDim = f rounded up to 64; same shape as cg.cu:36-231, wave64 reductions).
__global__ void cg_global_kernel(const float* __restrict__ A, float* __restrict__ x, const float* __restrict__ b,
                                 int f, int cg_iters, int a_half) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  float* ps = smem;             // f
  float* red = smem + blockDim.x;  // nwaves
  const float* As = A + (size_t)blockIdx.x * f * f;
  const _Float16* Ah = reinterpret_cast<const _Float16*>(A) + (size_t)blockIdx.x * f * f;  // a_half (cg.cu:253,289)
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
    if (own) {
      if (a_half)
        for (int j = 0; j < f; ++j) s = fmaf((float)Ah[(size_t)j * f + tid], ps[j], s);
      else
        for (int j = 0; j < f; ++j) s = fmaf(As[(size_t)j * f + tid], ps[j], s);
    }
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
// Packed upper triangle of a batch of symmetric f x f Grams (row i keeps columns i .. f-1,
// f (f + 1) / 2 floats per system): the payload of the multi-GPU partial-Gram reduction
// (hugewiki.cu:2703-2717 moves the full f x f per GPU; half of it is redundant).
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void pack_upper_kernel(const float* __restrict__ full, float* __restrict__ packed,
                                                              int f) {
  const size_t sys = blockIdx.x;
  const float* A = full + sys * (size_t)f * f;
  float* P = packed + sys * (size_t)(f * (f + 1) / 2);
  for (int e = threadIdx.x; e < f * f; e += kThreads) {
    const int i = e / f, j = e - i * f;
    if (j >= i) P[i * f - i * (i - 1) / 2 + (j - i)] = A[e];
  }
}
__global__ __launch_bounds__(kThreads) void unpack_upper_kernel(const float* __restrict__ packed, float* __restrict__ full,
                                                                int f) {
  const size_t sys = blockIdx.x;
  float* A = full + sys * (size_t)f * f;
  const float* P = packed + sys * (size_t)(f * (f + 1) / 2);
  for (int e = threadIdx.x; e < f * f; e += kThreads) {
    const int i = e / f, j = e - i * f;
    const int a = i < j ? i : j, b = i < j ? j : i;
    A[e] = P[a * f - a * (a - 1) / 2 + (b - a)];
  }
}
// ----------------------------------------------------------------------------------
// Train SSE from materialised systems (round 4; the multi-GPU `reduce` scheme, where the Gram batch is reduced across
// ranks and solved by a batched solver): sum_u (r - x_u . t)^2 = sum r^2 - (2 t.b - t^T G t) with G = A - reg I.  One
// workgroup per system adds 2 t.b - t^T A t + reg |t|^2 (fp64) to *out; sum r^2 is a constant of the data.  A is read by
// columns (symmetric: y_j = sum_i A[i][j] t_i, coalesced over j).  Systems with reg < 0 (the caller's mark for "no rating": its solution is NaN) are skipped;
// reg == 0 is a valid system (lambda = 0).
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void quadratic_terms_kernel(const float* __restrict__ A, const float* __restrict__ b,
                                                                   const float* __restrict__ x, const float* __restrict__ reg,
                                                                   int f, double* __restrict__ out) {
  __shared__ float xs[256];
  __shared__ double red[kThreads / 64];
  const size_t sys = blockIdx.x;
  const float rg = reg[sys];
  if (!(rg >= 0.f)) return;  // uniform: negative (or NaN) = no rating
  const int tid = threadIdx.x;
  if (tid < f) xs[tid] = x[sys * f + tid];
  __syncthreads();
  double t = 0.0;
  if (tid < f) {
    const float* col = A + sys * (size_t)f * f + tid;
    float y = 0.f;
    for (int i = 0; i < f; ++i) y = fmaf(col[(size_t)i * f], xs[i], y);
    const float xj = xs[tid];
    t = (double)xj * (2.0 * (double)b[sys * f + tid] - (double)y + (double)rg * (double)xj);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if ((tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) sum += red[w];
    atomicAdd(out, sum);
  }
}

// Gram mode "fast": factor table -> (h, l) f16 words of 4096 x (round to nearest even; als_wave.hip
// kArithFast).  Values whose scaled magnitude leaves the f16 range (|x| >= 15.99, +-inf included) are reported
// through *flag (bit 0); NaN entries (rows without ratings) are not.
__global__ __launch_bounds__(256) void presplit_f16x2_kernel(const float* __restrict__ src,
                                                            unsigned* __restrict__ dst, size_t n4, size_t n,
                                                            int* __restrict__ flag) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  auto word = [](float x, bool& bad) {
    const float s = x * 4096.0f;
    // NaN is NOT a range violation: rows / columns without ratings carry NaN factors by design (0/0 in CG, a
    // zero pivot in LU: cg.cu:128) and are never gathered; a NaN that IS gathered shows up in the Gram
    // kernel's own probe (bit 1).  +-inf and finite values beyond the f16 range are flagged.
    bad = bad || (__builtin_fabsf(s) >= 65504.0f);
    const _Float16 h = (_Float16)s;
    const _Float16 l = (_Float16)(s - (float)h);
    h2 w = {h, l};
    return __builtin_bit_cast(unsigned, w);
  };
  bool bad = false;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 o = {word(v[0], bad), word(v[1], bad), word(v[2], bad), word(v[3], bad)};
    reinterpret_cast<u4*>(dst)[i] = o;
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = word(src[i], bad);
  if (bad) atomicOr(flag, 1);
}
hipError_t launch_presplit(const float* src, unsigned* dst, size_t n, int* flag, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const bool aligned = (reinterpret_cast<uintptr_t>(src) % 16 == 0) && (reinterpret_cast<uintptr_t>(dst) % 16 == 0);
  const size_t n4 = aligned ? n / 4 : 0;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(presplit_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n4, n, flag);
  return hipGetLastError();
}

hipError_t launch_pack_upper(const float* full, float* packed, long batch, int f, int unpack, hipStream_t stream) {
  if (batch <= 0) return hipSuccess;
  if (unpack)
    hipLaunchKernelGGL(unpack_upper_kernel, dim3((unsigned)batch), dim3(kThreads), 0, stream, full, packed, f);
  else
    hipLaunchKernelGGL(pack_upper_kernel, dim3((unsigned)batch), dim3(kThreads), 0, stream, full, packed, f);
  return hipGetLastError();
}

hipError_t launch_quadratic_terms(const float* A, const float* b, const float* x, const float* reg, long batch, int f,
                                  double* out, hipStream_t stream) {
  if (batch <= 0) return hipSuccess;
  if (f > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(quadratic_terms_kernel, dim3((unsigned)batch), dim3(kThreads), 0, stream, A, b, x, reg, f, out);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Launchers
// ----------------------------------------------------------------------------------

// Optional per-kernel HIP-event timing of the last half-iteration (bench.py's roofline
// leg): events are recorded on the SAME stream the kernels are launched on.
#endif  // CUMF_SLICE_COMMON

#if CUMF_SLICE_COMMON
bool g_timing = false;
hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
bool g_timed_item = false, g_timed_reduce = false;
#else
extern bool g_timing;
extern hipEvent_t g_ev[3];
extern bool g_timed_item, g_timed_reduce;
#endif
// Every timed launch sequence starts with timing_begin(): it takes the next event triple of a pool (so that the
// launches of a half-iteration made of several -- X_BATCH / THETA_BATCH plans, the pipeline pieces of the multi-GPU
// gather scheme -- can be summed afterwards, kernel_ms_since_reset) and leaves it in g_ev.
void timing_begin();

#if CUMF_SLICE_COMMON
// the Gram(+solve) kernel the last half-iteration dispatched (bench.py reads its name for roofline.kernel)
static std::atomic<const void*> g_last_item_kernel{nullptr};
void note_item_kernel(const void* host_function) { g_last_item_kernel.store(host_function, std::memory_order_relaxed); }
const void* last_item_kernel() { return g_last_item_kernel.load(std::memory_order_relaxed); }
namespace {
struct TimedLaunch {
  hipEvent_t ev[3];
  bool item, reduce;
};
constexpr size_t kTimedPool = 1024;
std::vector<TimedLaunch> g_timed;  // pool of event triples, created on first use
size_t g_timed_used = 0;           // launches since the last reset (the newest one is g_timed[g_timed_used - 1])
std::mutex g_timed_mutex;
}  // namespace
void set_kernel_timing(bool on) {
  std::lock_guard<std::mutex> lock(g_timed_mutex);
  g_timing = on;
  if (on && g_timed.empty()) {
    g_timed.resize(kTimedPool);
    for (auto& t : g_timed) {
      for (auto& e : t.ev) (void)hipEventCreate(&e);
      t.item = t.reduce = false;
    }
  }
}
void timing_begin() {
  if (!g_timing) return;
  std::lock_guard<std::mutex> lock(g_timed_mutex);
  if (g_timed_used > 0) {  // the flags of the previous launch are final now
    g_timed[g_timed_used - 1].item = g_timed_item;
    g_timed[g_timed_used - 1].reduce = g_timed_reduce;
  }
  if (g_timed_used == kTimedPool) g_timed_used = 0;  // nobody read for 1024 launches: start over
  TimedLaunch& t = g_timed[g_timed_used++];
  for (int i = 0; i < 3; ++i) g_ev[i] = t.ev[i];
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
hipError_t kernel_ms_since_reset(float* item_ms, float* reduce_ms, int* launches) {
  std::lock_guard<std::mutex> lock(g_timed_mutex);
  *item_ms = 0.f;
  *reduce_ms = 0.f;
  *launches = (int)g_timed_used;
  if (g_timed_used > 0) {
    g_timed[g_timed_used - 1].item = g_timed_item;
    g_timed[g_timed_used - 1].reduce = g_timed_reduce;
  }
  for (size_t i = 0; i < g_timed_used; ++i) {
    TimedLaunch& t = g_timed[i];
    hipError_t e = hipEventSynchronize(t.ev[2]);
    if (e != hipSuccess) return e;
    float ms = 0.f;
    if (t.item && hipEventElapsedTime(&ms, t.ev[0], t.ev[1]) == hipSuccess) *item_ms += ms;
    if (t.reduce && hipEventElapsedTime(&ms, t.ev[1], t.ev[2]) == hipSuccess) *reduce_ms += ms;
  }
  g_timed_used = 0;
  return hipSuccess;
}
#endif  // CUMF_SLICE_COMMON

template <int NB, typename VT, int MODE>
static hipError_t launch_nb(const KernelArgs& a, long n_items, long n_mrows, hipStream_t stream) {
  const size_t stage_floats = (2 * (size_t)kStage + 8) * Geo<NB>::LD;  // + read-ahead pad of mma_stage
  size_t floats = stage_floats;
  if (MODE != kModeMaterialize) {
    const size_t solve = MODE == kModeLU ? lu_fused_lds_floats<NB>(a.f) : solve_lds_floats(a.f, MODE);
    floats = floats > solve ? floats : solve;
  }
  static const size_t lds_pad = getenv("CUMF_ALS_LDS_PAD") ? (size_t)atol(getenv("CUMF_ALS_LDS_PAD")) : 0;  // occupancy experiments
  const size_t lds = floats * sizeof(float) + lds_pad;
  hipError_t e;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_item_kernel<NB, VT, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_reduce_kernel<NB, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  timing_begin();
  if (g_timing) (void)hipEventRecord(g_ev[0], stream);
  g_timed_item = n_items > 0;
  g_timed_reduce = n_mrows > 0;
  if (n_items > 0) {
    note_item_kernel(reinterpret_cast<const void*>(als_item_kernel<NB, VT, MODE>));
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
    if (mode == kModeCG && a.f <= kVecLd)  // cg_solve_lds holds two vector elements per lane: f <= 128
      return v4 ? launch_nb<NB, f32x4, kModeCG>(a, n_items, n_mrows, stream)
                : launch_nb<NB, f32x2, kModeCG>(a, n_items, n_mrows, stream);
  }
  // the register LU keeps only the packed upper triangle in LDS: fused up to f = 200
  if (mode == kModeLU)
    return v4 ? launch_nb<NB, f32x4, kModeLU>(a, n_items, n_mrows, stream)
              : launch_nb<NB, f32x2, kModeLU>(a, n_items, n_mrows, stream);
  return hipErrorInvalidValue;
}

template <int NB, int MODE>
static hipError_t launch_solve_nb(const float* A, const float* b, float* x, long batch, int f, int cg_iters,
                                  hipStream_t stream, int a_half = 0) {
  const size_t floats = NB == 0 ? solve_lds_floats(f, kModeLUExact)
                                : (MODE == kModeLU ? lu_lds_floats(NB, f) : solve_lds_floats(f, MODE));
  const size_t lds = floats * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(solve_lds_kernel<NB, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((solve_lds_kernel<NB, MODE>), dim3((unsigned)batch), dim3(kThreads), lds, stream, A, b, x, f,
                     cg_iters, a_half);
  return hipGetLastError();
}

// wave-per-item kernels (als_wave.hip), one translation unit per NB
template <int NB>
hipError_t wave_item_launch(const KernelArgs& a, int mode, long n_items, hipStream_t stream);
template <int NB>
hipError_t wave_solve_launch(const KernelArgs& a, int mode, long n_rows, hipStream_t stream);
#define CUMF_DECLARE_WAVE(N)                                                                       \
  template <>                                                                                      \
  hipError_t wave_item_launch<N>(const KernelArgs& a, int mode, long n_items, hipStream_t stream); \
  template <>                                                                                      \
  hipError_t wave_solve_launch<N>(const KernelArgs& a, int mode, long n_rows, hipStream_t stream);
CUMF_DECLARE_WAVE(2) CUMF_DECLARE_WAVE(3) CUMF_DECLARE_WAVE(4) CUMF_DECLARE_WAVE(5)
CUMF_DECLARE_WAVE(6) CUMF_DECLARE_WAVE(7) CUMF_DECLARE_WAVE(8) CUMF_DECLARE_WAVE(9) CUMF_DECLARE_WAVE(10)
CUMF_DECLARE_WAVE(11) CUMF_DECLARE_WAVE(12) CUMF_DECLARE_WAVE(13)

// reduce kernel of the chunked rows on its own (the items came from the wave-per-item kernel)
template <int NB>
static hipError_t launch_reduce_only(const KernelArgs& a, int mode, long n_mrows, hipStream_t stream) {
  if (n_mrows <= 0) return hipSuccess;
  if constexpr (NB >= 2) {
    if (gram_mode() != kGramExact && !getenv("CUMF_ALS_NO_WAVE_SOLVE")) {
      // CG on the tiles at wave level: the chunked rows of the wave kernels (NB <= 7) and the systems too
      // large for the LDS-resident 4-wave CG (f > 128; two waves share the tiles).  NB = 8, 9 (f = 112 ..
      // 128) stay on the 4-wave solvers: measured 38.8 vs 56.3 ms (CG) and 49.0 vs 62.2 ms (LU) per Netflix
      // iteration at f = 128 -- one wave holding 45 tiles spills and has nobody to overlap with.
      if (mode == kModeCG && (NB <= kMaxWaveNB || a.f > kVecLd)) return wave_solve_launch<NB>(a, mode, n_mrows, stream);
    }
  }
  if (mode == kModeMaterialize) {
    hipLaunchKernelGGL((als_reduce_kernel<NB, kModeMaterialize>), dim3((unsigned)n_mrows), dim3(kThreads), 0, stream, a);
  } else if (mode == kModeCG) {
    if constexpr (NB <= kMaxFusedNB) {
      if (a.f > kVecLd) return hipErrorInvalidValue;
      const size_t lds = solve_lds_floats(a.f, kModeCG) * sizeof(float);
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_reduce_kernel<NB, kModeCG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL((als_reduce_kernel<NB, kModeCG>), dim3((unsigned)n_mrows), dim3(kThreads), lds, stream, a);
    } else {
      return hipErrorInvalidValue;
    }
  } else {
    const size_t lds = lu_fused_lds_floats<NB>(a.f) * sizeof(float);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_reduce_kernel<NB, kModeLU>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((als_reduce_kernel<NB, kModeLU>), dim3((unsigned)n_mrows), dim3(kThreads), lds, stream, a);
  }
  return hipGetLastError();
}
template <int NB>
hipError_t slice_reduce_only(const KernelArgs& a, int mode, long n_mrows, hipStream_t stream);


// Large systems (f >= 112): the Gram of every row is dumped as accumulator tiles (two waves per item,
// als_wave_multi_kernel) and a solver kernel (single-wave LU up to NB = 10, the 4-wave lu_solve_mfma
// above that, wave-level CG on the tiles) picks them up -- the reference's own data flow ("Gram batch
// in device memory, separate solver", als.cu:782-831), with tiles instead of full f x f matrices and
// in batches of the pooled tile buffer (als_plan.cpp: up to 48 GiB, usually ONE batch).
template <int NB>
static hipError_t launch_batched_nb(const KernelArgs& a0, int mode, const PlanLists& L, hipStream_t stream) {
  if constexpr (NB < 2) {
    return hipErrorInvalidValue;
  } else {
  hipError_t e = hipSuccess;
  // 1. chunked rows: their items write the plan's slots, the reduce kernel sums and solves
  if (L.n_citems > 0) {
    KernelArgs a = a0;
    a.item_row = L.c_row;
    a.item_begin = L.c_begin;
    a.item_len = L.c_len;
    a.item_slot = L.c_slot;
    a.item_rowlen = L.c_rowlen;
    e = wave_item_launch<NB>(a, kModeLU, L.n_citems, stream);  // every item has a slot: nothing is solved in place
    if (e != hipSuccess) return e;
    e = launch_reduce_only<NB>(a0, mode, L.n_mrows, stream);
    if (e != hipSuccess) return e;
  }
  // 2. whole rows.  CG, and LU up to NB = 9: solved by the two waves that formed the Gram, in one launch.
  // Larger LUs go on through the tile buffer: two waves on a 200 x 200 elimination (one wave per SIMD, a
  // barrier per panel) lose more than the round trip costs (Netflix f = 200: 112 vs 106 ms, f = 160: 85
  // vs 78; f = 128: 39.0 vs 41.6 the other way).
  if (mode == kModeCG || (mode == kModeLU && NB <= kMaxFusedLuWaveNB)) {
    if (L.n_witems <= 0) return hipSuccess;
    KernelArgs a = a0;
    a.item_row = L.w_row;
    a.item_begin = L.w_begin;
    a.item_len = L.w_len;
    a.item_rowlen = L.w_rowlen;
    a.item_slot = nullptr;  // no slots: nothing is dumped
    a.dense_slots = 0;
    return wave_item_launch<NB>(a, mode, L.n_witems, stream);
  }
  // large LU / materialise (cumf_get_hermitian): in batches of part2_rows dense slots
  for (long w0 = 0; w0 < L.n_witems; w0 += L.part2_rows) {
    const long cnt = L.n_witems - w0 < L.part2_rows ? L.n_witems - w0 : L.part2_rows;
    KernelArgs a = a0;
    a.item_row = L.w_row + w0;
    a.item_begin = L.w_begin + w0;
    a.item_len = L.w_len + w0;
    a.item_rowlen = L.w_rowlen + w0;
    a.item_slot = nullptr;
    a.dense_slots = 1;
    a.part = L.part2;
    a.mrow_row = L.w_row + w0;
    a.mrow_rowlen = L.w_rowlen + w0;
    e = wave_item_launch<NB>(a, kModeLU, cnt, stream);
    if (e != hipSuccess) return e;
    e = launch_reduce_only<NB>(a, mode, cnt, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
  }
}
template <int NB>
hipError_t slice_batched(const KernelArgs& a, int mode, const PlanLists& L, hipStream_t stream);

// Per-NB entry points (one translation unit each, see CUMF_NB_SLICE above).
template <int NB>
hipError_t slice_half_iteration(const KernelArgs& a, int mode, long n_items, long n_mrows, hipStream_t stream);
template <int NB>
hipError_t slice_solve(const float* A, const float* b, float* x, long batch, int f, int mode, int cg_iters,
                       hipStream_t stream);

#define CUMF_DECLARE_SLICE(N)                                                                                   \
  template <>                                                                                                    \
  hipError_t slice_half_iteration<N>(const KernelArgs& a, int mode, long n_items, long n_mrows, hipStream_t stream); \
  template <>                                                                                                    \
  hipError_t slice_solve<N>(const float* A, const float* b, float* x, long batch, int f, int mode, int cg_iters, \
                            hipStream_t stream);                                                                 \
  template <>                                                                                                    \
  hipError_t slice_reduce_only<N>(const KernelArgs& a, int mode, long n_mrows, hipStream_t stream);              \
  template <>                                                                                                    \
  hipError_t slice_batched<N>(const KernelArgs& a, int mode, const PlanLists& L, hipStream_t stream);
#define CUMF_STUB_SLICE(N)                                                                                       \
  template <>                                                                                                    \
  hipError_t slice_half_iteration<N>(const KernelArgs&, int, long, long, hipStream_t) {                          \
    return hipErrorInvalidValue;                                                                                 \
  }                                                                                                              \
  template <>                                                                                                    \
  hipError_t slice_solve<N>(const float*, const float*, float*, long, int, int, int, hipStream_t) {              \
    return hipErrorInvalidValue;                                                                                 \
  }                                                                                                              \
  template <>                                                                                                    \
  hipError_t slice_reduce_only<N>(const KernelArgs&, int, long, hipStream_t) {                                   \
    return hipErrorInvalidValue;                                                                                 \
  }                                                                                                              \
  template <>                                                                                                    \
  hipError_t slice_batched<N>(const KernelArgs&, int, const PlanLists&, hipStream_t) {                           \
    return hipErrorInvalidValue;                                                                                 \
  }
#define CUMF_DEFINE_SLICE(N)                                                                                     \
  template <>                                                                                                    \
  hipError_t slice_half_iteration<N>(const KernelArgs& a, int mode, long n_items, long n_mrows,                 \
                                     hipStream_t stream) {                                                       \
    return launch_mode<N>(a, mode, n_items, n_mrows, stream);                                                    \
  }                                                                                                              \
  template <>                                                                                                    \
  hipError_t slice_solve<N>(const float* A, const float* b, float* x, long batch, int f, int mode, int cg_iters, \
                            hipStream_t stream) {                                                                \
    if (mode == kModeCG || mode == kModeCGHalf) {                                                                \
      if constexpr (N <= kMaxFusedNB)                                                                            \
        return launch_solve_nb<N, kModeCG>(A, b, x, batch, f, cg_iters, stream, mode == kModeCGHalf);            \
      return hipErrorInvalidValue;                                                                               \
    }                                                                                                            \
    return launch_solve_nb<N, kModeLU>(A, b, x, batch, f, cg_iters, stream);                                     \
  }                                                                                                              \
  template <>                                                                                                    \
  hipError_t slice_reduce_only<N>(const KernelArgs& a, int mode, long n_mrows, hipStream_t stream) {             \
    return launch_reduce_only<N>(a, mode, n_mrows, stream);                                                      \
  }                                                                                                              \
  template <>                                                                                                    \
  hipError_t slice_batched<N>(const KernelArgs& a, int mode, const PlanLists& L, hipStream_t stream) {           \
    return launch_batched_nb<N>(a, mode, L, stream);                                                             \
  }

CUMF_DECLARE_SLICE(1) CUMF_DECLARE_SLICE(2) CUMF_DECLARE_SLICE(3) CUMF_DECLARE_SLICE(4) CUMF_DECLARE_SLICE(5)
CUMF_DECLARE_SLICE(6) CUMF_DECLARE_SLICE(7) CUMF_DECLARE_SLICE(8) CUMF_DECLARE_SLICE(9) CUMF_DECLARE_SLICE(10)
CUMF_DECLARE_SLICE(11) CUMF_DECLARE_SLICE(12) CUMF_DECLARE_SLICE(13)
#if CUMF_SLICE_HAS(1)
CUMF_DEFINE_SLICE(1)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(1)
#endif
#if CUMF_SLICE_HAS(2)
CUMF_DEFINE_SLICE(2)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(2)
#endif
#if CUMF_SLICE_HAS(3)
CUMF_DEFINE_SLICE(3)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(3)
#endif
#if CUMF_SLICE_HAS(4)
CUMF_DEFINE_SLICE(4)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(4)
#endif
#if CUMF_SLICE_HAS(5)
CUMF_DEFINE_SLICE(5)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(5)
#endif
#if CUMF_SLICE_HAS(6)
CUMF_DEFINE_SLICE(6)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(6)
#endif
#if CUMF_SLICE_HAS(7)
CUMF_DEFINE_SLICE(7)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(7)
#endif
#if CUMF_SLICE_HAS(8)
CUMF_DEFINE_SLICE(8)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(8)
#endif
#if CUMF_SLICE_HAS(9)
CUMF_DEFINE_SLICE(9)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(9)
#endif
#if CUMF_SLICE_HAS(10)
CUMF_DEFINE_SLICE(10)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(10)
#endif
#if CUMF_SLICE_HAS(11)
CUMF_DEFINE_SLICE(11)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(11)
#endif
#if CUMF_SLICE_HAS(12)
CUMF_DEFINE_SLICE(12)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(12)
#endif
#if CUMF_SLICE_HAS(13)
CUMF_DEFINE_SLICE(13)
#elif CUMF_ONLY_NB != 0
CUMF_STUB_SLICE(13)
#endif

#if CUMF_SLICE_COMMON
#define CUMF_NB_CASE(N, call) case N: return call;

static int g_gram_mode = -1;
void set_gram_mode(int mode) { g_gram_mode = (mode == kGramExact || mode == kGramFast) ? mode : kGramAuto; }
int gram_mode() {
  if (g_gram_mode < 0) {
    const char* e = getenv("CUMF_ALS_GRAM");  // exact | fast | split (default)
    g_gram_mode = (e && (e[0] == 'e' || e[0] == 'E')) ? kGramExact : (e && (e[0] == 'f' || e[0] == 'F')) ? kGramFast : kGramAuto;
  }
  return g_gram_mode;
}
bool wave_path_available(int f, int mode) {
  // NB = 1 (f <= 14: one tile, a 32-rating MFMA K and one wave per row are all overhead) stays on the
  // workgroup kernels: measured 0.56 vs 3.4 ms per iteration at f = 10, while f = 20 .. 48 is 1.4-1.7x
  // faster on the wave kernels
  return gram_mode() != kGramExact && nb_for_f(f) >= 2 && nb_for_f(f) <= kMaxWaveNB &&
         (mode == kModeLU || mode == kModeMaterialize || mode == kModeCG);
}

bool wave_batched_path(int f, int mode) {
  if (gram_mode() == kGramExact || getenv("CUMF_ALS_NO_BATCHED")) return false;
  const int nb = nb_for_f(f);
  // f >= 112 (NB 8 .. 13): two waves per item form the Gram, a solver kernel picks the tiles up
  // (f <= 111: everything runs inside the wave-per-item kernel, see wave_path_available)
  return nb > kMaxWaveNB && nb <= nb_for_f(kMaxF) && (mode == kModeLU || mode == kModeMaterialize || mode == kModeCG);
}

hipError_t launch_half_iteration(const KernelArgs& a, int mode, long n_items, long n_mrows, hipStream_t stream,
                                 const PlanLists* lists) {
  if (lists != nullptr && wave_batched_path(a.f, mode)) {
    timing_begin();
    if (g_timing) (void)hipEventRecord(g_ev[0], stream);
    g_timed_item = true;
    g_timed_reduce = false;
    hipError_t e = hipErrorInvalidValue;
    switch (nb_for_f(a.f)) {
#define CUMF_BATCHED(N) case N: e = slice_batched<N>(a, mode, *lists, stream); break;
      CUMF_BATCHED(8) CUMF_BATCHED(9) CUMF_BATCHED(10) CUMF_BATCHED(11) CUMF_BATCHED(12) CUMF_BATCHED(13)
#undef CUMF_BATCHED
      default: break;
    }
    if (g_timing) {
      (void)hipEventRecord(g_ev[1], stream);
      (void)hipEventRecord(g_ev[2], stream);
    }
    return e;
  }
  if (wave_path_available(a.f, mode)) {
    timing_begin();
    if (g_timing) (void)hipEventRecord(g_ev[0], stream);
    g_timed_item = n_items > 0;
    g_timed_reduce = n_mrows > 0;
    hipError_t e = hipSuccess;
    KernelArgs aw = a;
    aw.whole_only = n_mrows == 0;  // no chunked row in the plan: every item is a whole row
    // A plan with a FEW chunked rows (a Theta side with a handful of very long columns): their chunk items go first, in a
    // launch of their own, and the whole rows keep the instance without the dump exit (no spilled accumulators, VERDICT r03
    // weak 7).  When the chunks are most of the work (the Netflix X side: 86 % of the ratings) one combined launch stays:
    // the whole rows fill the tail of the equal-sized chunk items, worth more than the spills cost.  (Round 5 measured
    // the alternative for that side too -- the chunk items through the dump-only instance, 0 spilled registers, one after
    // the other or side by side on a second stream: 6.83 / 6.82 ms against 6.69-6.85 combined, same box,
    // profiles/r05/stage_variants_ab.txt: the 63 spilled registers of the combined instance cost nothing measurable.)
    KernelArgs ac = a;
    long n_chunk_first = 0;
    if (mode == kModeLU && n_mrows > 0 && lists != nullptr && lists->n_citems > 0 && lists->n_witems > 0 &&
        lists->chunk_share < 0.25) {  // (LU: the only mode with an instance for whole rows alone)
      n_chunk_first = lists->n_citems;
      ac.item_row = lists->c_row, ac.item_begin = lists->c_begin, ac.item_len = lists->c_len;
      ac.item_slot = lists->c_slot, ac.item_rowlen = lists->c_rowlen;
      ac.whole_only = 0;
      aw.item_row = lists->w_row, aw.item_begin = lists->w_begin, aw.item_len = lists->w_len;
      aw.item_slot = nullptr, aw.item_rowlen = lists->w_rowlen;
      aw.whole_only = 1;
      n_items = lists->n_witems;
    }
    // Round 6: the short whole rows of a CG plan (the last n_short items) never form their Gram matrix (als_short.hip)
    static const bool short_cg_on = !getenv("CUMF_ALS_SHORT_CG") || atoi(getenv("CUMF_ALS_SHORT_CG")) != 0;
    long n_short = 0;
    if (mode == kModeCG && short_cg_on && lists != nullptr && short_cg_available(a.f) && !a.dense_slots) n_short = lists->n_short;
    KernelArgs as = aw;
    if (n_short > 0) {
      n_items -= n_short;
      as.item_row += n_items, as.item_begin += n_items, as.item_len += n_items, as.item_rowlen += n_items;
      e = launch_short_cg(as, n_short, stream);  // first: the tail of the long items then fills in behind it
      if (e != hipSuccess) return e;
    }
    switch (nb_for_f(a.f)) {
#define CUMF_WAVE(N)                                              \
  case N:                                                         \
    if (n_chunk_first > 0) {                                      \
      e = wave_item_launch<N>(ac, mode, n_chunk_first, stream);   \
      if (e != hipSuccess) return e;                              \
    }                                                             \
    e = wave_item_launch<N>(aw, mode, n_items, stream);           \
    if (e != hipSuccess) return e;                                \
    if (g_timing) (void)hipEventRecord(g_ev[1], stream);          \
    e = slice_reduce_only<N>(a, mode, n_mrows, stream);           \
    break;
      CUMF_WAVE(2) CUMF_WAVE(3) CUMF_WAVE(4) CUMF_WAVE(5) CUMF_WAVE(6) CUMF_WAVE(7)
#undef CUMF_WAVE
      default: return hipErrorInvalidValue;
    }
    if (g_timing) (void)hipEventRecord(g_ev[2], stream);
    return e;
  }
#define CUMF_HALF(N) CUMF_NB_CASE(N, slice_half_iteration<N>(a, mode, n_items, n_mrows, stream))
  switch (nb_for_f(a.f)) {
    CUMF_HALF(1) CUMF_HALF(2) CUMF_HALF(3) CUMF_HALF(4) CUMF_HALF(5) CUMF_HALF(6) CUMF_HALF(7)
    CUMF_HALF(8) CUMF_HALF(9) CUMF_HALF(10) CUMF_HALF(11) CUMF_HALF(12) CUMF_HALF(13)
    default: return hipErrorInvalidValue;
  }
#undef CUMF_HALF
}

hipError_t launch_solve_batched(const float* A, const float* b, float* x, long batch, int f, int mode, int cg_iters,
                                hipStream_t stream) {
  if (batch <= 0) return hipSuccess;
  if (mode == kModeLUExact) return launch_solve_nb<0, kModeLU>(A, b, x, batch, f, 0, stream);
  if (f > 128 && (mode == kModeCG || mode == kModeCGHalf)) {  // system too large for the LDS: A streamed from global memory
    const int threads = ((f + 63) / 64) * 64;
    const size_t lds = (threads + 16) * sizeof(float);
    hipLaunchKernelGGL(cg_global_kernel, dim3((unsigned)batch), dim3(threads), lds, stream, A, x, b, f, cg_iters,
                       (int)(mode == kModeCGHalf));
    return hipGetLastError();
  }
  // LDS-resident CG (f <= 128) or register-resident LU (f <= 200)
#define CUMF_SOLVE(N) CUMF_NB_CASE(N, slice_solve<N>(A, b, x, batch, f, mode, cg_iters, stream))
  switch (nb_for_f(f)) {
    CUMF_SOLVE(1) CUMF_SOLVE(2) CUMF_SOLVE(3) CUMF_SOLVE(4) CUMF_SOLVE(5) CUMF_SOLVE(6) CUMF_SOLVE(7)
    CUMF_SOLVE(8) CUMF_SOLVE(9) CUMF_SOLVE(10) CUMF_SOLVE(11) CUMF_SOLVE(12) CUMF_SOLVE(13)
    default: return hipErrorInvalidValue;
  }
#undef CUMF_SOLVE
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

#endif  // CUMF_SLICE_COMMON

}  // namespace cumf
