// als_wave.hip -- wave-level ALS half-iteration kernels for gfx950 (16 <= f <= 207).
//
// Same job as als_item_kernel (als_kernels.hip) -- RHS + Gram + solve of one plan item,
// replacing cusparseScsrmm2 + cublasSgeam (als.cu:750-757), get_hermitian100 /
// get_hermitianT10 (als.cu:443-569 / 575-659), the batched LU (als.cu:58-189) and the CG of
// cg.cu:36-231 -- with a different mapping onto the chip:
//
//   * als_wave_kernel (NB = 2 .. 7, f <= 111): ONE wave owns one item and all NB (NB + 1) / 2
//     upper-triangular 16 x 16 accumulator tiles of its system.  No workgroup barrier anywhere: every
//     lane gathers straight into the MFMA operand layout -- lane (g, c) = (lane >> 4, lane & 15)
//     fetches feature 16 b + c of the eight ratings 8 g .. 8 g + 7 of a 32-rating stage, one
//     4-byte global_load_lds_dword per (feature block b, rating): a wave instruction touches four
//     64-byte segments of four gathered factor rows and lands as one 256-byte chunk in LDS, a full
//     stage ahead of its use and without holding registers.  64-bit lane addresses: no 4 GiB limit.
//     als_wave_multi_kernel (NB = 8 .. 13): two waves per item share the chunks and split the tiles.
//   * fp32 on the bf16 matrix pipe.  The fp32 MFMA runs at 1/16 of the bf16 rate, and a
//     16-wide tiling of a 101-column system computes 1.42x the useful flops: at 157 TF that is
//     9.1 ms per Netflix half-iteration against 5.05 ms of HBM time.  Here every
//     gathered fp32 value x is split exactly into three bf16 terms x = h + m + l
//     (round-to-nearest; 8 + 8 + 8 significand bits) and the product of two values is
//     evaluated as hh + hm + mh + mm + hl + lh on v_mfma_f32_16x16x32_bf16 with fp32
//     accumulation: the dropped terms (ml, lm, ll) are below 2^-23 of the product, i.e.
//     below the rounding error of one fp32 fmaf on a sum of two such products.  Every bf16
//     product is exact in fp32.  Error against an fp64 Gram is the same class as the fmaf
//     chain's (tests/test_gpu_parity.py::test_split_gram_error_class); the bit-exact fp32
//     MFMA path stays available (cumf_set_gram_mode / CUMF_ALS_GRAM=exact), and an opt-in 22-bit
//     arithmetic on pre-split f16 pairs (kArithFast, CUMF_ALS_GRAM=fast) halves the matrix-pipe work.
//   * LU on the accumulators of the one wave (lu_wave): four pivots per step as one rank-4
//     v_mfma_f32_16x16x4_f32 per live tile (the elimination of lu_solve_mfma, als_lu_wg.h),
//     but the 4 x 4 pivot block comes from v_readlane, the panel rows reach the other lane
//     groups through ds_bpermute_b32, and nothing waits on another wave.  Back substitution
//     straight from the tiles through a 16-column LDS window (back_substitute_tiles).
//   * CG on the accumulators (cg_wave_core): vectors in a column layout, the mat-vec on the upper
//     tiles with DPP / ds_bpermute reductions, 1, 2 or 4 waves per system.
//
// The accumulator layout (C/D of every 16 x 16 MFMA: lane (g, c), register r = element
// (4 g + r, c)) and the partial-tile scratch layout are those of als_kernels.hip, so chunked
// rows go through the same als_reduce_kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "als_device.h"
#include "als_lu_wg.h"
#include "als_lu_blocked.h"
#include "als_internal.h"

namespace cumf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(3))) float* lds_float_ptr;  // LDS pointer kept in its own address space (M0 operand)

#ifndef CUMF_WAVE_NB
#error "compile with -DCUMF_WAVE_NB=<feature blocks>"
#endif
#ifndef CUMF_ABLATE_STAGE
#define CUMF_ABLATE_STAGE 0
#endif

// feature-block counts whose kernels also exist on the pre-split table (kArithPre): f = 96 .. 111 and f = 64 .. 79 -- the
// headline f = 100 and BASELINE configs[4]'s f = 64 (presplit_nb_ok in als_internal.h is the host's copy of this list)
#define CUMF_WAVE_PRE (CUMF_WAVE_NB == 7 || CUMF_WAVE_NB == 5)
// feature-block counts with kArithSplitPk instances (the in-kernel split with the rating-only last block packed, f % 16 == 0)
#define CUMF_WAVE_SPLITPK (CUMF_WAVE_NB <= 7)
constexpr int kWaveStage = 32;   // ratings per stage = K of v_mfma_f32_16x16x32_bf16
constexpr int kZeroFloats = 256; // >= 16 * kMaxWaveNB + 16

// Zeros that stand in for "no rating here": ratings past the end of an item gather from
// this row, the pad lanes of the last feature block read it too.
static __device__ __attribute__((aligned(16))) float g_wave_zeros[kZeroFloats];
#if CUMF_ABLATE
// profiling build, switch 65536: how many CG iterations (mat-vecs behind the initial residual) the rows actually ran before
// ||r||^2 < 1e-4 ended the loop (cg.cu:195) -- bin k = rows that ran k iterations (cumf_debug_cg_histogram)
static __device__ __attribute__((unused)) unsigned long long g_cg_hist[16];
#endif

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Arithmetic of the Gram pass.  kArithSplit3: exact bf16x3 split, six products (24 significand bits).
// kArithFast: the gather table arrives PRE-SPLIT (presplit_f16x2_kernel, als_kernels.hip): one 32-bit word
// per value = (h, l) f16 pair of 4096 x, x ~ (h + l) / 4096 to 2^-22; three f16 products hh + hl + lh
// (the dropped ll is < 2^-22 of the product), fp32 accumulation, accumulators scaled back by 2^-24.
// kArithPre (round 6): the 24-bit arithmetic of kArithSplit3 -- the same three bf16 planes, the same products, the same
// K slots, bit-identical accumulators -- from a gather table that arrives PRE-SPLIT (presplit_bf16x3_kernel: per row the
// h | m | l planes as 16-bit arrays): the rows are copied into LDS by 16-byte LDS-DMA and the MFMA operands come out of
// ds_read_b64_tr_b16, the 16-bit transposing read of gfx950 -- no split, no pack, no select on the VALU.  For gather
// tables that live in the caches (the Netflix Theta side: X = 7 MB; the hugewiki X side: Theta = 16 MB); an HBM-resident
// table stays fp32 (1.5 x the bytes would cost more than the VALU work saves).
enum { kArithSplit3 = 0, kArithFast = 1, kArithPre = 2, kArithPrePk = 3, kArithSplitPk = 4 };
// kArithSplitPk (round 6): the in-kernel split with the last block PACKED, for f % 16 == 0 -- there the last feature block holds
// nothing but the rating slot, and six products per tile of its column (four on its diagonal tile) multiply 15 zero columns.
// The rating is loaded by the lanes of columns 0, 1, 2, split like a feature, and lane c keeps plane c: the packed operand
// [r_h r_m r_l 0 ...] of kArithPrePk -- three products h_I pk + m_I pk + l_I pk per tile (all nine plane products), one (pk pk^T)
// on the diagonal tile, folded back once per item by wave_fold_strip; the gather skips the block (no LDS-DMA, no chunk).
// 65 instead of 80 MFMAs and 32 instead of 40 gathers per stage at f = 64.
constexpr float kFastScale = 4096.0f;             // values must stay below 65504 / 4096 = 15.99 in magnitude
constexpr float kFastUnscale = 1.0f / (4096.0f * 4096.0f);

// (h, l) word of one fp32 value, as presplit_f16x2_kernel makes them (round to nearest even)
__device__ __forceinline__ unsigned fast_word(float x) {
  const float s = x * kFastScale;
  const _Float16 h = (_Float16)s;
  const _Float16 l = (_Float16)(s - (float)h);
  f16x2 w = {h, l};
  return __builtin_bit_cast(unsigned, w);
}

// ----------------------------------------------------------------------------------
// One 32-rating stage in flight: the raw gathered values in the operand layout and its
// column indices.  Planes: the three bf16 terms of a converted stage.
// ----------------------------------------------------------------------------------
template <int NB>
struct WaveStage {
  float raw[NB][8];  // raw[b][e]: feature 16 b + c of rating 8 g + e
  float rv[8];       // rating values (lanes of slot f; zeros elsewhere)
  int idx[8];        // column indices of a stage whose gathers are still to be issued
};
template <int NB, int ARITH = kArithSplit3>
struct Planes {
  u32x4 h[NB], m[NB], l[NB];
};
template <int NB>
struct Planes<NB, kArithFast> {
  u32x4 h[NB], l[NB];
};

// Loop-invariant per-lane state of the gather.  FULL forms assume every rating of the stage
// exists (16-byte index / rating loads, no selects); the generic forms clamp the index loads to
// the item, point the ratings past its end at the zero row and zero their rating value.
template <int NB>
struct WaveGather {
  const char* lane_base;   // gather table + 4 c
  const char* zero_base;   // zero row + 4 c
  const float* val_base;   // val + begin + 8 g in the lanes of slot f, the zero row elsewhere
  const int* idx_base;     // colidx + begin + 8 g
  long long last_off;      // last feature block: byte offset of this lane's load from the row pointer
  unsigned row_bytes;
  int g, len;
#if CUMF_ABLATE
  int dbg;
  unsigned idx_mask;
#endif
  bool is_feat, is_val;    // last feature block: this lane holds a feature / the rating slot
  bool has_val;            // is_val and the item has ratings (an empty row reads zeros instead)

  __device__ __forceinline__ void init(const KernelArgs& a, int f, long long begin, int len_, int lane, bool pk3 = false) {
    const int c = lane & 15;
    g = lane >> 4;
    len = len_;
    row_bytes = (unsigned)f * 4u;
#if CUMF_ABLATE
    if (a.dbg & 8) row_bytes = 0u;  // ablation: every gather hits row 0
    dbg = a.dbg;
    // ablation: the gathers keep their shape (four 64-byte row segments per instruction) but only touch the first
    // 64 / 4096 / 65536 rows of the table (L1- / L2- / MALL-resident at f = 100): profiles/r03/gather_ablation.txt
    idx_mask = (a.dbg & 32) ? 63u : (a.dbg & 64) ? 4095u : (a.dbg & 128) ? 65535u : 0xffffffffu;
#endif
    lane_base = reinterpret_cast<const char*>(a.gather) + 4 * c;
    zero_base = reinterpret_cast<const char*>(g_wave_zeros) + 4 * c;
    const int fi = 16 * (NB - 1) + c;
    is_feat = fi < f;
    is_val = pk3 ? c < 3 : fi == f;  // kArithSplitPk (f % 16 == 0): the rating in the lanes of columns 0, 1, 2
    // lanes behind the features re-read the start of the row (in bounds) and drop the value
    last_off = is_feat ? 64 * (NB - 1) : -4 * c;
    // A row without ratings still runs ONE stage, on zeros (the accumulators then flow from the stage loop into
    // the solver without a merge with a "no stage" path: that merge cost 41 spilled registers in the LU kernel).
    // Its index / rating loads must not touch colidx / val (begin may be the end of the arrays): they read the
    // zero row; "+ 8 g + 1" because the clamped index forms address the item's last rating at base[-1 - 8 g].
    has_val = is_val && len_ > 0 && a.val != nullptr;  // val == nullptr: no ratings given, the slot reads zeros
    val_base = has_val ? a.val + begin + 8 * g : g_wave_zeros;
    idx_base = len_ > 0 ? a.colidx + begin + 8 * g : reinterpret_cast<const int*>(g_wave_zeros) + 8 * g + 1;
  }

  template <bool FULL>
  __device__ __forceinline__ void load_idx(WaveStage<NB>& st, int s) const {
    const int* p = idx_base + kWaveStage * s;
    if constexpr (FULL) {
      const i32x4u lo = *reinterpret_cast<const i32x4u*>(p);
      const i32x4u hi = *reinterpret_cast<const i32x4u*>(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st.idx[e] = lo[e];
        st.idx[4 + e] = hi[e];
      }
    } else {
      const int last = len - 1 - (kWaveStage * s + 8 * g);  // offset of the item's last rating from p
#pragma unroll
      for (int e = 0; e < 8; ++e) st.idx[e] = p[e < last ? e : last];
    }
  }

  // the rating rides in slot f of the last block (als.cu:750-757 fused into the Gram: column f of
  // the last tile column is sum r * theta = the right-hand side)
  template <bool FULL>
  __device__ __forceinline__ void load_val(WaveStage<NB>& st, int s) const {
    const float* vp = val_base + (has_val ? kWaveStage * s : 0);
    if constexpr (FULL) {
      const f32x4u lo = *reinterpret_cast<const f32x4u*>(vp);
      const f32x4u hi = *reinterpret_cast<const f32x4u*>(vp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st.rv[e] = lo[e];
        st.rv[4 + e] = hi[e];
      }
    } else {
      // ratings past the end of the item read the item's last rating (in bounds) and are then zeroed: their factor
      // rows are the zero row, so the right-hand side never saw them, but entry (f, f) of the augmented Gram --
      // sum r^2, what the fused train SSE starts from (wave_tile_ff) -- would
      const int last = has_val ? len - 1 - (kWaveStage * s + 8 * g) : 7;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = vp[e < last ? e : last];
        st.rv[e] = e <= last ? v : 0.f;
      }
    }
  }

  // row pointer (+ 4 c) of rating e of stage s, from st.idx
  template <bool FULL, int E>
  __device__ __forceinline__ const char* row_ptr(const WaveStage<NB>& st, int s) const {
#if CUMF_ABLATE
    const char* row = lane_base + (unsigned long long)((unsigned)st.idx[E] & idx_mask) * row_bytes;
#else
    const char* row = lane_base + (unsigned long long)(unsigned)st.idx[E] * row_bytes;  // v_mad_u64_u32
#endif
    if constexpr (!FULL) {
      const int left = len - (kWaveStage * s + 8 * g);  // ratings of the item from this lane group's first on
      const char* z = zero_base;
      row = (E < left) ? row : z;
    }
    return row;
  }
  template <int B, int E>
  __device__ __forceinline__ void load_one(WaveStage<NB>& st, const char* row) const {
    if constexpr (B + 1 < NB)
      st.raw[B][E] = *reinterpret_cast<const float*>(row + 64 * B);
    else
      st.raw[B][E] = *reinterpret_cast<const float*>(row + last_off);
  }
  template <bool FULL>
  __device__ __forceinline__ void issue_all(WaveStage<NB>& st, int s) const {
    static_for<8>([&](auto ec) {
      constexpr int E = decltype(ec)::value;
      const char* row = row_ptr<FULL, E>(st, s);
      static_for<NB>([&](auto bc) { load_one<decltype(bc)::value, E>(st, row); });
    });
    load_val<FULL>(st, s);
  }
  // ---- prefetch through LDS: global_load_lds_dword writes lane l's dword to (LDS pointer in M0) +
  // instruction offset + 4 l, so every (rating, feature block) gather of the wave lands as one 256-byte
  // chunk, in flight without holding registers.  Chunk k = e * NB + b at floats [64 k, 64 k + 64).
  template <bool FULL, bool SKIP_LAST = false>
  __device__ __forceinline__ void dma_issue(const WaveStage<NB>& st, lds_float_ptr lds, int s) const {
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    static_for<8>([&](auto ec) {
      constexpr int E = decltype(ec)::value;
      const char* row = row_ptr<FULL, E>(st, s);
      static_for<SKIP_LAST ? NB - 1 : NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        constexpr int k = E * NB + B;
        if constexpr (B + 1 < NB) {
          // the instruction offset (64 B: the block's byte offset in the row) moves BOTH addresses:
          // take it back out of the LDS pointer
          __builtin_amdgcn_global_load_lds((gptr)row, (lptr)(lds + 64 * k - 16 * B), 4, 64 * B, 0);
        } else {
          __builtin_amdgcn_global_load_lds((gptr)(row + last_off), (lptr)(lds + 64 * k), 4, 0, 0);
        }
      });
    });
  }
  // chunk -> registers (after s_waitcnt vmcnt(0))
  template <bool SKIP_LAST = false>
  __device__ __forceinline__ void dma_read(WaveStage<NB>& st, const float* lds_lane) const {
    static_for<8>([&](auto ec) {
      constexpr int E = decltype(ec)::value;
      static_for<SKIP_LAST ? NB - 1 : NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        st.raw[B][E] = lds_lane[64 * (E * NB + B)];
      });
    });
  }

  // after the loads have landed: the rating / the zero padding into rating E of the last block
  template <int E, int ARITH = kArithSplit3>
  __device__ __forceinline__ void finish_one(WaveStage<NB>& st) const {
    if constexpr (ARITH == kArithFast) {  // the table words are pre-split, the rating is split here
      float w = __builtin_bit_cast(float, fast_word(st.rv[E]));
      asm volatile("" : "+v"(w));  // computed by every lane, then ONE select (a conditional conversion compiles to exec-mask branches)
      st.raw[NB - 1][E] = is_feat ? st.raw[NB - 1][E] : w;
    } else if constexpr (ARITH == kArithSplitPk) {
      st.raw[NB - 1][E] = st.rv[E];  // no feature in the block: the rating (columns 0 .. 2) or zero
    } else
      st.raw[NB - 1][E] = is_feat ? st.raw[NB - 1][E] : st.rv[E];
  }
};

// Exact three-way split x = h + m + l of ratings 2 V, 2 V + 1 of feature block B.  x - h and
// (x - h) - m are exact in fp32 (h carries the leading 8 significand bits of x, m the next 8), and the
// last residual has at most 8 significant bits, so its conversion is exact too.  Cut into six
// micro-steps of 1-3 VALU instructions so that the pipeline can drop one behind every MFMA.
struct SplitState {
  float a, b, ra, rb, ta, tb;
  unsigned H, M;
};
template <int NB, int B, int V, int STEP, class PL>
__device__ __forceinline__ void split_micro(const WaveStage<NB>& st, PL& P, SplitState& x) {
  if constexpr (STEP == 0) {
    x.a = st.raw[B][2 * V];
    x.b = st.raw[B][2 * V + 1];
    x.H = pack_bf16(x.a, x.b);
  } else if constexpr (STEP == 1) {
    x.ra = sub_bf16_lo(x.a, x.H);
  } else if constexpr (STEP == 2) {
    x.rb = sub_bf16_hi(x.b, x.H);
  } else if constexpr (STEP == 3) {
    x.M = pack_bf16(x.ra, x.rb);
  } else if constexpr (STEP == 4) {
    x.ta = sub_bf16_lo(x.ra, x.M);
    x.tb = sub_bf16_hi(x.rb, x.M);
  } else {
    P.h[B][V] = x.H;
    P.m[B][V] = x.M;
    P.l[B][V] = pack_bf16(x.ta, x.tb);
  }
}
template <int NB, int B, int V, class PL>
__device__ __forceinline__ void split_pair(const WaveStage<NB>& st, PL& P) {
  SplitState x;
  static_for<6>([&](auto sc) { split_micro<NB, B, V, decltype(sc)::value>(st, P, x); });
}
// kArithFast: the stage registers hold (h, l) words; the operand words pair two ratings: two v_perm_b32
template <int NB, int B, int V>
__device__ __forceinline__ void split_pair(const WaveStage<NB>& st, Planes<NB, kArithFast>& P) {
  const unsigned wa = __builtin_bit_cast(unsigned, st.raw[B][2 * V]);
  const unsigned wb = __builtin_bit_cast(unsigned, st.raw[B][2 * V + 1]);
  P.h[B][V] = __builtin_amdgcn_perm(wb, wa, 0x05040100u);  // (h of rating 2 V, h of rating 2 V + 1)
  P.l[B][V] = __builtin_amdgcn_perm(wb, wa, 0x07060302u);
}

// n-th MFMA of a stage, n in [0, 6 NT): product n / NT of tile n % NT -- consecutive MFMAs hit
// different accumulators.  tile(I, J) += sum over the 32 ratings of theta[16 I + i] theta[16 J + j]
// as lh + hl + mm + mh + hm + hh (small terms first).
template <int ARITH>
constexpr int gram_products() { return ARITH == kArithFast ? 3 : 6; }
// kArithFast: lh + hl + hh
template <int NB, int N>
__device__ __forceinline__ void gram_mfma(const Planes<NB, kArithFast>& P, f32x4 (&acc)[NB * (NB + 1) / 2]) {
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int prod = N / NT, t = N % NT;
  constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
  if constexpr (prod == 0) acc[t] = mfma_f16(P.l[I], P.h[J], acc[t]);
  if constexpr (prod == 1) acc[t] = mfma_f16(P.h[I], P.l[J], acc[t]);
  if constexpr (prod == 2) acc[t] = mfma_f16(P.h[I], P.h[J], acc[t]);
}
template <int NB, int N>
__device__ __forceinline__ void gram_mfma(const Planes<NB>& P, f32x4 (&acc)[NB * (NB + 1) / 2]) {
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int prod = N / NT, t = N % NT;
  constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
  if constexpr (prod == 0) acc[t] = mfma_bf16(P.l[I], P.h[J], acc[t]);
  if constexpr (prod == 1) acc[t] = mfma_bf16(P.h[I], P.l[J], acc[t]);
  if constexpr (prod == 2) acc[t] = mfma_bf16(P.m[I], P.m[J], acc[t]);
  if constexpr (prod == 3) acc[t] = mfma_bf16(P.m[I], P.h[J], acc[t]);
  if constexpr (prod == 4) acc[t] = mfma_bf16(P.h[I], P.m[J], acc[t]);
  if constexpr (prod == 5) acc[t] = mfma_bf16(P.h[I], P.h[J], acc[t]);
}

// ----------------------------------------------------------------------------------
// Diagonal tiles at four products instead of six (round 5).  On tile (I, I) both operands come from the same feature
// block, so mh = (hm)^T and lh = (hl)^T: the tile accumulates D + 2 S with D = hh + mm (symmetric) and S = h m^T + h l^T
// -- two MFMAs whose A operand is the h plane with every exponent raised by one (2 h: one v_pk_add_u16 per register,
// exact) -- and ONCE per item, after the last stage, (T + T^T) / 2 = D + S + S^T restores the tile (wave_symmetrise_diag:
// a 16 x 16 transpose through the idle stage buffer).  2 NB of the 6 NT MFMAs of every stage go: 14 of 168 at f = 100.
// The doubled plane lives for three MFMAs: [2h l^T on (I, I)] [an independent tile] [2h m^T on (I, I)].
// Exponent + 1 on a bf16 zero gives 2^-126 -- multiplied by the m / l of a zero value, which are zero; an Inf / NaN still
// reaches the accumulator through hh.  Supported range: finite values with |x| < 2^127 whose leading term h is a
// NORMAL bf16 number or zero -- at biased exponent 0xFE the increment wraps into the Inf / NaN encodings, and a subnormal h is
// not doubled by it; there the diagonal tile's D + 2 S differs from the six-product form of the off-diagonal tiles (factors
// of that size do not occur in ALS: test_split_gram_adversarial_per_entry covers 1e-36 products and nine decades).
// ----------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 bf16x8_times2(u32x4 v) {
  u32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned w = v[e];  // through a scalar first: __builtin_bit_cast applied to a vector-element lvalue reads element 0
    r[e] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, w) + u16x2{0x0080, 0x0080}));
  }
  return r;
}
enum { kLH = 0, kHL = 1, kMM = 2, kMH = 3, kHM = 4, kHH = 5, kD2L = 6, kD2M = 7 };
template <int NB>
struct GramSched {
  static constexpr int NT = NB * (NB + 1) / 2;
  static constexpr int N = 6 * NT - 2 * NB;
  int tile[N], kind[N];
};
template <int NB>
__host__ __device__ constexpr GramSched<NB> make_gram_sched() {
  constexpr int NT = NB * (NB + 1) / 2, NOFF = NT - NB;
  GramSched<NB> s{};
  int off[NOFF > 0 ? NOFF : 1] = {}, dg[NB] = {};
  int no = 0, nd = 0;
  for (int t = 0; t < NT; ++t) {
    if (tile_I<NB>(t) == tile_J<NB>(t)) dg[nd++] = t; else off[no++] = t;
  }
  int n = 0;
  for (int k = 0; k < NOFF; ++k) { s.tile[n] = off[k]; s.kind[n++] = kLH; }
  for (int k = 0; k < NOFF; ++k) { s.tile[n] = off[k]; s.kind[n++] = kHL; }
  int used = 0;  // off-diagonal mm products spent as separators inside the diagonal triples
  for (int I = 0; I < NB; ++I) {
    s.tile[n] = dg[I]; s.kind[n++] = kD2L;
    if (used < NOFF) { s.tile[n] = off[used++]; s.kind[n++] = kMM; }
    s.tile[n] = dg[I]; s.kind[n++] = kD2M;
  }
  for (int k = used; k < NOFF; ++k) { s.tile[n] = off[k]; s.kind[n++] = kMM; }
  for (int I = 0; I < NB; ++I) { s.tile[n] = dg[I]; s.kind[n++] = kMM; }
  for (int k = 0; k < NOFF; ++k) { s.tile[n] = off[k]; s.kind[n++] = kMH; }
  for (int k = 0; k < NOFF; ++k) { s.tile[n] = off[k]; s.kind[n++] = kHM; }
  for (int t = 0; t < NT; ++t) { s.tile[n] = t; s.kind[n++] = kHH; }
  return s;
}
// h2: the doubled plane, two copies used in turn by successive feature blocks (a write straight behind the MFMA that reads a
// register measured safe; what the hardware does need is wait states between the v_pk_add_u16 that WRITES h2 and the MFMA
// that reads it -- two for dword 0 / 1 of the operand quad, one for dword 2 / 3: tools/probes/mfma_k32_hazard_probe.hip,
// profiles/r05/mfma_k32_operand_hazard.txt.  The compiler's hazard recogniser supplies them (s_nop); that it does is checked
// on the shipped ISA by tests/test_capi_symbols.py::test_k32_mfma_operand_wait_states_in_wave_kernels.)
template <int NB, int N>
__device__ __forceinline__ void gram_mfma_sched(const Planes<NB>& P, f32x4 (&acc)[NB * (NB + 1) / 2], u32x4 (&h2)[2]) {
  constexpr GramSched<NB> S = make_gram_sched<NB>();
  constexpr int t = S.tile[N], kind = S.kind[N];
  constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
  if constexpr (kind == kLH) acc[t] = mfma_bf16(P.l[I], P.h[J], acc[t]);
  if constexpr (kind == kHL) acc[t] = mfma_bf16(P.h[I], P.l[J], acc[t]);
  if constexpr (kind == kMM) acc[t] = mfma_bf16(P.m[I], P.m[J], acc[t]);
  if constexpr (kind == kMH) acc[t] = mfma_bf16(P.m[I], P.h[J], acc[t]);
  if constexpr (kind == kHM) acc[t] = mfma_bf16(P.h[I], P.m[J], acc[t]);
  if constexpr (kind == kHH) acc[t] = mfma_bf16(P.h[I], P.h[J], acc[t]);
  if constexpr (kind == kD2L) {
    h2[I & 1] = bf16x8_times2(P.h[I]);
    acc[t] = mfma_bf16(h2[I & 1], P.l[I], acc[t]);
  }
  if constexpr (kind == kD2M) acc[t] = mfma_bf16(h2[I & 1], P.m[I], acc[t]);
}
// (T + T^T) / 2 on the diagonal tiles, once per item: lane (g, c) register r holds T[4 g + r][c]; every tile goes through
// its own 16 x 17 window of the wave's stage buffer (idle: the last stage prefetches nothing; LDS operations of one wave
// execute in order).  A diagonal entry comes back as itself.
template <int NB>
__device__ __forceinline__ void wave_symmetrise_diag(f32x4 (&acc)[NB * (NB + 1) / 2], float* win, int lane) {
  const int c = lane & 15, g = lane >> 4;
  static_for<NB>([&](auto ic) {
    constexpr int I = decltype(ic)::value;
    constexpr int t = tile_of<NB>(I, I);
    float* w = win + I * 16 * 17;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[(4 * g + r) * 17 + c] = acc[t][r];
  });
  static_for<NB>([&](auto ic) {
    constexpr int I = decltype(ic)::value;
    constexpr int t = tile_of<NB>(I, I);
    const float* w = win + I * 16 * 17;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.5f * (acc[t][r] + w[c * 17 + 4 * g + r]);
  });
}

// ----------------------------------------------------------------------------------
// One stage, sized for TWO waves per SIMD (<= 256 registers per lane).  A wave cannot hide its own
// VALU work behind its own 16-cycle MFMAs (measured: stage time = MFMA time + VALU time with one
// wave per SIMD, whatever the interleave), so the overlap comes from the partner wave: this
// wave's split burst runs under the partner's MFMA phase and vice versa.  The gather latency
// (~4 000 cycles under load, measured) is covered by prefetching the NEXT stage through LDS
// (global_load_lds: no registers in flight) before the split + MFMAs of this one:
//   wait for the chunks of stage s -> registers -> issue the chunks of stage s + 1 -> split ->
//   6 NT MFMAs.
// (Program order, not issue order: the compiler's scheduler sinks most of the LDS-DMA instructions behind the MFMA burst.
// Round 5 pinned them in front of the split (sched_barrier) and interleaved them with it (sched_group_barrier): no
// gain either way -- X side 6.75-6.87 vs 6.69-6.85 ms, Theta side 10.75-11.06 vs 10.52-10.60, same box,
// profiles/r05/stage_variants_ab.txt -- the partner wave covers the gather latency; the tree keeps the compiler's order.)
// ----------------------------------------------------------------------------------
template <int NB, int PROD, int ARITH>
__device__ __forceinline__ void gram_product(const Planes<NB, ARITH>& P, f32x4 (&acc)[NB * (NB + 1) / 2]) {
  constexpr int NT = NB * (NB + 1) / 2;
  static_for<NT>([&](auto tc) { gram_mfma<NB, PROD * NT + decltype(tc)::value>(P, acc); });
}

enum { kStepPartial = 0, kStepFull = 1, kStepLast = 2 };  // prefetch of the step: clamped / select-free / none (last stage of the item)
template <int NB, int KIND, int ARITH>
__device__ __forceinline__ void stage_step(const WaveGather<NB>& wg, Planes<NB, ARITH>& P, WaveStage<NB>& R, lds_float_ptr lds,
                                           const float* lds_lane, f32x4 (&acc)[NB * (NB + 1) / 2], int s_next,
                                           int s_idx) {
  // in flight: chunks + ratings of the stage that is multiplied now, indices of stage s_next
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA chunks have landed
  wg.dma_read(R, lds_lane);
  static_for<8>([&](auto ec) { wg.template finish_one<decltype(ec)::value, ARITH>(R); });  // consumes R.rv
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the chunks are in registers, the buffer is free
  if constexpr (KIND == kStepFull) {
#if CUMF_ABLATE
    if (!(wg.dbg & 16))  // ablation: no gather DMA
#endif
      wg.template dma_issue<true>(R, lds, s_next);  // consumes R.idx
    wg.template load_val<true>(R, s_next);
    wg.template load_idx<true>(R, s_idx);
  } else if constexpr (KIND == kStepPartial) {
    // the last steps of an item: wave-uniform choices between the select-free and the clamped forms
    // (every step waits for vmcnt(0) anyway, so loads under a uniform branch cost nothing extra)
    const int nfull = wg.len / kWaveStage, nst = (wg.len + kWaveStage - 1) / kWaveStage;
    if (s_next < nfull) {
      wg.template dma_issue<true>(R, lds, s_next);
      wg.template load_val<true>(R, s_next);
    } else {
      wg.template dma_issue<false>(R, lds, s_next);
      wg.template load_val<false>(R, s_next);
    }
    if (s_idx < nfull)
      wg.template load_idx<true>(R, s_idx);
    else if (s_idx < nst)
      wg.template load_idx<false>(R, s_idx);
  }
  static_for<4 * NB>([&](auto uc) { split_pair<NB, decltype(uc)::value / 4, decltype(uc)::value % 4>(R, P); });
  if constexpr (ARITH == kArithSplit3) {
    u32x4 h2[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    static_for<GramSched<NB>::N>([&](auto nc) { gram_mfma_sched<NB, decltype(nc)::value>(P, acc, h2); });
  } else {
    static_for<gram_products<ARITH>()>([&](auto pc) { gram_product<NB, decltype(pc)::value>(P, acc); });
  }
}

// ----------------------------------------------------------------------------------
// kArithPre / kArithPrePk: the stage on a pre-split gather table (round 6; tools/probes/tr16_dma_probe.hip pins the two
// instructions it is built on).
//
// Table row (presplit_bf16x3_kernel), FB = f / 16 full feature blocks, SP = (f % 16) / 4 in {0, 1} strip pieces:
//   [h: 16 FB bf16][m: 16 FB bf16][l: 16 FB bf16] [strip, if SP: h, m, l of features 16 FB .. 16 FB + 3 (8 B each) + 8 B of zeros]
// LDS image of a 32-rating stage:
//   main    chunk (E, p), E = 0..7, p = plane: the plane of FOUR ratings rho = 4 E + q, q = 0..3, at RP bytes each, written
//           by ONE global_load_lds_dwordx4 (lane l = LP q + piece: 16 bytes -> chunk + 16 l; the lanes behind the 2 FB pieces
//           of a rating are masked off); 24 chunks instead of 56 dword gathers.  RP = 192 (64 for FB <= 2) and a 32-byte
//           skew per E pair put the eight 32-byte row pieces a transposing read touches per half wave into eight bank groups.
//   strip   [rho][h 8 B | m 8 B | l 8 B | pad 8 B] of the 32 ratings: one more 16-byte LDS-DMA (lane l: rating l / 2, half l % 2)
//   rating  the rating value rides in slot f (als.cu:750-757 fused into the Gram) and is no table entry: lane rho splits the
//           value of rating rho of the stage (loaded a stage ahead) and stores it once the stage's chunks have landed --
//           kArithPrePk: [r_h r_m r_l 0] as ONE 8-byte store into the strip's pad; kArithPre: [r_h 0 0 0 | r_m 0 0 0 | r_l 0 0 0]
//   zeros   24 bytes: what the lanes behind slot f read
// Operands: ds_read_b64_tr_b16 hands lane 4 a + b of a 16-lane group, as element j, halfword b of the 8-byte piece that lane
// 4 j + a addresses.  Lane (g, 4 j + a) addresses features 16 B + 4 a .. + 3 of rating rho = 8 g + 4 u + j: lane (g, c) receives
// feature 16 B + c of the ratings 8 g + 4 u + 0 .. 3 -- K slots 4 u .. 4 u + 3 of the MFMA, exactly the slots the in-kernel
// split gives them (P.h[B][2 u], [2 u + 1]).
//   kArithPre    every block like that, the last one from strip / rating pieces per plane: same operands in the same slots,
//                same MFMA sequence -- the accumulators are BIT-IDENTICAL to kArithSplit3's
//                (tests/test_gpu_parity.py::test_presplit_is_bit_identical); the verification form (cumf_set_presplit(2)).
//   kArithPrePk  the production form: the last feature block (f = 100: four features + the rating, 11 of 16 columns zero) is
//                read as ONE packed operand pk whose columns are [h of the strip features | m | l | r_h r_m r_l 0]: tile
//                (I, NB - 1) takes three products h_I pk + m_I pk + l_I pk (all nine plane products at once) instead of six,
//                tile (NB - 1, NB - 1) one (pk pk^T) instead of four -- 133 MFMAs per stage instead of 154 at f = 100 -- and
//                once per item the column groups are folded back (wave_fold_strip).  Error class of kArithSplit3 (the three
//                dropped products ml, lm, ll are now included), not its bits in the last block column.
// ----------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;
typedef char __attribute__((address_space(3)))* lds_byte_ptr;
template <int NB>
struct PreGeo {
  static constexpr int FB = NB - 1;
  static constexpr int RP = FB <= 2 ? 64 : 192;       // bytes of one rating inside a chunk (>= 32 FB)
  static constexpr int LP = RP / 16;                  // DMA lanes per rating, 2 FB of them fetch
  static constexpr int CS = 4 * RP;                   // one chunk: a plane of four ratings
  static constexpr int kMain = 24 * CS + 3 * 32;      // + the skews: chunk (E, p) at (3 E + p) CS + 32 (E >> 1)
  static constexpr int kStrip = kMain;                // 32 ratings x 32 B
  static constexpr int kZero = kStrip + 1024;         // 24 B (32 reserved)
  static constexpr int kRating = kZero + 32;          // kArithPre only: 32 ratings x 24 B
  __host__ __device__ static constexpr int bytes(bool packed) { return packed ? kRating : kRating + 768; }
  static_assert(32 * FB <= RP && 4 * LP <= 64, "a rating's plane fits its slot, four ratings fit the wave");
  __host__ __device__ static constexpr int chunk(int E, int p) { return (3 * E + p) * CS + 32 * (E >> 1); }
};

template <int NB>
struct PreStage {
  int idx[8];   // column indices of the ratings 4 E + q of a stage whose main chunks are still to be issued (lanes of DMA group q)
  int sidx;     // ... of rating lane / 2 (strip)
  float rv;     // rating value of rating lane & 31 of the stage whose chunks are in flight
};
template <int NB>
struct Planes<NB, kArithPrePk> {
  u32x4 h[NB], m[NB], l[NB];  // blocks 0 .. NB - 2
  u32x4 pk;                   // the last block, packed
};
template <int NB>
struct Planes<NB, kArithSplitPk> : Planes<NB, kArithPrePk> {};

template <int NB, bool PK>
struct PreGather {
  using G = PreGeo<NB>;
  const char* lane_base;   // table + 16 piece
  const char* zero_base;   // zero row + 16 piece
  const char* strip_base;  // table + 96 FB + 16 (lane & 1)
  const char* strip_zero;
  const int* ib;           // colidx + begin (the zero row for an item without ratings)
  const float* vb;         // val + begin (the zero row without ratings / values)
  lds_tr_ptr tr_main;      // lane part of the addresses of the transposing reads of blocks 0 .. FB - 1
  lds_tr_ptr tr_last[2];   // ... of the last block, per quad u
  unsigned pitch;
  int len, q, lane;
#if CUMF_ABLATE_STAGE
  int dbg;
#endif
  bool dma_active, sp;

  __device__ __forceinline__ void init(const KernelArgs& a, int f, long long begin, int len_, int lane_, float* smem) {
    lane = lane_;
    len = len_;
    pitch = a.pre_pitch;
#if CUMF_ABLATE
    if (a.dbg & 8) pitch = 0u;  // profiling build: 8 = every gather hits row 0
#endif
#if CUMF_ABLATE_STAGE
    dbg = a.dbg;
#endif
    sp = ((f & 15) >> 2) != 0;
    q = lane / G::LP;
    const int piece = lane % G::LP;
    dma_active = q < 4 && piece < 2 * G::FB;
    q = q < 4 ? q : 3;
    lane_base = reinterpret_cast<const char*>(a.gather) + 16 * piece;
    zero_base = reinterpret_cast<const char*>(g_wave_zeros) + 16 * piece;
    strip_base = reinterpret_cast<const char*>(a.gather) + 96 * G::FB + 16 * (lane & 1);
    strip_zero = reinterpret_cast<const char*>(g_wave_zeros) + 16 * (lane & 1);
    ib = len_ > 0 ? a.colidx + begin : reinterpret_cast<const int*>(g_wave_zeros);
    vb = (len_ > 0 && a.val != nullptr) ? a.val + begin : g_wave_zeros;
    const int g = lane >> 4, j = (lane >> 2) & 3, aa = lane & 3;
    lds_byte_ptr base = (lds_byte_ptr)smem;
    tr_main = (lds_tr_ptr)(base + 6 * g * G::CS + 32 * g + G::RP * j + 8 * aa);
    const int spn = sp ? 1 : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rho = 8 * g + 4 * u + j;
      int off;
      if constexpr (PK)  // columns [h feats | m feats | l feats | rating]: the strip's pieces in order, the rating piece in its pad
        off = (sp || aa == 0) ? G::kStrip + 32 * rho + (sp ? 8 * aa : 24) : G::kZero;
      else
        off = aa < spn ? G::kStrip + 32 * rho + 8 * aa : (aa == spn ? G::kRating + 24 * rho : G::kZero);
      tr_last[u] = (lds_tr_ptr)(base + off);
    }
    // the zero pieces (and the rating pieces' zero halfwords), once (LDS operations of one wave execute in order)
    constexpr int kClear = (G::bytes(PK) - G::kZero) / 16;
    if (lane < kClear) reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::kZero)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // column indices of stage s (FULL: every rating of the stage exists)
  template <bool FULL>
  __device__ __forceinline__ void load_idx(PreStage<NB>& st, int s) const {
    const int top = len > 0 ? len - 1 : 0;
    if constexpr (FULL) {
      const int* p = ib + kWaveStage * s + q;
#pragma unroll
      for (int E = 0; E < 8; ++E) st.idx[E] = p[4 * E];
      st.sidx = ib[kWaveStage * s + (lane >> 1)];
    } else {
#pragma unroll
      for (int E = 0; E < 8; ++E) {
        const int pos = kWaveStage * s + 4 * E + q;
        st.idx[E] = ib[pos < top ? pos : top];
      }
      const int ps = kWaveStage * s + (lane >> 1);
      st.sidx = ib[ps < top ? ps : top];
    }
  }
  // rating value of stage s; ratings past the end of the item: zero rows AND a zero rating (sum r^2 of the fused SSE)
  template <bool FULL>
  __device__ __forceinline__ void load_rv(PreStage<NB>& st, int s) const {
    const int pv = kWaveStage * s + (lane & 31);
    if constexpr (FULL) {
      st.rv = vb[pv];
    } else {
      const int top = len > 0 ? len - 1 : 0;
      const float v = vb[pv < top ? pv : top];
      st.rv = pv < len ? v : 0.f;
    }
  }

  // the rating values of the stage that has just landed (st.rv) as three bf16 terms into its rating pieces
  __device__ __forceinline__ void put_rating(const PreStage<NB>& st, float* smem) const {
    unsigned H, M, L;
    split3_pair(st.rv, 0.f, H, M, L);
    if (lane < 32) {
      if constexpr (PK) {
        u32x2 w = {(H & 0xffffu) | (M << 16), L & 0xffffu};
        *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(smem) + G::kStrip + 32 * lane + 24) = w;
      } else {
        unsigned short* rp = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(smem) + G::kRating + 24 * lane);
        rp[0] = (unsigned short)H;
        rp[4] = (unsigned short)M;
        rp[8] = (unsigned short)L;
      }
    }
  }

  template <bool FULL>
  __device__ __forceinline__ void dma_issue(const PreStage<NB>& st, float* smem, int s) const {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc rejects the 16-byte form of the builtin: it checks it against the host target)
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    lds_byte_ptr lds = (lds_byte_ptr)smem;
    if (dma_active) {
      static_for<8>([&](auto ec) {
        constexpr int E = decltype(ec)::value;
        const char* row = lane_base + (unsigned long long)(unsigned)st.idx[E] * pitch;  // v_mad_u64_u32
        if constexpr (!FULL) row = (kWaveStage * s + 4 * E + q < len) ? row : zero_base;
        static_for<3>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          // the instruction offset (the plane's byte offset in the row) moves BOTH addresses: taken back out of the LDS pointer
          __builtin_amdgcn_global_load_lds((gptr)row, (lptr)(lds + G::chunk(E, p) - 32 * G::FB * p), 16, 32 * G::FB * p, 0);
        });
      });
    }
    if (sp) {  // wave-uniform
      const char* row = strip_base + (unsigned long long)(unsigned)st.sidx * pitch;
      if constexpr (!FULL) row = (kWaveStage * s + (lane >> 1) < len) ? row : strip_zero;
      __builtin_amdgcn_global_load_lds((gptr)row, (lptr)(lds + G::kStrip), 16, 0, 0);
    }
#endif
  }

  // the landed stage -> MFMA operands
  static __device__ __forceinline__ u32x2 tr_read(lds_tr_ptr p, int byte_off) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)((lds_byte_ptr)p + byte_off)));
  }
  // the planes of the full blocks B0 .. B1 - 1
  template <int B0, int B1, class PL>
  __device__ __forceinline__ void read_blocks(PL& P) const {
    static_for<B1 - B0>([&](auto bc) {
      constexpr int B = B0 + decltype(bc)::value;
      static_for<2>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const u32x2 vh = tr_read(tr_main, (3 * u + 0) * G::CS + 32 * B);
        const u32x2 vm = tr_read(tr_main, (3 * u + 1) * G::CS + 32 * B);
        const u32x2 vl = tr_read(tr_main, (3 * u + 2) * G::CS + 32 * B);
        P.h[B][2 * u] = vh[0], P.h[B][2 * u + 1] = vh[1];
        P.m[B][2 * u] = vm[0], P.m[B][2 * u + 1] = vm[1];
        P.l[B][2 * u] = vl[0], P.l[B][2 * u + 1] = vl[1];
      });
    });
  }
  template <class PL>
  __device__ __forceinline__ void read_pk(PL& P) const {  // the packed last block (kArithPrePk)
    const u32x2 v0 = tr_read(tr_last[0], 0), v1 = tr_read(tr_last[1], 0);
    P.pk = u32x4{v0[0], v0[1], v1[0], v1[1]};
  }
  template <class PL>
  __device__ __forceinline__ void read(PL& P) const {
    if constexpr (PK) read_pk(P);
    static_for<G::FB>([&](auto bc) {
      constexpr int B = decltype(bc)::value;
      static_for<2>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const u32x2 vh = tr_read(tr_main, (3 * u + 0) * G::CS + 32 * B);
        const u32x2 vm = tr_read(tr_main, (3 * u + 1) * G::CS + 32 * B);
        const u32x2 vl = tr_read(tr_main, (3 * u + 2) * G::CS + 32 * B);
        P.h[B][2 * u] = vh[0], P.h[B][2 * u + 1] = vh[1];
        P.m[B][2 * u] = vm[0], P.m[B][2 * u + 1] = vm[1];
        P.l[B][2 * u] = vl[0], P.l[B][2 * u + 1] = vl[1];
      });
    });
    // the last block (no generic lambda here: the form not taken must be discarded, not just skipped)
    if constexpr (!PK) {
      constexpr int B = G::FB;
      const u32x2 h0 = tr_read(tr_last[0], 0), m0 = tr_read(tr_last[0], 8), l0 = tr_read(tr_last[0], 16);
      const u32x2 h1 = tr_read(tr_last[1], 0), m1 = tr_read(tr_last[1], 8), l1 = tr_read(tr_last[1], 16);
      P.h[B] = u32x4{h0[0], h0[1], h1[0], h1[1]};
      P.m[B] = u32x4{m0[0], m0[1], m1[0], m1[1]};
      P.l[B] = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
  }
};

// MFMA schedule of kArithPrePk: as make_gram_sched on the full blocks (six products, four on their diagonal tiles); the last
// block column three products against the packed operand (kPL, kPM, kPH: small terms first), its diagonal tile one (kPP).
// In TWO groups: the tiles among the first HB feature blocks (+ their strip tiles) first -- their operands are the first
// transposing reads to return, so these MFMAs run while the reads of the other blocks are still in flight (the waits are
// the compiler's, per operand); the LDS-DMA of the next stage is issued between the groups, behind the last read.  The
// order of the products of any ONE tile is the same in both groups and as in make_gram_sched.
enum { kPL = 8, kPM = 9, kPH = 10, kPP = 11 };
template <int NB>
struct GramSchedPk {
  static constexpr int FB = NB - 1, NTF = FB * (FB + 1) / 2;
  static constexpr int N = 6 * NTF - 2 * FB + 3 * FB + 1;
  static constexpr int HB = FB >= 4 ? FB / 2 : FB;  // blocks of the first group (all of them for small systems)
  int tile[N], kind[N];
  int n1;  // MFMAs of the first group
};
template <int NB>
__host__ __device__ constexpr GramSchedPk<NB> make_gram_sched_pk() {
  constexpr int FB = NB - 1, HB = GramSchedPk<NB>::HB;
  GramSchedPk<NB> s{};
  int n = 0;
  for (int grp = 0; grp < 2; ++grp) {
    // tiles of this group: (I, J) with max(I, J) < HB (group 0) or >= HB (group 1); strip tile I with I < HB / >= HB
    auto in = [&](int I, int J) { return ((I > J ? I : J) < HB) == (grp == 0); };
    int off[FB * FB + 1] = {}, dg[FB + 1] = {}, st[FB + 1] = {};
    int no = 0, nd = 0, ns = 0;
    for (int I = 0; I < FB; ++I) {
      if (in(I, I)) { dg[nd++] = I; st[ns++] = tile_of<NB>(I, NB - 1); }
      for (int J = I + 1; J < FB; ++J)
        if (in(I, J)) off[no++] = tile_of<NB>(I, J);
    }
    for (int k = 0; k < no; ++k) { s.tile[n] = off[k]; s.kind[n++] = kLH; }
    for (int k = 0; k < ns; ++k) { s.tile[n] = st[k]; s.kind[n++] = kPL; }
    for (int k = 0; k < no; ++k) { s.tile[n] = off[k]; s.kind[n++] = kHL; }
    int used = 0;  // separators inside the diagonal triples: off-diagonal mm products, then the strip's m products
    for (int k = 0; k < nd; ++k) {
      s.tile[n] = tile_of<NB>(dg[k], dg[k]); s.kind[n++] = kD2L;
      if (used < no) { s.tile[n] = off[used]; s.kind[n++] = kMM; }
      else if (used - no < ns) { s.tile[n] = st[used - no]; s.kind[n++] = kPM; }
      ++used;
      s.tile[n] = tile_of<NB>(dg[k], dg[k]); s.kind[n++] = kD2M;
    }
    for (int k = used; k < no; ++k) { s.tile[n] = off[k]; s.kind[n++] = kMM; }
    for (int k = (used > no ? used - no : 0); k < ns; ++k) { s.tile[n] = st[k]; s.kind[n++] = kPM; }
    for (int k = 0; k < nd; ++k) { s.tile[n] = tile_of<NB>(dg[k], dg[k]); s.kind[n++] = kMM; }
    for (int k = 0; k < no; ++k) { s.tile[n] = off[k]; s.kind[n++] = kMH; }
    for (int k = 0; k < no; ++k) { s.tile[n] = off[k]; s.kind[n++] = kHM; }
    for (int k = 0; k < ns; ++k) { s.tile[n] = st[k]; s.kind[n++] = kPH; }
    for (int k = 0; k < nd; ++k) { s.tile[n] = tile_of<NB>(dg[k], dg[k]); s.kind[n++] = kHH; }
    for (int k = 0; k < no; ++k) { s.tile[n] = off[k]; s.kind[n++] = kHH; }
    if (grp == 0) {
      s.tile[n] = tile_of<NB>(NB - 1, NB - 1); s.kind[n++] = kPP;
      s.n1 = n;
    }
  }
  return s;
}
template <int NB, int N, class PL>
__device__ __forceinline__ void gram_mfma_sched_pk(const PL& P, f32x4 (&acc)[NB * (NB + 1) / 2], u32x4 (&h2)[2]) {
  constexpr GramSchedPk<NB> S = make_gram_sched_pk<NB>();
  constexpr int t = S.tile[N], kind = S.kind[N];
  constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
  if constexpr (kind == kLH) acc[t] = mfma_bf16(P.l[I], P.h[J], acc[t]);
  if constexpr (kind == kHL) acc[t] = mfma_bf16(P.h[I], P.l[J], acc[t]);
  if constexpr (kind == kMM) acc[t] = mfma_bf16(P.m[I], P.m[J], acc[t]);
  if constexpr (kind == kMH) acc[t] = mfma_bf16(P.m[I], P.h[J], acc[t]);
  if constexpr (kind == kHM) acc[t] = mfma_bf16(P.h[I], P.m[J], acc[t]);
  if constexpr (kind == kHH) acc[t] = mfma_bf16(P.h[I], P.h[J], acc[t]);
  if constexpr (kind == kD2L) {
    h2[I & 1] = bf16x8_times2(P.h[I]);
    acc[t] = mfma_bf16(h2[I & 1], P.l[I], acc[t]);
  }
  if constexpr (kind == kD2M) acc[t] = mfma_bf16(h2[I & 1], P.m[I], acc[t]);
  if constexpr (kind == kPL) acc[t] = mfma_bf16(P.l[I], P.pk, acc[t]);
  if constexpr (kind == kPM) acc[t] = mfma_bf16(P.m[I], P.pk, acc[t]);
  if constexpr (kind == kPH) acc[t] = mfma_bf16(P.h[I], P.pk, acc[t]);
  if constexpr (kind == kPP) acc[t] = mfma_bf16(P.pk, P.pk, acc[t]);
}

// kArithPrePk, once per item behind the last stage: the packed column groups of the last block column back into the
// layout every consumer expects (columns 0 .. 4 SP - 1: the strip's features, column 4 SP: the right-hand side, zeros
// behind).  With n = 4 SP:  G[.][j] = S[.][j] + S[.][n + j] + S[.][2 n + j],  G[.][n] = S[.][3 n] + S[.][3 n + 1] + S[.][3 n + 2];
// the diagonal tile (NB - 1, NB - 1) = pk pk^T folds its rows the same way (rows 4 g + r: plane g of feature r, g = 3: the
// rating's three terms) -- two ds_bpermute per register + one for the rating row.
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) {  // lane c of every 16-lane row reads lane c + N of its row (0 past its end)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float fold_strip_cols(float v, bool sp, int c) {
  const float u = (v + dpp_row_shl<1>(v)) + dpp_row_shl<2>(v);  // lane 3 n: the rating's three terms
  if (sp) {  // wave-uniform
    const float a = (v + dpp_row_shl<4>(v)) + dpp_row_shl<8>(v);
    const float r = dpp_row_shl<8>(u);                          // lane 4 <- lane 12
    return c < 4 ? a : (c == 4 ? r : 0.f);
  }
  return c == 0 ? u : 0.f;
}
template <int NB>
__device__ __forceinline__ void wave_fold_strip(f32x4 (&acc)[NB * (NB + 1) / 2], bool sp, int lane) {
  const int c = lane & 15, g = lane >> 4;
  auto bperm = [](int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
  };
  static_for<NB>([&](auto ic) {
    constexpr int t = tile_of<NB>(decltype(ic)::value, NB - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = fold_strip_cols(acc[t][r], sp, c);
  });
  constexpr int t = tile_of<NB>(NB - 1, NB - 1);
  float w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) w[r] = acc[t][r];
  const float rt = (w[0] + w[1]) + w[2];  // lane group 3 (SP) / 0 (no strip): the rating row's three terms
  if (sp) {  // wave-uniform
    const float y = bperm(4 * ((lane + 32) & 63), rt);  // lane group 1 <- lane group 3
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x1 = bperm(4 * ((lane + 16) & 63), w[r]), x2 = bperm(4 * ((lane + 32) & 63), w[r]);
      const float rows = (w[r] + x1) + x2;  // lane group 0: planes h + m + l of feature row r
      acc[t][r] = g == 0 ? rows : ((g == 1 && r == 0) ? y : 0.f);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = (g == 0 && r == 0) ? rt : 0.f;
  }
}

// kArithSplitPk: stage_step (the in-kernel split) without the last block's gathers and with its packed operand.
template <int NB, int KIND>
__device__ __forceinline__ void stage_step_splitpk(const WaveGather<NB>& wg, Planes<NB, kArithSplitPk>& P, WaveStage<NB>& R,
                                                   lds_float_ptr lds, const float* lds_lane, f32x4 (&acc)[NB * (NB + 1) / 2],
                                                   int s_next, int s_idx, int c) {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA chunks have landed
  wg.template dma_read<true>(R, lds_lane);
  static_for<8>([&](auto ec) { wg.template finish_one<decltype(ec)::value, kArithSplitPk>(R); });  // consumes R.rv
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the chunks are in registers, the buffer is free
  if constexpr (KIND == kStepFull) {
    wg.template dma_issue<true, true>(R, lds, s_next);  // consumes R.idx
    wg.template load_val<true>(R, s_next);
    wg.template load_idx<true>(R, s_idx);
  } else if constexpr (KIND == kStepPartial) {
    const int nfull = wg.len / kWaveStage, nst = (wg.len + kWaveStage - 1) / kWaveStage;
    if (s_next < nfull) {
      wg.template dma_issue<true, true>(R, lds, s_next);
      wg.template load_val<true>(R, s_next);
    } else {
      wg.template dma_issue<false, true>(R, lds, s_next);
      wg.template load_val<false>(R, s_next);
    }
    if (s_idx < nfull)
      wg.template load_idx<true>(R, s_idx);
    else if (s_idx < nst)
      wg.template load_idx<false>(R, s_idx);
  }
  static_for<4 * (NB - 1)>([&](auto uc) { split_pair<NB, decltype(uc)::value / 4, decltype(uc)::value % 4>(R, P); });
  // the rating: lanes of column 0 keep its h, column 1 its m, column 2 its l (the other columns hold zeros in all three)
  const bool c1 = c == 1, c2 = c == 2;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    unsigned H, M, L;
    split3_pair(R.raw[NB - 1][2 * v], R.raw[NB - 1][2 * v + 1], H, M, L);
    P.pk[v] = c2 ? L : (c1 ? M : H);
  }
  u32x4 h2[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  static_for<GramSchedPk<NB>::N>([&](auto nc) { gram_mfma_sched_pk<NB, decltype(nc)::value>(P, acc, h2); });
}

// One stage: wait for the chunks -> rating pieces -> 6 NB transposing reads -> chunks of the next stage, indices of the one
// after, rating values of the next -> MFMAs.  At entry R holds the indices of stage s_next and the rating values of this one.
template <int NB, int KIND, bool PK, class PL>
__device__ __forceinline__ void stage_step_pre(const PreGather<NB, PK>& wg, PL& P, PreStage<NB>& R, float* smem,
                                               f32x4 (&acc)[NB * (NB + 1) / 2], int s_next, int s_load) {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the chunks of this stage have landed, R is complete
  wg.put_rating(R, smem);
  wg.read(P);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the operands are in registers, the image is free
  if constexpr (KIND == kStepFull) {
    wg.template dma_issue<true>(R, smem, s_next);
    wg.template load_idx<true>(R, s_load);
    wg.template load_rv<true>(R, s_next);
  } else if constexpr (KIND == kStepPartial) {
    const int nfull = wg.len / kWaveStage, nst = (wg.len + kWaveStage - 1) / kWaveStage;
    if (s_next < nfull) {
      wg.template dma_issue<true>(R, smem, s_next);
      wg.template load_rv<true>(R, s_next);
    } else {
      wg.template dma_issue<false>(R, smem, s_next);
      wg.template load_rv<false>(R, s_next);
    }
    if (s_load < nfull)
      wg.template load_idx<true>(R, s_load);
    else if (s_load < nst)
      wg.template load_idx<false>(R, s_load);
  }
  u32x4 h2[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  static_for<GramSched<NB>::N>([&](auto nc) { gram_mfma_sched<NB, decltype(nc)::value>(P, acc, h2); });
}

// kArithPrePk: the same step with the MFMAs in two groups around the prefetch (make_gram_sched_pk) -- the first group needs
// only the operands of the first blocks and runs under the transposing reads of the others.
template <int NB, int KIND>
__device__ __forceinline__ void stage_step_pk(const PreGather<NB, true>& wg, Planes<NB, kArithPrePk>& P, PreStage<NB>& R,
                                              float* smem, f32x4 (&acc)[NB * (NB + 1) / 2], int s_next, int s_load) {
  constexpr GramSchedPk<NB> S = make_gram_sched_pk<NB>();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the chunks of this stage have landed, R is complete
  wg.put_rating(R, smem);
  constexpr int FB = NB - 1, HB = GramSchedPk<NB>::HB, kLate = 6 * (FB - HB);  // transposing reads of the second group's blocks
  u32x4 h2[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#if CUMF_ABLATE_STAGE  // one-off timing builds only (tools/wave_variants.sh EXTRA="-DCUMF_ABLATE=1 -DCUMF_ABLATE_STAGE=1"): a
  // switch inside the stage changes its scheduling regions, so the profiling build proper has none here
  if (wg.dbg & 32) {  // 32 = no transposing reads (the operands keep the first stage's values)
    static_for<S.n1>([&](auto nc) { gram_mfma_sched_pk<NB, decltype(nc)::value>(P, acc, h2); });
  } else
#endif
  {
    wg.read_pk(P);
    wg.template read_blocks<0, HB>(P);
    __builtin_amdgcn_sched_barrier(0);
    // One region: the reads of the other blocks go out ONE BEHIND EACH of the first MFMAs (at most 16 LDS operations are in
    // flight per wave -- a burst of reads in front of the MFMAs would hold the wave until all but 16 have returned)
    wg.template read_blocks<HB, FB>(P);
    static_for<S.n1>([&](auto nc) { gram_mfma_sched_pk<NB, decltype(nc)::value>(P, acc, h2); });
    static_for<(kLate < S.n1 ? kLate : S.n1)>([&](auto) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
    });
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): every operand is in registers, the image is free
  if constexpr (KIND == kStepFull) {
#if CUMF_ABLATE_STAGE
    if (!(wg.dbg & 16))  // 16 = no LDS-DMA in the steady state
#endif
    wg.template dma_issue<true>(R, smem, s_next);
    wg.template load_idx<true>(R, s_load);
    wg.template load_rv<true>(R, s_next);
  } else if constexpr (KIND == kStepPartial) {
    const int nfull = wg.len / kWaveStage, nst = (wg.len + kWaveStage - 1) / kWaveStage;
    if (s_next < nfull) {
      wg.template dma_issue<true>(R, smem, s_next);
      wg.template load_rv<true>(R, s_next);
    } else {
      wg.template dma_issue<false>(R, smem, s_next);
      wg.template load_rv<false>(R, s_next);
    }
    if (s_load < nfull)
      wg.template load_idx<true>(R, s_load);
    else if (s_load < nst)
      wg.template load_idx<false>(R, s_load);
  }
  static_for<GramSchedPk<NB>::N - S.n1>([&](auto nc) { gram_mfma_sched_pk<NB, S.n1 + decltype(nc)::value>(P, acc, h2); });
}

// ----------------------------------------------------------------------------------
// Epilogues on the full tile set of one wave (same element layout as als_kernels.hip).
// ----------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void wave_tiles_to_partial(const f32x4 (&acc)[NB * (NB + 1) / 2], float* __restrict__ part,
                                                      int lane) {
  static_for<NB*(NB + 1) / 2>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r) part[((size_t)t * 4 + r) * 64 + lane] = acc[t][r];
  });
}

// row-major f x f Gram, both triangles, lambda * n on the diagonal (als.cu:545-566) + RHS
template <int NB, typename T>
__device__ __forceinline__ void wave_tiles_to_global(const f32x4 (&acc)[NB * (NB + 1) / 2], T* __restrict__ tt,
                                                     float* __restrict__ rhs, int f, float reg, int lane,
                                                     bool packed = false) {
  const int c = lane & 15, kk = lane >> 4;
  static_for<NB*(NB + 1) / 2>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * I + 4 * kk + r, j = 16 * J + c;
      float v = acc[t][r];
      if (i < f && j < f) {
        if (i == j) v += reg;
        // both triangles from ONE accumulator entry (als.h:39-143 writes tt[i][j] and tt[j][i] from the same
        // temp): inside a diagonal tile the split products reach (i, j) and (j, i) in different orders, so
        // only the upper entry is used there
        if (I != J || i <= j) {
          if (packed) {  // row i keeps columns i .. f - 1 (cumf_get_hermitian_packed)
            tt[(size_t)i * f - (size_t)(i * (i - 1) / 2) + (j - i)] = (T)v;
          } else {
            tt[(size_t)i * f + j] = (T)v;  // T = _Float16: fp16 Gram storage (als.cu:335-441), round to nearest even
            if (i != j) tt[(size_t)j * f + i] = (T)v;
          }
        }
      } else if (i < f && j == f && rhs != nullptr) {
        rhs[i] = v;
      }
    }
  });
}

// ----------------------------------------------------------------------------------
// Back substitution U x = y straight from the accumulators of one wave, through a small LDS
// window (16 NB rows x 17 floats = 7.6 KB at NB = 7 instead of the 29 KB packed row store, so
// that eight waves fit a CU).  After the elimination tile (I, J), I <= J, holds U (rows above
// and on the diagonal) in the C/D layout and column f holds y.  Same recurrence as
// back_substitute_zeroed (als_device.h): lane i owns rows i, i + 64, ...; row i is scaled by
// 1 / u_ii (z_i = y_i / u_ii, v_ik = u_ik / u_ii), x_k = z_k; per 16-pivot block column kb the
// tiles (0..kb, kb) are written to the window (entries at and left of the diagonal as zeros) and
// every lane reads the 16 entries of its rows in that block column, one block ahead of their use
// (LDS operations of one wave execute in order: the window is rewritten behind the reads).
// ----------------------------------------------------------------------------------
constexpr int kBsPitch = 17;
// window + pivot reciprocals + 16 zeros + dummy line, then (128-byte aligned) the SIDE ROWS of lu_wave_blocked: rows 12 .. 15 of
// every tile above the last block row, as rows of W.  The panel-row exchange of lu_prep_step_s (4 lane groups x NB blocks x 16
// columns x 4 rows = 256 NB floats) aliases the window (272 NB floats): the window is written by the back substitution only,
// the exchange is dead by then.
// Side store: tile t = (I, J) in a slot of kSideSlot = 80 floats, row 12 + k, column c at 80 t + skew(I) + 17 k + c with
// skew(I) = 32 I + 12 + 16 (tile_of(I, I) & 1) (a tile's rows reach 14 floats into the next slot: the 32 I keep block rows
// with different skews apart).  ds_read_b32 serves lanes 0 .. 31 / 32 .. 63 in one cycle each when their banks
// (dword address mod 32) differ; the lane of row i reads the window at 17 i + j -- bank 17 i + j -- and the banks of the lanes
// of rows 12 .. 15 (mod 16) are what the side rows must take over: 80 t + skew = 16 (I + J) + 12 (mod 32), so row 12 + k of
// block I sits on bank 12 + 17 k + 16 I + 16 J + j -- its window bank for even J, that of its partner lane (row + 16: the other
// side lane of the group) for odd J.  A first layout with 16-float rows (banks j and j + 16 only: five lanes per bank) cost
// more LDS cycles than the MFMAs it replaced.
constexpr int kSideSlot = 80;
template <int NB>
__host__ __device__ constexpr int wave_lu_side_offset(int f) {
  return (16 * NB * kBsPitch + ((f + 3) & ~3) + 16 + 64 + 31) & ~31;
}
template <int NB>
__host__ __device__ constexpr int wave_lu_side_skew(int I) { return 32 * I + 12 + 16 * (tile_of<NB>(I, I) & 1); }
template <int NB>
__host__ __device__ constexpr int wave_lu_lds_floats(int f) {
  return wave_lu_side_offset<NB>(f) + kSideSlot * (NB * (NB + 1) / 2) + 32 * NB + 32;
}
template <int NB, int ARITH = kArithSplit3>
__host__ __device__ constexpr int wave_stage_lds_floats() {
  if constexpr (ARITH == kArithPre || ARITH == kArithPrePk)
    return PreGeo<NB>::bytes(ARITH == kArithPrePk) / 4;  // the pre-split image of a stage
  else
    return 64 * 8 * NB;             // 8 NB chunks of 64 floats
}

// Round 6 (side rows): rows 12 .. 15 of the blocks above the last block row are not in the accumulators as rows of -U -- the
// fourth panel of a block row skips its fp32 MFMAs -- but in `side` as rows of W (scaled by 1 / sqrt(u_kk), rdiag holds the
// matching reciprocal): the lanes of those rows read tile (I, kb) of the side store, one slot further per block column,
// instead of the window (layout and banks: wave_lu_side_offset).
template <int NB, int NQ>
__device__ __forceinline__ float back_substitute_tiles(const f32x4 (&acc)[NB * (NB + 1) / 2], float* T,
                                                      const float* rdiag, const float* zpad, const float* side, int f,
                                                      float* __restrict__ x_global, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const int top = f - 1;
  // block column kb -> window
  auto dump = [&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    static_for<kb + 1>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int t = tile_of<NB>(I, kb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[t][r];
        if constexpr (I == kb) v = (c > 4 * g + r) ? v : 0.f;
        T[(16 * I + 4 * g + r) * kBsPitch + c] = v;
      }
    });
  };
  float z[NQ], rdl[NQ];
  const float* rowp[NQ];  // this lane's row in block column kb (walks down with kb for the side rows)
  int step[NQ];           // floats per block column: kSideSlot for a side row, 0 for a row of the window
  int ib[NQ];  // block of this lane's row
  static_for<NQ>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int i = lane + 64 * q;
    const int ic = i < f ? i : f - 1;
    const int I = ic >> 4;
    ib[q] = i < f ? I : 1 << 20;  // rows past f never take part
    const bool srow = I < NB - 1 && (ic & 15) >= 12;
    const int tII = I * NB - I * (I - 1) / 2;  // tile_of(I, I); tile_of(I, kb) = tII + kb - I
    const float* sp = side + (tII + (NB - 1) - I) * kSideSlot + 32 * I + 12 + 16 * (tII & 1) + 17 * (ic & 3);
    rowp[q] = srow ? sp : T + ic * kBsPitch;
    step[q] = srow ? kSideSlot : 0;
    rdl[q] = i < f ? rdiag[ic] : 0.f;
  });
  float col[2][16][NQ];
  auto issue = [&](auto kbc, auto bufc) {  // called once per block column, kb = NB - 1 first
    constexpr int kb = decltype(kbc)::value, buf = decltype(bufc)::value, Q = kb >> 2;
    const float* base[Q + 1];
    static_for<Q + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      base[q] = (ib[q] > kb) ? zpad : rowp[q];
      rowp[q] -= step[q];
    });
    static_for<16>([&](auto jc) {  // issued in the order they are consumed (LDS returns in order)
      constexpr int j = 15 - decltype(jc)::value;
      static_for<Q + 1>([&](auto qc) { col[buf][j][decltype(qc)::value] = base[decltype(qc)::value][j]; });
    });
  };
  // y sits in column f of the last block column
  dump(std::integral_constant<int, NB - 1>{});
  static_for<NQ>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    z[q] = rowp[q][f - 16 * (NB - 1)] * rdl[q];
  });
  constexpr int NBLK = NB;
  static_for<NBLK>([&](auto bc) {
    constexpr int n = decltype(bc)::value;
    constexpr int kb = NBLK - 1 - n;
    constexpr int buf = n & 1;
    constexpr int Q = kb >> 2;  // pivots of this block live in z[Q]
    if constexpr (Q < NQ) {
      if constexpr (n == 0) issue(std::integral_constant<int, kb>{}, std::integral_constant<int, buf>{});
      if constexpr (kb > 0) {
        dump(std::integral_constant<int, kb - 1>{});
        issue(std::integral_constant<int, kb - 1>{}, std::integral_constant<int, buf ^ 1>{});
      }
      if (16 * kb <= top) {  // uniform: the last block column may hold nothing but y
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
  float ssq = 0.f;  // this lane's share of ||x||^2 (rows past f hold zeros); the fused train SSE wants it
  static_for<NQ>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    if (lane + 64 * q < f) x_global[lane + 64 * q] = z[q];
    ssq = fmaf(z[q], z[q], ssq);
  });
  return ssq;
}

// ----------------------------------------------------------------------------------
// Train SSE of one row for free (round 4; als.cu:191-219 + 979-991 folded into the Theta update).  The rating rides in
// slot f of the gathered rows, so the Gram pass has also accumulated entry (f, f) of the augmented matrix
// [Theta r]^T [Theta r]: S = sum r^2.  With G = sum x x^T, b = sum r x, A = G + reg I (reg = lambda n):
//   sum_u (r - x_u . t)^2 = S - 2 t.b + t^T G t                                   for ANY t;
//   LU:  the elimination treats row / column f like every other trailing row, so entry (f, f) ends as the Schur
//        complement S + reg - b^T A^-1 b (the diagonal got reg everywhere, slot f included); with A t = b this is
//        S + reg - t.b, and t^T G t = t.b - reg |t|^2, hence SSE = (f, f) - reg (1 + |t|^2);
//   CG:  the tiles are untouched; with the recursive residual r = b - A t:  t^T G t = t.b - t.r - reg |t|^2, hence
//        SSE = S - t.b - t.r - reg |t|^2  (three dot products on vectors the solver holds anyway).
// No rating and no factor row is read again.  One fp64 atomic per row into kSseBins bins (the reference's own
// error bins, als.cu:216, hold fp32 partial sums); rows without ratings contribute nothing.
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sse_add(double* bins, double sse, int rowlen, int lane) {
  if (lane == 0 && rowlen > 0) atomicAdd(bins + (blockIdx.x & (kSseBins - 1)), sse);
}

// ----------------------------------------------------------------------------------
// Unpivoted Gaussian elimination of [A | b] on the accumulators of ONE wave + back substitution: the content of
// cublasSgetrfBatched(PivotArray = NULL) + cublasSgetrsBatched (als.cu:77,98 / 146,166).  Panels of four pivots
// p0 .. p0 + 3 (block row Ip, lane group q); rounds 2-3 ran every panel's rank-4 update on all live tiles with fp32 MFMAs
// (lu_wave, and a software-pipelined form of it: profiles/r04/lu_wave_serial_and_pipelined.hip.txt).
// ----------------------------------------------------------------------------------
// n-th tile (row-major) of the part of the upper triangle below block row I0: rows I0 .. NB - 1
template <int NB, int I0>
__host__ __device__ constexpr int lu_trailing_tile(int n) {
  for (int I = I0; I < NB; ++I) {
    if (n < NB - I) return tile_of<NB>(I, I + n);
    n -= NB - I;
  }
  return -1;
}
// product PROD (small terms first, as in the Gram pass) of the rank-16 bf16 update of tile t
template <int NB, int t, int PROD>
__device__ __forceinline__ void lu_trailing_mfma(f32x4 (&acc)[NB * (NB + 1) / 2], const u32x2 (&h)[NB], const u32x2 (&m)[NB],
                                                 const u32x2 (&l)[NB]) {
  constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
  if constexpr (PROD == 0) acc[t] = mfma_bf16_k16(l[I], h[J], acc[t]);
  if constexpr (PROD == 1) acc[t] = mfma_bf16_k16(h[I], l[J], acc[t]);
  if constexpr (PROD == 2) acc[t] = mfma_bf16_k16(m[I], m[J], acc[t]);
  if constexpr (PROD == 3) acc[t] = mfma_bf16_k16(m[I], h[J], acc[t]);
  if constexpr (PROD == 4) acc[t] = mfma_bf16_k16(h[I], m[J], acc[t]);
  if constexpr (PROD == 5) acc[t] = mfma_bf16_k16(h[I], h[J], acc[t]);
}

// product PROD of the rank-32 update of tile t by TWO block rows at once (round 6): K slots 0 .. 3 of a lane = the first row's
// four pivots 4 e + g, slots 4 .. 7 = the second row's -- v_mfma_f32_16x16x16_bf16 costs what the K = 32 form costs, so pairing
// the block rows halves the MFMAs of every tile that lies below both (204 instead of 336 per 100 x 100 system)
template <int NB, int t, int PROD>
__device__ __forceinline__ void lu_trailing_mfma32(f32x4 (&acc)[NB * (NB + 1) / 2], const u32x2 (&hA)[NB], const u32x2 (&mA)[NB],
                                                   const u32x2 (&lA)[NB], const u32x2 (&hB)[NB], const u32x2 (&mB)[NB],
                                                   const u32x2 (&lB)[NB]) {
  constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
  auto q = [](const u32x2& a, const u32x2& b) { return u32x4{a[0], a[1], b[0], b[1]}; };
  if constexpr (PROD == 0) acc[t] = mfma_bf16(q(lA[I], lB[I]), q(hA[J], hB[J]), acc[t]);
  if constexpr (PROD == 1) acc[t] = mfma_bf16(q(hA[I], hB[I]), q(lA[J], lB[J]), acc[t]);
  if constexpr (PROD == 2) acc[t] = mfma_bf16(q(mA[I], mB[I]), q(mA[J], mB[J]), acc[t]);
  if constexpr (PROD == 3) acc[t] = mfma_bf16(q(mA[I], mB[I]), q(hA[J], hB[J]), acc[t]);
  if constexpr (PROD == 4) acc[t] = mfma_bf16(q(hA[I], hB[I]), q(mA[J], mB[J]), acc[t]);
  if constexpr (PROD == 5) acc[t] = mfma_bf16(q(hA[I], hB[I]), q(hA[J], hB[J]), acc[t]);
}

template <int NB, int FC>
__device__ __forceinline__ float lu_wave_blocked(f32x4 (&acc)[NB * (NB + 1) / 2], float* T, int f_rt, float reg,
                                                 float* __restrict__ x_global, int lane, int dbg = 0) {
  constexpr int NT = NB * (NB + 1) / 2;
  const int f = FC ? FC : f_rt;
  LuLaneS ln;
  ln.c = lane & 15;
  ln.kk = (lane >> 4) & 3;
  ln.k1 = ln.kk == 1, ln.k2 = ln.kk == 2, ln.k3 = ln.kk == 3;
  ln.e1c = ln.k1 ? 1.0f : 0.f, ln.e2c = ln.k2 ? 1.0f : 0.f, ln.e3c = ln.k3 ? 1.0f : 0.f;  // unit diagonal of E
  ln.j0 = (lane & 3) == 0, ln.j1 = (lane & 3) >= 1, ln.j2 = (lane & 3) >= 2, ln.j3 = (lane & 3) == 3;
  ln.d1 = (lane & 3) == 1 ? 1.0f : 0.f, ln.d2 = (lane & 3) == 2 ? 1.0f : 0.f;
  // the system, negated: -(A + lambda n_u I) (als.cu:545-557 for the diagonal term)
  static_for<NT>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr bool diag = tile_I<NB>(t) == tile_J<NB>(t);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[t][r];
      if constexpr (diag) v = (4 * ln.kk + r == ln.c) ? v + reg : v;
      acc[t][r] = -v;
    }
  });
  float* rdiag = T + 16 * NB * kBsPitch;  // pivot reciprocals, then 16 zeros (rows outside a pivot block read these)
  float* zpad = rdiag + ((f + 3) & ~3);
  if (lane < 16) zpad[lane] = 0.f;
  float* xbuf = T;  // panel-row exchange: [lane group 4][block NB][column 16][row 4], in the (still unused) window
  float* side = T + wave_lu_side_offset<NB>(f);  // rows 12 .. 15 of the tiles above the last block row, as rows of W

  LuPrepS<NB> s;
  // bf16 planes of the w of the block row that has just been eliminated (blocks below it): the operands of its rank-16
  // update.  The tiles of the NEXT block row get theirs at once (its panels read them); the tiles below that are updated
  // ONE MFMA AT A TIME IN FRONT OF THE MICRO-STEPS of the next block row's panels: a bf16 MFMA runs beside the VALU work
  // of its own wave only when the two alternate in program order (in-order issue), and the partner wave covers but a
  // third of a burst (measured: the update in one burst per block row costs 0.92 ms of the Theta side's 4.6 ms solve).
  // Round 6: block rows in PAIRS.  The first row of a pair (Ip even) updates only the second row's tiles at once (rank 16: its
  // panels read them); everything below both waits for the second row and then takes ONE rank-32 update with the planes of
  // both (A, B) -- the tiles of the next block row at once, the rest one MFMA at a time in front of the micro-steps of the NEXT
  // pair's first row.
  u32x2 hA[NB], mA[NB], lA[NB], hB[NB], mB[NB], lB[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) hA[b] = mA[b] = lA[b] = hB[b] = mB[b] = lB[b] = u32x2{0u, 0u};
  static_for<NB>([&](auto ipc) {
    constexpr int Ip = decltype(ipc)::value;
    constexpr int L = NB - Ip;
    constexpr bool FIRST = (Ip & 1) == 0;
    // pending (first rows only): the previous pair's rank-32 update of the tiles below block row Ip
    constexpr int NTl = (FIRST && Ip >= 2) ? (L - 1) * L / 2 : 0;  // tiles of rows Ip + 1 .. NB - 1
    constexpr int TP = 6 * NTl;
    constexpr int S = lu_prep_steps<NB, Ip>();
    float w[4][NB];  // w[e][b]: panel e of this block row at feature block b >= Ip
    static_for<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int p0 = 16 * Ip + 4 * q;
      constexpr bool last_row = Ip == NB - 1;
      constexpr bool exists_static = !last_row || (FC != 0 && p0 < FC);
      constexpr bool dyn = last_row && FC == 0;            // the panel exists only if p0 < f (run time)
      constexpr bool dynp = (last_row && (FC & 3) != 0) || dyn;  // ... and may be short (a compile-time f that is no multiple of 4 too)
      auto panel = [&]() {
        float wm = 0.f;
#if CUMF_ABLATE
        // profiling build: 256 = no panel preparation (constants instead), 512 = no fp32 MFMAs
        if (dbg & 256) {
          static_for<L>([&](auto bc2) { w[q][Ip + decltype(bc2)::value] = acc[tile_of<NB>(Ip, Ip + decltype(bc2)::value)][q]; });
          wm = w[q][Ip];
        } else
#endif
        static_for<S>([&](auto sc) {
          constexpr int i = decltype(sc)::value;
          constexpr int gs = q * S + i;  // micro-step of the block row
          // pending MFMAs n in [gs TP / 4S, (gs + 1) TP / 4S): product-major, consecutive ones hit different tiles
          constexpr int n0 = gs * TP / (4 * S), n1 = (gs + 1) * TP / (4 * S);
          static_for<n1 - n0>([&](auto nc) {
            constexpr int n = n0 + decltype(nc)::value;
            lu_trailing_mfma32<NB, lu_trailing_tile<NB, Ip + 1>(n % NTl), n / NTl>(acc, hA, mA, lA, hB, mB, lB);
          });
#if CUMF_ABLATE
          lu_prep_step_s<NB, Ip, q, dynp, i>(acc, s, w[q], wm, rdiag, xbuf, f, ln, dbg);
#else
          lu_prep_step_s<NB, Ip, q, dynp, i>(acc, s, w[q], wm, rdiag, xbuf, f, ln);
#endif
          if constexpr (TP > 0) __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (q == 3 && !last_row) {
          // Round 6: the fourth panel's update of the block row would only finish its own rows 13 .. 15 for the back
          // substitution (no later panel reads this block row) -- and v_mfma_f32_16x16x4_f32 holds the SIMD for 36 cycles,
          // nothing issues beside it.  Those rows are kept as rows of W instead (lane group kk = row 12 + kk; zeros at and left
          // of the diagonal: wm), one ds_write_b32 per tile; back_substitute_tiles reads them from there.  82 instead of 109
          // fp32 MFMAs per 100 x 100 system.
          static_for<L>([&](auto bc2) {
            constexpr int b = Ip + decltype(bc2)::value;
            side[tile_of<NB>(Ip, b) * kSideSlot + wave_lu_side_skew<NB>(Ip) + 17 * ln.kk + ln.c] = b == Ip ? wm : w[q][b];
          });
        } else {
          // the block row's own tiles: what its next panel reads (and the rows the back substitution reads later)
#if CUMF_ABLATE
          if (!(dbg & 512))
#endif
          static_for<L>([&](auto bc2) {
            constexpr int b = Ip + decltype(bc2)::value;
            constexpr int t = tile_of<NB>(Ip, b);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm, w[q][b], acc[t], 0, 0, 0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);  // no instruction motion across panels (lu_wave: hoisted broadcasts spill)
      };
      if constexpr (exists_static) {
        panel();
      } else if constexpr (dyn) {
        if (p0 < f) panel();  // wave-uniform
      }
    });
#if CUMF_ABLATE
    if (!(dbg & 1024))  // profiling build: 1024 = no trailing update
#endif
    if constexpr (L > 1) {
      // planes of this block row's w; the tiles of block row Ip + 1 at once, block by block as the planes appear: rank 16 by
      // the first row of a pair, rank 32 (both rows' planes) by the second
      static_for<L - 1>([&](auto bc2) {
        constexpr int b = Ip + 1 + decltype(bc2)::value;
        unsigned H0, M0, L0, H1, M1, L1;
        split3_pair(w[0][b], w[1][b], H0, M0, L0);
        split3_pair(w[2][b], w[3][b], H1, M1, L1);
        constexpr int t = tile_of<NB>(Ip + 1, b);
        if constexpr (FIRST) {
          hA[b] = u32x2{H0, H1};
          mA[b] = u32x2{M0, M1};
          lA[b] = u32x2{L0, L1};
          static_for<6>([&](auto pc) { lu_trailing_mfma<NB, t, decltype(pc)::value>(acc, hA, mA, lA); });
        } else {
          hB[b] = u32x2{H0, H1};
          mB[b] = u32x2{M0, M1};
          lB[b] = u32x2{L0, L1};
          static_for<6>([&](auto pc) { lu_trailing_mfma32<NB, t, decltype(pc)::value>(acc, hA, mA, lA, hB, mB, lB); });
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  });
  __syncthreads();  // one wave: orders the rdiag writes before the reads below
#if CUMF_ABLATE
  if (dbg & 2048) {  // profiling build: no back substitution
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    if (lane < f) x_global[lane] = sum;
    return sum;
  }
#endif
  return back_substitute_tiles<NB, (16 * NB + 63) / 64>(acc, T, rdiag, zpad, side, f, x_global, lane);
}

// ----------------------------------------------------------------------------------
// Conjugate gradient on dumped tiles (cg.cu:36-231: warm start, r = b - A x, <= cg_iters iterations,
// stop when ||r||^2 < 1e-4), NW waves per system, wave W holding the tiles t % NW == W in registers.
// Vectors live in "column layout": one register per 16-feature block, lane (g, c) = element
// 16 J + c, replicated over the four lane groups g; every wave keeps all vectors and performs the
// vector updates and dot products redundantly (identical instruction sequences on identical data:
// alpha / beta / the exit test are uniform without communication).  Mat-vec y = A v on a tile
// T = T(I, J), I <= J, in the C/D layout (lane (g, c), register r = T[4 g + r][c]):
//   (1) y_I[4 g + r] += sum_c T[r][c] v_J[c]        4 FMAs, then a 16-lane DPP reduction per (I, r)
//   (2) y_J[c]       += sum_{g, r} T[r][c] v_I[4 g + r]   (I < J: the mirrored half)   4 FMAs with v_I in
//       "row layout" (ds_bpermute from the column layout), then a 4-lane-group reduction per J
// and (1)'s result is brought back to the column layout with 3 selects + 1 ds_bpermute per block.
// NW > 1: the partial y of the waves go through LDS, one workgroup barrier pair per mat-vec.
// Dot products: per-lane FMAs over the blocks + the 16-lane DPP reduction (fixed order), in place of
// the reference's order-dependent shared-memory atomics (device_utilities.h:36-48).
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum(float v) {  // all-reduce over the 16 lanes of a DPP row
  v += dpp_term<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_term<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_term<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_term<0x140, 0xf>(v);  // row_mirror
  return v;
}

// Sums of FOUR registers over the 16 lanes of a DPP row, transposed: every lane c ends with the full sum of
// register c & 3.  11 instructions instead of 4 x row16_sum = 16, and the result is already where the
// row -> column layout change wants it (lane (g, c) holds row 4 g + (c & 3) of the block).
//   xor 1: lanes keep the register of their parity and send the other one  (4 selects + 2 adds)
//   xor 2: the same on the two pair sums                                      (2 selects + 1 add)
//   the four quads of the row hold the same register in the same position: row_ror 4, row_ror 8 (2 adds)
__device__ __forceinline__ float row16_sum4_transposed(float r0, float r1, float r2, float r3, bool c1, bool c2) {
  const float keep01 = c1 ? r1 : r0, send01 = c1 ? r0 : r1;
  const float keep23 = c1 ? r3 : r2, send23 = c1 ? r2 : r3;
  const float t01 = keep01 + dpp_term<0xB1, 0xf>(send01);  // quad_perm [1,0,3,2]
  const float t23 = keep23 + dpp_term<0xB1, 0xf>(send23);
  const float keep = c2 ? t23 : t01, send = c2 ? t01 : t23;
  float w = keep + dpp_term<0x4E, 0xf>(send);  // quad_perm [2,3,0,1]
  w += dpp_term<0x124, 0xf>(w);                // row_ror:4
  w += dpp_term<0x128, 0xf>(w);                // row_ror:8
  return w;
}
// Two FMAs on a register pair, as two scalar v_fma_f32.  Round 3 saw wrong CG mat-vecs (1-4 % of the long Netflix X rows, a
// different set every run) in a build whose compiler had formed v_pk_fma_f32 here (profiles/r03/pk_fma_bisect.txt) and
// kept packed fp32 math out ever since (-fno-slp-vectorize; tests/test_capi_symbols.py disassembles the objects).  Round 4
// looked again: a standalone probe (tools/probes/pk_fma_probe.hip: 3e12 packed FMAs feeding DPP reductions and ds_bpermute
// beside bf16 and fp32 MFMA waves, against scalar FMAs: 0 mismatches) and this very CG with fma2 spelled as ONE inline-asm
// v_pk_fma_f32 (full-size oracle rows green three times, RMSE identical to 1e-16, profiles/r04/pk_fma_cg_ab.txt) are
// clean -- the instruction is not at fault, that build's generated code was -- and the packed CG is 5 % SLOWER (the packed
// FMA issues at half rate and needs aligned register pairs): the scalar form stays, for speed.
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) {
  return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
}

template <int NB, int NW, int W, int I>
__host__ __device__ constexpr bool cg_row_has_offdiag() {
  for (int J = I + 1; J < NB; ++J)
    if (tile_of<NB>(I, J) % NW == W) return true;
  return false;
}
template <int NB, int NW, int W, int I>
__host__ __device__ constexpr bool cg_row_has_any() {
  for (int J = I; J < NB; ++J)
    if (tile_of<NB>(I, J) % NW == W) return true;
  return false;
}

template <int NB, int NW, int W>
__device__ __forceinline__ void cg_wave_core(f32x4 (&T)[(NB * (NB + 1) / 2 + NW - 1) / NW], float* smem,
                                             const KernelArgs& a, int f, int row, int rowlen, int lane) {
  const int c = lane & 15, g = lane >> 4;
  auto bperm = [](int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
  };
  const float reg = (float)rowlen * a.lambda;  // lambda * n_u on the diagonal (als.cu:545-557)
  static_for<NB>([&](auto ic) {
    constexpr int t = tile_of<NB>(decltype(ic)::value, decltype(ic)::value);
    if constexpr (t % NW == W) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = T[t / NW][r] + reg;
        T[t / NW][r] = (4 * g + r == c) ? d : T[t / NW][r];
      }
    }
  });
  bool live[NB];  // this lane's element of block J exists (16 J + c < f)
  static_for<NB>([&](auto jc) { live[decltype(jc)::value] = 16 * decltype(jc)::value + c < f; });
  float* xch = smem;  // NW > 1: [wave][NB][16] partial vectors
  const int sel_addr = 4 * (16 * (c >> 2) + c);  // lane (c >> 2, c): holds element c of a row-layout block after (1)
  const int row_addr = 4 * (20 * g);             // + 4 r: lane (g, 4 g + r) holds v[16 I + 4 g + r] in the column layout
  // row layout (lanes of group g, registers r: element 4 g + r, the same in all 16 lanes) -> column layout
  const bool cr1 = (c & 3) == 1, cr2 = (c & 3) == 2, cr3 = (c & 3) == 3;
  auto to_col = [&](const float (&R)[4]) {
    float w = R[0];  // flat selects (v_cndmask): a nested ?: becomes exec-mask branches
    w = cr1 ? R[1] : w;
    w = cr2 ? R[2] : w;
    w = cr3 ? R[3] : w;
    return bperm(sel_addr, w);
  };
  // sum of the waves' partial column-layout vectors (NW > 1)
  auto combine = [&](float (&y)[NB]) {
    if constexpr (NW > 1) {
      __syncthreads();  // the previous exchange has been read
      if (g == 0) {
        static_for<NB>([&](auto jc) { xch[(W * NB + decltype(jc)::value) * 16 + c] = y[decltype(jc)::value]; });
      }
      __syncthreads();
      static_for<NB>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        float t = 0.f;
        static_for<NW>([&](auto wc) { t += xch[(decltype(wc)::value * NB + J) * 16 + c]; });  // same order in every wave
        y[J] = t;
      });
    }
  };
  // ---- right-hand side: column f of the last tile column, b[16 I + i] = T(I, NB - 1)[i][f - 16 (NB - 1)]
  float b[NB];
  {
    const int cf = f - 16 * (NB - 1);
    static_for<NB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int t = tile_of<NB>(I, NB - 1);
      b[I] = 0.f;
      if constexpr (t % NW == W) {
        float R[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) R[r] = bperm(4 * (16 * g + cf), T[t / NW][r]);  // lane (g, cf) over its row
        b[I] = to_col(R);
      }
    });
    combine(b);
    static_for<NB>([&](auto jc) { b[decltype(jc)::value] = live[decltype(jc)::value] ? b[decltype(jc)::value] : 0.f; });
  }
  // ---- y = A v.  Per tile 4 + 4 FMAs (direct half: rows of the tile against v_J; mirrored half: columns
  // against v_I in the row layout), per block row ONE transposed 4-register reduction and one ds_bpermute.
  const bool c1 = (c & 1) != 0, c2 = (c & 2) != 0;
  auto matvec = [&](const float (&v)[NB], float (&y)[NB]) {
    f32x2 ca[NB];
    static_for<NB>([&](auto jc) {
      ca[decltype(jc)::value] = f32x2{0.f, 0.f};
      y[decltype(jc)::value] = 0.f;
    });
    static_for<NB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (cg_row_has_any<NB, NW, W, I>()) {
        f32x2 pr01 = {0.f, 0.f}, pr23 = {0.f, 0.f};
        if constexpr (cg_row_has_offdiag<NB, NW, W, I>()) {
          pr01 = f32x2{bperm(row_addr, v[I]), bperm(row_addr + 4, v[I])};
          pr23 = f32x2{bperm(row_addr + 8, v[I]), bperm(row_addr + 12, v[I])};
        }
        f32x2 ra01 = {0.f, 0.f}, ra23 = {0.f, 0.f};
        static_for<NB>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          if constexpr (J >= I && tile_of<NB>(I, J) % NW == W) {
            constexpr int s = tile_of<NB>(I, J) / NW;
            const f32x2 t01 = __builtin_shufflevector(T[s], T[s], 0, 1), t23 = __builtin_shufflevector(T[s], T[s], 2, 3);
            const f32x2 vj = {v[J], v[J]};
            ra01 = fma2(t01, vj, ra01);
            ra23 = fma2(t23, vj, ra23);
            if constexpr (J > I) ca[J] = fma2(t23, pr23, fma2(t01, pr01, ca[J]));
          }
        });
        y[I] = bperm(sel_addr, row16_sum4_transposed(ra01[0], ra01[1], ra23[0], ra23[1], c1, c2));
      }
    });
    static_for<NB>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      float t = ca[J][0] + ca[J][1];
      t += bperm(4 * (lane ^ 16), t);
      t += bperm(4 * (lane ^ 32), t);
      y[J] += t;
    });
    combine(y);
    static_for<NB>([&](auto jc) { y[decltype(jc)::value] = live[decltype(jc)::value] ? y[decltype(jc)::value] : 0.f; });
  };
  // vector operations on block pairs (see fma2: scalar FMAs); NB odd: the last block alone
  auto pair = [](const float (&u)[NB], int j) { return f32x2{u[j], u[j + 1]}; };
  auto dot = [&](const float (&u)[NB], const float (&v)[NB]) {
    f32x2 t2 = {0.f, 0.f};
    static_for<NB / 2>([&](auto jc) {
      constexpr int j = 2 * decltype(jc)::value;
      t2 = fma2(pair(u, j), pair(v, j), t2);
    });
    float t = t2[0] + t2[1];
    if constexpr (NB & 1) t = fmaf(u[NB - 1], v[NB - 1], t);
    return row16_sum(t);
  };
  // y = a * u + y
  auto axpy = [&](float a, const float (&u)[NB], float (&y)[NB]) {
    const f32x2 a2 = {a, a};
    static_for<NB / 2>([&](auto jc) {
      constexpr int j = 2 * decltype(jc)::value;
      const f32x2 t = fma2(a2, pair(u, j), pair(y, j));
      y[j] = t[0];
      y[j + 1] = t[1];
    });
    if constexpr (NB & 1) y[NB - 1] = fmaf(a, u[NB - 1], y[NB - 1]);
  };
  // ---- CG (cg.cu:36-231)
  float* xg = a.update + (size_t)row * f;
  float x[NB], r[NB], p[NB], ap[NB];
  static_for<NB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const float xv = xg[live[J] ? 16 * J + c : 0];  // warm start (cg.cu:48); dead lanes read element 0 and drop it
    x[J] = live[J] ? xv : 0.f;
  });
  matvec(x, ap);
  static_for<NB>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    r[J] = b[J] - ap[J];
    p[J] = r[J];
  });
  float rsold = dot(r, r);
#if CUMF_ABLATE
  int iters_run = 0;
#endif
  for (int iter = 0; iter < a.cg_iters; ++iter) {
#if CUMF_ABLATE
    ++iters_run;
#endif
    matvec(p, ap);
    const float pap = dot(p, ap);
    const float alpha = rsold / pap;
    axpy(alpha, p, x);
    axpy(-alpha, ap, r);
    const float rsnew = dot(r, r);
    if ((double)rsnew < 1e-4) break;  // CG_ERROR (cg.cu:31,195); uniform: every wave computes the same bits
    const float beta = rsnew / rsold;
    rsold = rsnew;
    // p = r + beta p
    const f32x2 b2 = {beta, beta};
    static_for<NB / 2>([&](auto jc) {
      constexpr int j = 2 * decltype(jc)::value;
      const f32x2 t = fma2(b2, pair(p, j), pair(r, j));
      p[j] = t[0];
      p[j + 1] = t[1];
    });
    if constexpr (NB & 1) p[NB - 1] = fmaf(beta, p[NB - 1], r[NB - 1]);
  }
  if (W == 0 && g == 0) {
    static_for<NB>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      if (live[J]) xg[16 * J + c] = x[J];
    });
  }
#if CUMF_ABLATE
  if ((a.dbg & 65536) && W == 0 && lane == 0) atomicAdd(&g_cg_hist[iters_run < 15 ? iters_run : 15], 1ull);
#endif
  {
    // fused train SSE: S - x.b - x.r - reg |x|^2 (see wave_tile_ff).  Every wave holds all the vectors (identical bits);
    // the one that owns the last diagonal tile -- entry (f, f) = sum r^2 -- reports.
    constexpr int NT1 = NB * (NB + 1) / 2;
    if constexpr ((NT1 - 1) % NW == W) {
      if (a.sse_bins != nullptr) {
        const float S = wave_tile_ff<NB>(T[(NT1 - 1) / NW], f) - reg;  // the diagonal carries reg in slot f too
        const float xb = dot(x, b), xr = dot(x, r), xx = dot(x, x);
        wave_sse_add(a.sse_bins, (double)S - (double)xb - (double)xr - (double)reg * (double)xx, rowlen, lane);
      }
    }
  }
}

// tiles of wave W from the dumped slots (summed in slot order), then the CG
template <int NB, int NW, int W>
__device__ __forceinline__ void cg_wave_body(float* smem, const KernelArgs& a, int row, int slot0, int nslots,
                                             int rowlen, int lane) {
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int TPW = (NT + NW - 1) / NW;
  f32x4 T[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) T[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < nslots; ++sl) {
    const float* part = a.part + (size_t)(slot0 + sl) * NT * 256;
    static_for<TPW>([&](auto sc) {
      constexpr int t = W + NW * decltype(sc)::value;
      if constexpr (t < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) T[decltype(sc)::value][r] += part[((size_t)t * 4 + r) * 64 + lane];
      }
    });
  }
  cg_wave_core<NB, NW, W>(T, smem, a, a.f, row, rowlen, lane);
}

template <int NB, int NW>
__global__ __launch_bounds__(64 * NW, 2) void als_wave_cg_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int mr = blockIdx.x;
  const int row = a.mrow_row[mr];
  const int slot0 = a.dense_slots ? mr : a.mrow_slot0[mr];
  const int nslots = a.dense_slots ? 1 : a.mrow_nslots[mr];
  const int rowlen = a.mrow_rowlen[mr];
  if constexpr (NW == 1) {
    cg_wave_body<NB, 1, 0>(smem, a, row, slot0, nslots, rowlen, lane);
  } else if constexpr (NW == 2) {
    if ((threadIdx.x >> 6) == 0)
      cg_wave_body<NB, NW, 0>(smem, a, row, slot0, nslots, rowlen, lane);
    else
      cg_wave_body<NB, NW, 1>(smem, a, row, slot0, nslots, rowlen, lane);
  } else {
    static_assert(NW == 4, "1, 2 or 4 waves per system");
    switch (threadIdx.x >> 6) {
      case 0: cg_wave_body<NB, NW, 0>(smem, a, row, slot0, nslots, rowlen, lane); break;
      case 1: cg_wave_body<NB, NW, 1>(smem, a, row, slot0, nslots, rowlen, lane); break;
      case 2: cg_wave_body<NB, NW, 2>(smem, a, row, slot0, nslots, rowlen, lane); break;
      default: cg_wave_body<NB, NW, 3>(smem, a, row, slot0, nslots, rowlen, lane); break;
    }
  }
}

// ----------------------------------------------------------------------------------
// Kernel: one 64-thread workgroup (= one wave) per plan item.  FC != 0: f known at compile time.
// ----------------------------------------------------------------------------------
#define CUMF_WAVE_MIN_WAVES 2
// kArithFast epilogue of the Gram pass: the accumulators carry 4096^2 x the Gram; a rating beyond the
// f16 range (|r| >= 15.99; the table is checked by presplit_f16x2_kernel) shows as a non-finite
// right-hand side (column f lives in the tiles of the last block column) and is reported through
// a.fast_flag (cumf_gram_fast_status).  NW, W: tile t belongs to wave t % NW, slot t / NW.
template <int NB, int NW, int W>
__device__ __forceinline__ void fast_unscale(f32x4 (&acc)[(NB * (NB + 1) / 2 + NW - 1) / NW], int* flag) {
  constexpr int NT = NB * (NB + 1) / 2, TPW = (NT + NW - 1) / NW;
  float probe = 0.f;
  static_for<TPW>([&](auto sc) {
    constexpr int sl = decltype(sc)::value, t = W + NW * sl;
    if constexpr (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[sl][r] *= kFastUnscale;
      if constexpr (tile_J<NB>(t) == NB - 1) probe += (acc[sl][0] + acc[sl][1]) + (acc[sl][2] + acc[sl][3]);
    }
  });
  if (!(__builtin_fabsf(probe) <= 3.0e38f)) atomicOr(flag, 2);
}

// WHOLE: every item of the launch is a whole row (the plan has no chunked rows: the Netflix Theta side, the
// hugewiki X side).  The kernel then has no "dump the partial tiles" exit between the Gram pass and the solver,
// and THAT exit is what made the register allocator relocate the accumulator tiles at the hand-over and park
// five of them in scratch (41 spilled registers, 2.9 GB of scratch writes per Netflix Theta launch; 0 without it).
// (the CG instances of the small systems sit at the edge of three waves per SIMD -- 166 registers at NB = 5 -- and the packed form
// went over it: 174 registers, Theta side 4.6 -> 4.95 ms at f = 64; held at three)
template <int NB, int MODE, int FC, int ARITH = kArithSplit3, bool WHOLE = false>
__global__ __launch_bounds__(64, (NB <= 5 && MODE == kModeCG) ? 3 : CUMF_WAVE_MIN_WAVES) void als_wave_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = NB * (NB + 1) / 2;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int row = a.item_row[item];
  const long long begin = a.item_begin[item];
  const int len = a.item_len[item];
  const int slot = WHOLE ? -1 : (a.dense_slots ? item : a.item_slot[item]);
  const int rowlen = a.item_rowlen[item];
  const int f = FC ? FC : a.f;
  // (A static s_setprio for the wave in the odd hardware slot, meant to break a lockstep of the two waves
  // of a SIMD, measured neutral to slightly negative -- X side 6.70 vs 6.57 ms without it -- and is gone.)

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // at least one stage, also for a row without ratings (it gathers the zero row: see WaveGather::init)
  const int nst = len > 0 ? (len + kWaveStage - 1) / kWaveStage : 1;
  const int nfull = len / kWaveStage;
#if CUMF_ABLATE
  // the profiling build only (libALS_ablate.so, -DCUMF_ABLATE=1; results are wrong on purpose): 2 = no Gram pass
  if (!(a.dbg & 2))
#endif
  {
    auto clamp = [&](int s) { return s < nst ? s : nst - 1; };
    if constexpr (ARITH == kArithPre || ARITH == kArithPrePk) {
      constexpr bool PK = ARITH == kArithPrePk;
      PreGather<NB, PK> wg;
      wg.init(a, f, begin, len, lane, smem);
      PreStage<NB> R;
      Planes<NB, PK ? kArithPrePk : kArithSplit3> P;
      // prologue: chunks of stage 0 in flight, its rating values, the indices of stage 1
      wg.template load_idx<false>(R, 0);
      wg.template dma_issue<false>(R, smem, 0);
      wg.template load_rv<false>(R, 0);
      wg.template load_idx<false>(R, clamp(1));
      int s = 0;
      if constexpr (PK) {
        for (; s + 2 < nfull; ++s) stage_step_pk<NB, kStepFull>(wg, P, R, smem, acc, s + 1, s + 2);
        for (; s + 1 < nst; ++s) stage_step_pk<NB, kStepPartial>(wg, P, R, smem, acc, s + 1, s + 2);
        stage_step_pk<NB, kStepLast>(wg, P, R, smem, acc, 0, 0);
        wave_fold_strip<NB>(acc, wg.sp, lane);
      } else {
        for (; s + 2 < nfull; ++s) stage_step_pre<NB, kStepFull, PK>(wg, P, R, smem, acc, s + 1, s + 2);
        for (; s + 1 < nst; ++s) stage_step_pre<NB, kStepPartial, PK>(wg, P, R, smem, acc, s + 1, s + 2);
        stage_step_pre<NB, kStepLast, PK>(wg, P, R, smem, acc, 0, 0);
      }
    } else {
    constexpr bool SPK = ARITH == kArithSplitPk;
    WaveGather<NB> wg;
    wg.init(a, f, begin, len, lane, SPK);
    WaveStage<NB> R;
    Planes<NB, ARITH> P;
    lds_float_ptr lds = (lds_float_ptr)smem;  // staging chunks of this wave (the LU window aliases them later)
    const float* lds_lane = smem + lane;
    // prologue: chunks + ratings of stage 0 in flight, indices of stage 1
    wg.template load_idx<false>(R, 0);
    wg.template dma_issue<false, SPK>(R, lds, 0);
    wg.template load_val<false>(R, 0);
    wg.template load_idx<false>(R, clamp(1));
    int s = 0;
    // stages s + 1, s + 2 full: select-free steps; then the clamped form; the last stage of the item
    // prefetches nothing (three step bodies, each branch-free)
    if constexpr (SPK) {
      const int c = lane & 15;
      for (; s + 2 < nfull; ++s) stage_step_splitpk<NB, kStepFull>(wg, P, R, lds, lds_lane, acc, s + 1, s + 2, c);
      for (; s + 1 < nst; ++s) stage_step_splitpk<NB, kStepPartial>(wg, P, R, lds, lds_lane, acc, s + 1, s + 2, c);
      stage_step_splitpk<NB, kStepLast>(wg, P, R, lds, lds_lane, acc, 0, 0, c);
      wave_fold_strip<NB>(acc, false, lane);
    } else {
      for (; s + 2 < nfull; ++s) stage_step<NB, kStepFull, ARITH>(wg, P, R, lds, lds_lane, acc, s + 1, s + 2);
      for (; s + 1 < nst; ++s) stage_step<NB, kStepPartial, ARITH>(wg, P, R, lds, lds_lane, acc, s + 1, s + 2);
      stage_step<NB, kStepLast, ARITH>(wg, P, R, lds, lds_lane, acc, 0, 0);  // the last stage prefetches nothing
    }
    }
  }
  if constexpr (ARITH == kArithFast) fast_unscale<NB, 1, 0>(acc, a.fast_flag);
  if constexpr (ARITH != kArithFast) {
#if CUMF_ABLATE
    if (!(a.dbg & 2))
#endif
      wave_symmetrise_diag<NB>(acc, smem, lane);
  }

  if constexpr (!WHOLE) {
    if (slot >= 0) {
      wave_tiles_to_partial<NB>(acc, a.part + (size_t)slot * NT * 256, lane);
      return;
    }
  }
  const float reg = (float)rowlen * a.lambda;  // als.cu:547: (end - start) * lambda
#if CUMF_ABLATE
  if (a.dbg & 1) {  // ablation: no solve (keep the accumulators alive)
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    if (lane < f) a.update[(size_t)row * f + lane] = sum;
    return;
  }
#endif
  if constexpr (MODE == kModeMaterialize) {
    const size_t off = (size_t)(row - a.row_begin) * (a.tt_packed ? (size_t)f * (f + 1) / 2 : (size_t)f * f);
    float* rhs = a.rhs ? a.rhs + (size_t)(row - a.row_begin) * f : nullptr;
    if (a.tt_half)
      wave_tiles_to_global<NB>(acc, reinterpret_cast<_Float16*>(a.tt) + off, rhs, f, reg, lane);
    else
      wave_tiles_to_global<NB>(acc, a.tt + off, rhs, f, reg, lane, a.tt_packed != 0);
  } else if constexpr (MODE == kModeCG) {
    cg_wave_core<NB, 1, 0>(acc, smem, a, f, row, rowlen, lane);  // the reference's default solver (als.cu:28)
  } else {
#if CUMF_ABLATE
    const float ssq = lu_wave_blocked<NB, FC>(acc, smem, f, reg, a.update + (size_t)row * f, lane, a.dbg);
#else
    const float ssq = lu_wave_blocked<NB, FC>(acc, smem, f, reg, a.update + (size_t)row * f, lane);
#endif
    constexpr float ff_sign = -1.0f;  // the blocked elimination works on the negated system
    if (a.sse_bins != nullptr) {  // fused train SSE: (f, f) of the eliminated system - reg (1 + |theta|^2)
      const float ff = ff_sign * wave_tile_ff<NB>(acc[NT - 1], f);
      const float tt = wave_sum_uniform(ssq);
      wave_sse_add(a.sse_bins, (double)ff - (double)reg * (1.0 + (double)tt), rowlen, lane);
    }
  }
}

// ----------------------------------------------------------------------------------
// Large systems (NB = 8 .. 13, f = 112 .. 207): NW = 2 waves per item.  The upper-triangular tiles
// are dealt to the two waves (tile t belongs to wave t % 2: 46 / 45 tiles at NB = 13), both waves
// need every feature block of the stage as an operand, so the 8 NB gather chunks are shared: each
// wave issues the LDS-DMA loads of its half of the chunks, two workgroup barriers per stage make the
// hand-over (all chunks landed / all chunks read), and each wave splits all blocks for itself
// (redundant VALU: the alternative is a third pass through LDS).  No solve in this kernel: every
// item dumps its tiles (plan slots for chunk items, dense slots for whole rows) and
// als_reduce_kernel finishes the rows (LU, CG for f <= 128, or the materialised f x f Gram for
// cg_global_kernel) -- the reference's own data flow (als.cu:782-831).
// ----------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------
// kArithPre for the two-wave kernel (round 6): the stage image of the pre-split table is SHARED by the two waves of the item.
// Same construction as PreGather, sized for FB = 7 .. 12 full blocks: a rating's plane (32 FB bytes) sits in a slot of RP =
// 320 (FB <= 9) or 448 bytes, chunk (E, p) = plane p of the four ratings 4 E + q as before; wave W fetches the ratings
// q = 2 W, 2 W + 1 of every chunk (one global_load_lds_dwordx4 per (E, p) and wave: 24 per stage and wave instead of 52 dword
// gathers), the strip of the ratings 16 W .. 16 W + 15 (64 bytes per rating: h | m | l of up to eight features + 16 zero
// bytes) and their rating values.  ONE stage buffer, two barriers per stage: chunks landed + rating pieces written ->
// barrier -> both waves read ALL blocks (6 NB transposing reads, no split: the 468 VALU instructions per stage and wave of
// the in-kernel form are gone) -> barrier -> the chunks of the next stage -> this wave's MFMAs, which cover the gather.
// The last block is read per plane (strip pieces | the rating piece [r_p 0 0 0] | zeros): the same operands in the same K
// slots as the in-kernel split, the same MFMA order per tile -- bit-identical accumulators.
// ----------------------------------------------------------------------------------
template <int NB>
struct PreGeo2 {
  static constexpr int FB = NB - 1;
  static constexpr int RP = FB <= 9 ? 320 : 448;      // = 64 or 192 (mod 256): four ratings -> four 64-byte bank groups
  static constexpr int LP = RP / 16;
  static constexpr int CS = 4 * RP;
  static constexpr int kMain = 24 * CS + 3 * 32;
  static constexpr int kStrip = kMain;                // 32 ratings x 64 B
  static constexpr int kZero = kStrip + 2048;         // 64 B
  static constexpr int kRating = kZero + 64;          // 32 ratings x 3 planes x 16 B
  static constexpr int kBytes = kRating + 1536;
  static_assert(32 * FB <= RP && 2 * LP <= 64, "a rating's plane fits its slot, two ratings fit the wave");
  __host__ __device__ static constexpr int chunk(int E, int p) { return (3 * E + p) * CS + 32 * (E >> 1); }
};

template <int NB, int W>
struct PreGather2 {
  using G = PreGeo2<NB>;
  const char* lane_base;
  const char* zero_base;
  const char* strip_base;
  const char* strip_zero;
  const int* ib;
  const float* vb;
  lds_tr_ptr tr_main, tr_last[2];
  unsigned pitch;
  int len, q, lane;
  bool dma_active, sp;

  __device__ __forceinline__ void init(const KernelArgs& a, int f, long long begin, int len_, int lane_, float* smem) {
    lane = lane_;
    len = len_;
    pitch = a.pre_pitch;
    const int spn = (f & 15) >> 2;  // strip pieces per plane: 0, 1 or 2
    sp = spn != 0;
    const int piece = lane % G::LP;
    q = 2 * W + (lane / G::LP < 2 ? lane / G::LP : 1);
    dma_active = lane < 2 * G::LP && piece < 2 * G::FB;
    lane_base = reinterpret_cast<const char*>(a.gather) + 16 * piece;
    zero_base = reinterpret_cast<const char*>(g_wave_zeros) + 16 * piece;
    strip_base = reinterpret_cast<const char*>(a.gather) + 96 * G::FB + 16 * (lane & 3);
    strip_zero = reinterpret_cast<const char*>(g_wave_zeros) + 16 * (lane & 3);
    ib = len_ > 0 ? a.colidx + begin : reinterpret_cast<const int*>(g_wave_zeros);
    vb = (len_ > 0 && a.val != nullptr) ? a.val + begin : g_wave_zeros;
    const int g = lane >> 4, j = (lane >> 2) & 3, aa = lane & 3;
    lds_byte_ptr base = (lds_byte_ptr)smem;
    tr_main = (lds_tr_ptr)(base + 6 * g * G::CS + 32 * g + G::RP * j + 8 * aa);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rho = 8 * g + 4 * u + j;
      const int off = aa < spn ? G::kStrip + 64 * rho + 8 * aa : (aa == spn ? G::kRating + 48 * rho : G::kZero);
      tr_last[u] = (lds_tr_ptr)(base + off);
    }
    // the zero halfwords of this wave's rating pieces (16 ratings x 48 B) and, wave 0, the zero pieces -- once; each wave only
    // clears what it alone writes afterwards, the barrier of the first stage publishes it
    if (lane < 48) reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::kRating + 768 * W)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (W == 0 && lane < 4) reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::kZero)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void load_idx(PreStage<NB>& st, int s) const {
    const int top = len > 0 ? len - 1 : 0;
#pragma unroll
    for (int E = 0; E < 8; ++E) {
      const int pos = kWaveStage * s + 4 * E + q;
      st.idx[E] = ib[pos < top ? pos : top];
    }
    const int ps = kWaveStage * s + 16 * W + (lane >> 2);
    st.sidx = ib[ps < top ? ps : top];
  }
  __device__ __forceinline__ void load_rv(PreStage<NB>& st, int s) const {
    const int top = len > 0 ? len - 1 : 0;
    const int pv = kWaveStage * s + 16 * W + (lane & 15);
    const float v = vb[pv < top ? pv : top];
    st.rv = pv < len ? v : 0.f;
  }
  __device__ __forceinline__ void put_rating(const PreStage<NB>& st, float* smem) const {
    unsigned H, M, L;
    split3_pair(st.rv, 0.f, H, M, L);
    if (lane < 16) {
      unsigned short* rp = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(smem) + G::kRating + 48 * (16 * W + lane));
      rp[0] = (unsigned short)H;
      rp[8] = (unsigned short)M;
      rp[16] = (unsigned short)L;
    }
  }
  __device__ __forceinline__ void dma_issue(const PreStage<NB>& st, float* smem, int s) const {
#if defined(__HIP_DEVICE_COMPILE__)
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    lds_byte_ptr lds = (lds_byte_ptr)smem;
    if (dma_active) {
      static_for<8>([&](auto ec) {
        constexpr int E = decltype(ec)::value;
        const char* row = lane_base + (unsigned long long)(unsigned)st.idx[E] * pitch;
        row = (kWaveStage * s + 4 * E + q < len) ? row : zero_base;
        static_for<3>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          __builtin_amdgcn_global_load_lds((gptr)row, (lptr)(lds + G::chunk(E, p) + 2 * W * G::RP - 32 * G::FB * p), 16, 32 * G::FB * p, 0);
        });
      });
    }
    if (sp) {  // wave-uniform: the strip of the ratings 16 W .. 16 W + 15
      const char* row = strip_base + (unsigned long long)(unsigned)st.sidx * pitch;
      row = (kWaveStage * s + 16 * W + (lane >> 2) < len) ? row : strip_zero;
      __builtin_amdgcn_global_load_lds((gptr)row, (lptr)(lds + G::kStrip + 1024 * W), 16, 0, 0);
    }
#endif
  }
  static __device__ __forceinline__ u32x2 tr_read(lds_tr_ptr p, int byte_off) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)((lds_byte_ptr)p + byte_off)));
  }
  template <int B0, int B1>
  __device__ __forceinline__ void read_blocks(Planes<NB>& P) const {
    static_for<B1 - B0>([&](auto bc) {
      constexpr int B = B0 + decltype(bc)::value;
      static_for<2>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const u32x2 vh = tr_read(tr_main, (3 * u + 0) * G::CS + 32 * B);
        const u32x2 vm = tr_read(tr_main, (3 * u + 1) * G::CS + 32 * B);
        const u32x2 vl = tr_read(tr_main, (3 * u + 2) * G::CS + 32 * B);
        P.h[B][2 * u] = vh[0], P.h[B][2 * u + 1] = vh[1];
        P.m[B][2 * u] = vm[0], P.m[B][2 * u + 1] = vm[1];
        P.l[B][2 * u] = vl[0], P.l[B][2 * u + 1] = vl[1];
      });
    });
  }
  __device__ __forceinline__ void read_last(Planes<NB>& P) const {
    constexpr int B = G::FB;
    const u32x2 h0 = tr_read(tr_last[0], 0), m0 = tr_read(tr_last[0], 16), l0 = tr_read(tr_last[0], 32);
    const u32x2 h1 = tr_read(tr_last[1], 0), m1 = tr_read(tr_last[1], 16), l1 = tr_read(tr_last[1], 32);
    P.h[B] = u32x4{h0[0], h0[1], h1[0], h1[1]};
    P.m[B] = u32x4{m0[0], m0[1], m1[0], m1[1]};
    P.l[B] = u32x4{l0[0], l0[1], l1[0], l1[1]};
  }
};

template <int NB, int NW, int W, int MODE, int ARITH>
__device__ __forceinline__ void multi_body(float* smem, const KernelArgs& a, long long begin, int len, int slot,
                                           int row, int rowlen, int lane) {
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int TPW = (NT + NW - 1) / NW;
  const int f = a.f;
  f32x4 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = len > 0 ? (len + kWaveStage - 1) / kWaveStage : 1;  // a row without ratings: one stage on zeros
  if constexpr (ARITH == kArithPre) {
    PreGather2<NB, W> wg;
    wg.init(a, f, begin, len, lane, smem);
    PreStage<NB> R;
    Planes<NB> P;
    auto clamp = [&](int s) { return s < nst ? s : nst - 1; };
    wg.load_idx(R, 0);
    wg.dma_issue(R, smem, 0);
    wg.load_rv(R, 0);
    wg.load_idx(R, clamp(1));
    for (int s = 0; s < nst; ++s) {
      // this wave's MFMAs in two groups (as the one-wave stage_step_pk): the tiles among the first HB blocks run under the
      // transposing reads of the other blocks; the order of the six products of any one tile is unchanged
      constexpr int HB = NB / 2;
      auto group = [&](auto first) {
        constexpr bool FIRST = decltype(first)::value;
        static_for<6>([&](auto pc) {
          constexpr int PROD = decltype(pc)::value;
          static_for<TPW>([&](auto sc) {
            constexpr int t = W + NW * decltype(sc)::value;
            if constexpr (t < NT) {
              constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t), sl = decltype(sc)::value;
              if constexpr ((J < HB) == FIRST) {
                if constexpr (PROD == 0) acc[sl] = mfma_bf16(P.l[I], P.h[J], acc[sl]);
                if constexpr (PROD == 1) acc[sl] = mfma_bf16(P.h[I], P.l[J], acc[sl]);
                if constexpr (PROD == 2) acc[sl] = mfma_bf16(P.m[I], P.m[J], acc[sl]);
                if constexpr (PROD == 3) acc[sl] = mfma_bf16(P.m[I], P.h[J], acc[sl]);
                if constexpr (PROD == 4) acc[sl] = mfma_bf16(P.h[I], P.m[J], acc[sl]);
                if constexpr (PROD == 5) acc[sl] = mfma_bf16(P.h[I], P.h[J], acc[sl]);
              }
            }
          });
        });
      };
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's chunks of stage s, its indices of s + 1, its rating values of s
      wg.put_rating(R, smem);
      __syncthreads();                      // ... and the partner's; the rating pieces of both
      wg.template read_blocks<0, HB>(P);
      __builtin_amdgcn_sched_barrier(0);
      wg.template read_blocks<HB, NB - 1>(P);
      wg.read_last(P);
      group(std::true_type{});
      static_for<6 * (NB - HB)>([&](auto) {  // the late reads one behind each of the first MFMAs (<= 16 LDS operations in flight)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __syncthreads();                      // both waves have their operands: the image is free
      if (s + 1 < nst) {                    // uniform
        wg.dma_issue(R, smem, s + 1);
        wg.load_idx(R, clamp(s + 2));
        wg.load_rv(R, s + 1);
      }
      group(std::false_type{});
    }
    __syncthreads();  // (the solvers alias the image: nobody may still be reading the last stage's pieces -- they are not, but the
                      // CG's exchange buffer and the LU's are written right away)
  } else {
    WaveGather<NB> wg;
    wg.init(a, f, begin, len, lane);
    WaveStage<NB> R;
    Planes<NB, ARITH> P;
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    auto clamp = [&](int s) { return s < nst ? s : nst - 1; };
    // this wave's share of the chunks of stage s (feature blocks b with b % NW == W)
    // two stage buffers: the chunks of stage s + 1 are issued before stage s is converted (a full
    // conversion + MFMA phase of lead time, one workgroup barrier per stage)
    constexpr int kBuf = wave_stage_lds_floats<NB>();
    auto issue_share = [&](int s, int buf) {
      lds_float_ptr lds = (lds_float_ptr)smem + buf * kBuf;
      static_for<8>([&](auto ec) {
        constexpr int E = decltype(ec)::value;
        const char* row = wg.template row_ptr<false, E>(R, s);
        static_for<NB>([&](auto bc) {
          constexpr int B = decltype(bc)::value;
          if constexpr (B % NW == W) {
            constexpr int k = E * NB + B;
            if constexpr (B + 1 < NB)
              __builtin_amdgcn_global_load_lds((gptr)row, (lptr)(lds + 64 * k - 16 * B), 4, 64 * B, 0);
            else
              __builtin_amdgcn_global_load_lds((gptr)(row + wg.last_off), (lptr)(lds + 64 * k), 4, 0, 0);
          }
        });
      });
    };
    wg.template load_idx<false>(R, 0);
    issue_share(0, 0);
    wg.template load_val<false>(R, 0);
    wg.template load_idx<false>(R, clamp(1));
    for (int s = 0; s < nst; ++s) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's chunks of stage s (and the indices of s + 1) are here
      __syncthreads();                      // ... and the partner's; everybody is done reading stage s - 1
      if (s + 1 < nst) issue_share(s + 1, (s + 1) & 1);  // uniform: the last stage prefetches nothing
      const float* lds_lane = smem + (s & 1) * kBuf + lane;
      // chunk -> registers -> planes, block by block (the raw values of one block live at a time)
      static_for<NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        static_for<8>([&](auto ec) {
          constexpr int E = decltype(ec)::value;
          R.raw[B][E] = lds_lane[64 * (E * NB + B)];
        });
        if constexpr (B == NB - 1) static_for<8>([&](auto ec) { wg.template finish_one<decltype(ec)::value, ARITH>(R); });
        static_for<4>([&](auto vc) { split_pair<NB, B, decltype(vc)::value>(R, P); });
      });
      if (s + 1 < nst) {
        wg.template load_val<false>(R, s + 1);
        wg.template load_idx<false>(R, clamp(s + 2));
      }
      static_for<gram_products<ARITH>()>([&](auto pc) {
        constexpr int PROD = decltype(pc)::value;
        static_for<TPW>([&](auto sc) {
          constexpr int t = W + NW * decltype(sc)::value;
          if constexpr (t < NT) {
            constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t), sl = decltype(sc)::value;
            if constexpr (ARITH == kArithFast) {
              if constexpr (PROD == 0) acc[sl] = mfma_f16(P.l[I], P.h[J], acc[sl]);
              if constexpr (PROD == 1) acc[sl] = mfma_f16(P.h[I], P.l[J], acc[sl]);
              if constexpr (PROD == 2) acc[sl] = mfma_f16(P.h[I], P.h[J], acc[sl]);
            } else {
              // (the four-product form of the diagonal tiles -- make_gram_sched -- was measured here too, round 5: 13 of 91
              // tiles, the doubled plane recomputed per product, one more barrier per item: f = 200 X side 29.73 vs 29.75 ms,
              // Theta side CG 36.0 vs 35.6, f = 128 LU 26.9 vs 26.9 -- no gain with one wave per SIMD, not kept)
              if constexpr (PROD == 0) acc[sl] = mfma_bf16(P.l[I], P.h[J], acc[sl]);
              if constexpr (PROD == 1) acc[sl] = mfma_bf16(P.h[I], P.l[J], acc[sl]);
              if constexpr (PROD == 2) acc[sl] = mfma_bf16(P.m[I], P.m[J], acc[sl]);
              if constexpr (PROD == 3) acc[sl] = mfma_bf16(P.m[I], P.h[J], acc[sl]);
              if constexpr (PROD == 4) acc[sl] = mfma_bf16(P.h[I], P.m[J], acc[sl]);
              if constexpr (PROD == 5) acc[sl] = mfma_bf16(P.h[I], P.h[J], acc[sl]);
            }
          }
        });
      });
    }
  }
  if constexpr (ARITH == kArithFast) fast_unscale<NB, NW, W>(acc, a.fast_flag);
  // a whole row (no slot): the two waves solve it where the tiles are -- 93 KB per row at f = 200 that
  // neither go out to HBM nor come back (measured: 45 GB each way per Netflix X half-iteration)
  if (slot < 0) {
#if CUMF_ABLATE
    if (a.dbg & 1) {  // ablation: no solve (keep the accumulators alive)
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < TPW; ++s) sum += (acc[s][0] + acc[s][1]) + (acc[s][2] + acc[s][3]);
      if (64 * W + lane < f) a.update[(size_t)row * f + 64 * W + lane] = sum;
      return;
    }
#endif
    if constexpr (MODE == kModeCG) {
      cg_wave_core<NB, NW, W>(acc, smem, a, f, row, rowlen, lane);
    } else {
      __syncthreads();  // the partner is done with the stage buffers: the LU's exchange buffers alias them
      lu_solve_wg<NB, W, NW>(acc, smem, f, (float)rowlen * a.lambda, a.update + (size_t)row * f, lane, a.sse_bins, rowlen);
    }
    return;
  }
  float* part = a.part + (size_t)slot * NT * 256;
  static_for<TPW>([&](auto sc) {
    constexpr int t = W + NW * decltype(sc)::value;
    if constexpr (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[((size_t)t * 4 + r) * 64 + lane] = acc[decltype(sc)::value][r];
    }
  });
}

template <int NB, int NW, int MODE, int ARITH = kArithSplit3>
__global__ __launch_bounds__(64 * NW, NB >= 10 ? 1 : 2) void als_wave_multi_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x;
  const long long begin = a.item_begin[item];
  const int len = a.item_len[item];
  const int slot = a.dense_slots ? item : (a.item_slot ? a.item_slot[item] : -1);
  const int row = a.item_row[item];
  const int rowlen = a.item_rowlen[item];
  static_assert(NW == 2, "two waves per item");
  if ((threadIdx.x >> 6) == 0)
    multi_body<NB, NW, 0, MODE, ARITH>(smem, a, begin, len, slot, row, rowlen, lane);
  else
    multi_body<NB, NW, 1, MODE, ARITH>(smem, a, begin, len, slot, row, rowlen, lane);
}

// This file is compiled twice per NB <= 7 (Makefile): part 0 holds everything but the LU form of
// als_wave_kernel, part 1 only that (wave_lu_launch), built with -mllvm -enable-misched=false: the pre-RA
// machine scheduler triples the accumulator spills at the Gram -> LU hand-over of that kernel (125 vs 38
// registers; Netflix f = 100 LU 18.4 -> 18.0 ms on the same box) while every other kernel is faster with it
// (f = 100 CG 16.6 vs 18.4, f = 200 CG 67 vs 82).
#ifndef CUMF_WAVE_PART
#define CUMF_WAVE_PART 0
#endif
template <int NB>
hipError_t wave_lu_launch(const KernelArgs& a, long n_items, hipStream_t stream);

#if CUMF_WAVE_PART == 0
template <int NB>
hipError_t wave_solve_launch(const KernelArgs& a, int mode, long n_rows, hipStream_t stream);
template <>
hipError_t wave_solve_launch<CUMF_WAVE_NB>(const KernelArgs& a, int mode, long n_rows, hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  if (mode != kModeCG) {
    return hipErrorInvalidValue;  // LU / materialise of dumped tiles: als_reduce_kernel (als_kernels.hip)
  } else if (mode == kModeCG) {
    // the tiles are VALU operands (VGPRs only): 91 tiles at NB = 13 = four waves x 23 tiles next to the five
    // vectors, at two waves per SIMD
    constexpr int NW = CUMF_WAVE_NB >= 10 ? 4 : 1;
    const size_t lds = NW > 1 ? (size_t)NW * CUMF_WAVE_NB * 16 * sizeof(float) : 0;
    hipLaunchKernelGGL((als_wave_cg_kernel<CUMF_WAVE_NB, NW>), dim3((unsigned)n_rows), dim3(64 * NW), lds, stream, a);
  }
  return hipGetLastError();
}

#endif  // CUMF_WAVE_PART == 0


#if CUMF_WAVE_SPLITPK
// f % 16 == 0 on the fp32 table: the packed rating block (CUMF_ALS_SPLITPK=0 keeps the six-product form)
static bool splitpk_wanted(const KernelArgs& a) {
  static const bool on = [] {
    const char* e = getenv("CUMF_ALS_SPLITPK");
    return !(e && *e == '0');
  }();
  return on && !a.no_pack && !a.pre_words && !a.fast_words && (a.f & 15) == 0;
}
#endif

#if CUMF_WAVE_PART == 1 && CUMF_WAVE_NB <= 7
// ---- part 1: the LU form of the wave-per-item kernel
template <int NB, int FC, int ARITH, bool WHOLE>
static hipError_t launch_wave_lu_w(const KernelArgs& a, long n_items, hipStream_t stream) {
  const size_t stage_lds = wave_stage_lds_floats<NB, ARITH>() * sizeof(float);
  const size_t lu_lds = wave_lu_lds_floats<NB>(a.f) * sizeof(float);
  const size_t lds = lu_lds > stage_lds ? lu_lds : stage_lds;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(als_wave_kernel<NB, kModeLU, FC, ARITH, WHOLE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  note_item_kernel(reinterpret_cast<const void*>(als_wave_kernel<NB, kModeLU, FC, ARITH, WHOLE>));
  hipLaunchKernelGGL((als_wave_kernel<NB, kModeLU, FC, ARITH, WHOLE>), dim3((unsigned)n_items), dim3(64), lds, stream, a);
  return hipGetLastError();
}
template <int NB, int FC, int ARITH>
static hipError_t launch_wave_lu(const KernelArgs& a, long n_items, hipStream_t stream) {
  // whole_only: the plan has no chunked rows, no item of this launch dumps partial tiles
  return (a.whole_only && !a.dense_slots) ? launch_wave_lu_w<NB, FC, ARITH, true>(a, n_items, stream)
                                           : launch_wave_lu_w<NB, FC, ARITH, false>(a, n_items, stream);
}
template <>
hipError_t wave_lu_launch<CUMF_WAVE_NB>(const KernelArgs& a, long n_items, hipStream_t stream) {
#if CUMF_WAVE_NB == 7
  // the reference's own specialisation: get_hermitian100 for f == 100 (als.cu:788-817)
  if (a.f == 100)
    return a.pre_words == 1 ? launch_wave_lu<7, 100, kArithPrePk>(a, n_items, stream)
           : a.pre_words    ? launch_wave_lu<7, 100, kArithPre>(a, n_items, stream)
           : a.fast_words   ? launch_wave_lu<7, 100, kArithFast>(a, n_items, stream)
                            : launch_wave_lu<7, 100, kArithSplit3>(a, n_items, stream);
#endif
#if CUMF_WAVE_PRE
  if (a.pre_words)
    return a.pre_words == 1 ? launch_wave_lu<CUMF_WAVE_NB, 0, kArithPrePk>(a, n_items, stream)
                            : launch_wave_lu<CUMF_WAVE_NB, 0, kArithPre>(a, n_items, stream);
#endif
  if (a.pre_words) return hipErrorInvalidValue;
#if CUMF_WAVE_SPLITPK
  if (splitpk_wanted(a)) return launch_wave_lu<CUMF_WAVE_NB, 0, kArithSplitPk>(a, n_items, stream);
#endif
  return a.fast_words ? launch_wave_lu<CUMF_WAVE_NB, 0, kArithFast>(a, n_items, stream)
                      : launch_wave_lu<CUMF_WAVE_NB, 0, kArithSplit3>(a, n_items, stream);
}
#endif

#if CUMF_WAVE_PART == 0
// ----------------------------------------------------------------------------------
// Launcher (called by launch_half_iteration, als_kernels.hip)
// ----------------------------------------------------------------------------------
template <int NB, int FC, int ARITH>
static hipError_t launch_wave_fc(const KernelArgs& a, int mode, long n_items, hipStream_t stream) {
  const size_t stage_lds = wave_stage_lds_floats<NB, ARITH>() * sizeof(float);
  if (mode == kModeMaterialize) {
    if constexpr (ARITH != kArithSplit3) {
      return hipErrorInvalidValue;  // materialise: the 24-bit arithmetic on the fp32 table only
    } else {
      note_item_kernel(reinterpret_cast<const void*>(als_wave_kernel<NB, kModeMaterialize, FC, kArithSplit3>));
      hipLaunchKernelGGL((als_wave_kernel<NB, kModeMaterialize, FC, kArithSplit3>), dim3((unsigned)n_items), dim3(64),
                         stage_lds, stream, a);
    }
  } else if (mode == kModeCG) {
    note_item_kernel(reinterpret_cast<const void*>(als_wave_kernel<NB, kModeCG, FC, ARITH>));
    hipLaunchKernelGGL((als_wave_kernel<NB, kModeCG, FC, ARITH>), dim3((unsigned)n_items), dim3(64), stage_lds, stream, a);
  } else {
    return wave_lu_launch<NB>(a, n_items, stream);  // part 1 of this file (picks FC and the arithmetic itself)
  }
  return hipGetLastError();
}

template <int NB>
hipError_t wave_item_launch(const KernelArgs& a, int mode, long n_items, hipStream_t stream);
template <>
hipError_t wave_item_launch<CUMF_WAVE_NB>(const KernelArgs& a, int mode, long n_items, hipStream_t stream) {
  if (n_items <= 0) return hipSuccess;
#if CUMF_WAVE_NB > 7
  // two waves per item; items without a slot (whole rows) are solved in the kernel (CG, or the LU of
  // als_lu_wg.h with two wave roles), items with one dump their tiles
  size_t lds = 2 * wave_stage_lds_floats<CUMF_WAVE_NB>() * sizeof(float);  // double-buffered stages
  if (lu_wg_lds_floats<CUMF_WAVE_NB>(a.f) * sizeof(float) > lds) lds = lu_wg_lds_floats<CUMF_WAVE_NB>(a.f) * sizeof(float);
  auto go = [&](auto kernel) {
    note_item_kernel(reinterpret_cast<const void*>(kernel));
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_items), dim3(128), lds, stream, a);
  };
  if (a.pre_words) {  // the pre-split table (round 6): one shared stage image instead of two buffers of dword chunks
    if (mode == kModeMaterialize) return hipErrorInvalidValue;
    lds = PreGeo2<CUMF_WAVE_NB>::kBytes;
    if (lu_wg_lds_floats<CUMF_WAVE_NB>(a.f) * sizeof(float) > lds) lds = lu_wg_lds_floats<CUMF_WAVE_NB>(a.f) * sizeof(float);
    if (mode == kModeCG)
      go(als_wave_multi_kernel<CUMF_WAVE_NB, 2, kModeCG, kArithPre>);
    else
      go(als_wave_multi_kernel<CUMF_WAVE_NB, 2, kModeLU, kArithPre>);
  } else if (a.fast_words) {
    if (mode == kModeMaterialize) return hipErrorInvalidValue;
    if (mode == kModeCG)
      go(als_wave_multi_kernel<CUMF_WAVE_NB, 2, kModeCG, kArithFast>);
    else
      go(als_wave_multi_kernel<CUMF_WAVE_NB, 2, kModeLU, kArithFast>);
  } else if (mode == kModeCG) {
    go(als_wave_multi_kernel<CUMF_WAVE_NB, 2, kModeCG, kArithSplit3>);
  } else {
    go(als_wave_multi_kernel<CUMF_WAVE_NB, 2, kModeLU, kArithSplit3>);
  }
  return hipGetLastError();
#else
  if (mode != kModeMaterialize && mode != kModeLU && mode != kModeCG) return hipErrorInvalidValue;
#if CUMF_WAVE_NB == 7
  // the reference's own specialisation: get_hermitian100 for f == 100 (als.cu:788-817)
  if (a.f == 100)
    return a.pre_words == 1 ? launch_wave_fc<7, 100, kArithPrePk>(a, mode, n_items, stream)
           : a.pre_words    ? launch_wave_fc<7, 100, kArithPre>(a, mode, n_items, stream)
           : a.fast_words   ? launch_wave_fc<7, 100, kArithFast>(a, mode, n_items, stream)
                            : launch_wave_fc<7, 100, kArithSplit3>(a, mode, n_items, stream);
#endif
#if CUMF_WAVE_PRE
  if (a.pre_words)
    return a.pre_words == 1 ? launch_wave_fc<CUMF_WAVE_NB, 0, kArithPrePk>(a, mode, n_items, stream)
                            : launch_wave_fc<CUMF_WAVE_NB, 0, kArithPre>(a, mode, n_items, stream);
#endif
  if (a.pre_words) return hipErrorInvalidValue;
#if CUMF_WAVE_SPLITPK
  if (mode != kModeMaterialize && splitpk_wanted(a)) return launch_wave_fc<CUMF_WAVE_NB, 0, kArithSplitPk>(a, mode, n_items, stream);
#endif
  return a.fast_words ? launch_wave_fc<CUMF_WAVE_NB, 0, kArithFast>(a, mode, n_items, stream)
                      : launch_wave_fc<CUMF_WAVE_NB, 0, kArithSplit3>(a, mode, n_items, stream);
#endif
}

#endif  // CUMF_WAVE_PART == 0

#if CUMF_ABLATE && CUMF_WAVE_PART == 0
// profiling build: read (and clear) the CG iteration histogram of this feature-block count's kernels
template <int NB>
hipError_t wave_cg_hist(unsigned long long* out16);
template <>
hipError_t wave_cg_hist<CUMF_WAVE_NB>(unsigned long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cg_hist), 16 * sizeof(unsigned long long));
  if (e != hipSuccess) return e;
  const unsigned long long zero[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_cg_hist), zero, sizeof(zero));
}
#endif

#if CUMF_WAVE_PART == 0 && CUMF_WAVE_NB == 7
// ----------------------------------------------------------------------------------
// The gather table as bf16 h | m | l planes (kArithPre; one launch per half-iteration on the launch stream: the factors
// change every half-iteration -- 7 MB read + 11 MB written for the Netflix X table).  One thread per value; the split is
// split3_pair's, the instruction sequence of the in-kernel split: the planes are the same bits.
// ----------------------------------------------------------------------------------
// sw: halfwords per plane in the strip (4: the one-wave kernels' [h 4][m 4][l 4][0 4]; 8: the two-wave kernels'
// [h 8][m 8][l 8][0 8]); what the strip's features do not fill is zero.
__global__ __launch_bounds__(256) void presplit_bf16x3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                              long long n, int f, int fb, int sw, unsigned pitch_halfs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long row = i / f;
  const int k = (int)(i - row * f);
  unsigned H, M, L;
  split3_pair(src[i], 0.f, H, M, L);
  unsigned short* r = dst + row * pitch_halfs;
  if (k < 16 * fb) {
    r[k] = (unsigned short)H;
    r[16 * fb + k] = (unsigned short)M;
    r[32 * fb + k] = (unsigned short)L;
  } else {
    const int e = k - 16 * fb, sf = f - 16 * fb;
    unsigned short* st = r + 48 * fb;
    st[e] = (unsigned short)H;
    st[sw + e] = (unsigned short)M;
    st[2 * sw + e] = (unsigned short)L;
    if (e == 0) {  // the rest of the strip: zeros
      for (int p = 0; p < 3; ++p)
        for (int z = sf; z < sw; ++z) st[p * sw + z] = 0;
      for (int z = 0; z < sw; ++z) st[3 * sw + z] = 0;
    }
  }
}
hipError_t launch_presplit3(const float* src, void* dst, long long rows, int f, hipStream_t stream) {
  const long long n = rows * f;
  if (n <= 0) return hipSuccess;
  if (!presplit_supported(f)) return hipErrorInvalidValue;
  const int fb = f / 16;
  const unsigned pitch = presplit_pitch(f);
  hipLaunchKernelGGL(presplit_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src,
                     static_cast<unsigned short*>(dst), n, f, fb, nb_for_f(f) > kMaxWaveNB ? 8 : 4, pitch / 2);
  return hipGetLastError();
}
#endif

}  // namespace cumf
