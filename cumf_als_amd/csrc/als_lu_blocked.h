// als_lu_blocked.h -- the pieces of the blocked elimination (round 4: lu_wave_blocked, als_wave.hip): the exact bf16x3 split
// (also the Gram stage's and the pre-split kernel's), the preparation of one four-pivot panel on the accumulators of ONE wave
// (DPP quad broadcasts + ds_bpermute, no barrier), the rank-16 bf16 MFMA.  (Round 5 shared them with a row-owning workgroup
// LU of the large systems that was built, was correct and was slower: it lives as profiles/r05/lu_rows_experiment.patch,
// not in the tree.)
#pragma once
#include "als_device.h"

namespace cumf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two fp32 -> packed bf16x2 (a in the low half), round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// x - (low / high bf16 of p), exact whenever the difference is representable (it is for the residuals of
// the split): one v_dot2c_f32_bf16 (x += p.lo * s.lo + p.hi * s.hi with s = (-1, 0) / (0, -1)) instead of
// unpack + subtract.  The selector pairs sit in SGPRs: as immediates the compiler emits the inline
// constant -1.0, which the instruction does not read as the bf16 pair (tools/probes/dot2_probe.hip).
__device__ __forceinline__ float sub_bf16_lo(float x, unsigned p) {
  unsigned sel;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(sel));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, sel), x, false);
}
__device__ __forceinline__ float sub_bf16_hi(float x, unsigned p) {
  unsigned sel;
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(sel));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, sel), x, false);
}

// ----------------------------------------------------------------------------------
// Unpivoted Gaussian elimination of [A | b] on accumulator tiles: the content of
// cublasSgetrfBatched(PivotArray = NULL) + cublasSgetrsBatched (als.cu:77,98 / 146,166).  Panels of four pivots
// p0 .. p0 + 3 (block row Ip, lane group q).
// ----------------------------------------------------------------------------------
struct LuLane {  // loop-invariant lane constants
  int c, kk;
  bool k1, k2, k3;
  float e1c, e2c, e3c;
};
template <int NB, int Ip>
__host__ __device__ constexpr int lu_prep_steps() { return 2 * (NB - Ip) + 8; }

// ----------------------------------------------------------------------------------
// The elimination BLOCKED by block rows (round 4: lu_wave_blocked, the LU the wave kernels run).  Measured on this part
// (tools/probes/issue_probe3.hip, profiles/r04/issue_probe3.txt): v_mfma_f32_16x16x4_f32 takes 36 cycles and does NOT run beside
// VALU work -- neither of its own wave nor of the partner wave (8 MFMAs + 48 v_fma: 611 cycles against 293 + 379) -- so the
// 333 rank-4 updates of the panel-serial form (12 k cycles) simply add to its ~2 200 VALU instructions (10 k): 21.5 k cycles
// per system, which is what the Theta side showed, and why interleaving them inside the wave bought 1 %.  The bf16 MFMA is
// different: 20 cycles for K = 16 or 32 and VALU work runs in its shadow (8 MFMAs + 48 v_fma: 335 cycles against 166 + 362).
// So only what the NEXT panel of the same block row needs stays on the fp32 pipe, and everything below the block row waits
// for the block row to finish:
//   per panel (four pivots)  eliminated rows u_k as before, scaled to w_k = u_k / sqrt(u_kk) (v_rsq_f32), and ONE rank-4
//                            fp32 MFMA per tile of the block row itself: 109 of them instead of 333;
//   per block row            the four w of every feature block below it, split exactly into three bf16 terms (the split
//                            of the Gram pass), and the rank-16 update of every tile below the block row as six
//                            v_mfma_f32_16x16x16_bf16 (hh, hm, mh, mm, hl, lh: 24-bit products, fp32 accumulation).  K slot
//                            (lane group g, element e) = pivot 4 e + g of the block row: exactly where w of panel e sits,
//                            no data movement.  Both operands are the same w: the accumulators hold -A (negated once, with
//                            the diagonal term) so that the update is the positive product w_k[i] w_k[j].
// Cholesky-like scaling, LU-like pivots: u_kk > 0 for the positive definite systems of ALS; an all-zero system gives NaN as
// the reference's unpivoted LU does.  Parity is by tolerance (the order of operations is not that of the right-looking loop the test restatement runs).
// What it bought (profiles/r04/lu_parts.txt, lu_three_way_ab.txt): the solve alone 4.55 -> 4.25 ms per Netflix Theta pass
// (480 189 systems), Theta side at f = 64 6.30 -> 5.87 ms, at f = 100 10.9 -> 10.8 ms (inside the box-to-box noise): a fused
// half-iteration costs the SUM of its Gram pass and its solve -- both phases are bound by VALU-type issue slots (an MFMA is
// one), so a wave in its Gram phase and its SIMD partner in the LU do not hide each other.
// ----------------------------------------------------------------------------------
template <int NB>
struct LuPrepS {  // state of one panel's preparation
  float R[NB][4];  // raw panel rows of -A at this lane's columns (ds_bpermute from lane group q)
  float n0, n1, n2, n3;      // this lane's column of the panel's four rows of the diagonal tile (the pivot block in the quad 4 q .. 4 q + 3)
  float rs0, rs1, rs2, rs3;  // 1 / sqrt(u_kk), quad-uniform
  float t0, t1, t2;          // multipliers: quad lane i holds m_i0, m_i1, m_i2
  float e0, e1, e2, rsk;     // row kk of E and 1 / sqrt(u_kk) of this lane group's pivot (after the broadcast)
};
struct LuLaneS : LuLane {
  bool j1, j2, j3, j0;  // position in the quad: (c & 3) >= 1, >= 2, == 3, == 0
  float d1, d2;         // unit diagonal of E seen from the quad: (c & 3) == 1, == 2
};

// value of quad lane I in all four lanes of every quad (DPP quad_perm [I, I, I, I])
template <int I>
__device__ __forceinline__ float quad_bcast(float v) {
  // bound_ctrl set: lets the compiler fold the move into the consuming VOP1 / VOP2 instruction (v_rsq_f32_dpp, v_fmac_f32_dpp)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), I * 0x55, 0xf, 0xf, true));
}

// acc + (quad lane I of t) * n as ONE v_fmac_f32_dpp (the compiler's DPP combiner leaves the tied-operand fmac alone and emits
// v_mov_b32_dpp + v_fmac).  Inline asm is opaque to the hazard recogniser, so the two wait states a DPP read needs behind a
// VALU write of the same register are spelled out (s_nop 1).
template <int I>
__device__ __forceinline__ float fma_quad_bcast(float t, float n, float acc) {
  static_assert(I >= 0 && I < 4, "quad lane");
  if constexpr (I == 0) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(t), "v"(n));
  if constexpr (I == 1) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(t), "v"(n));
  if constexpr (I == 2) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(t), "v"(n));
  if constexpr (I == 3) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(t), "v"(n));
  return acc;
}

// Micro-step STEP of the preparation of panel (Ip, q) on the negated system N = -A; w[b]: w of this panel at block b, wm:
// w[Ip] masked to the rows below the pivot (the A operand of the block row's own update).
// Round 4: the 4 x 4 pivot block is eliminated WHERE IT IS -- quad lanes 4 q .. 4 q + 3 of lane group q hold its columns in
// registers 0..3 of the diagonal tile -- by every lane on its own column with DPP quad broadcasts: N_ij += (N_0i / u_00) N_0j
// etc.  No v_readlane (10 + 7 v_mov per panel), no wave-uniform scalar chain; the rows of E and 1 / sqrt(u_kk) come out in
// quad lane kk and reach lane group kk with four ds_bpermute.  ~30 VALU per panel instead of ~66.
template <int NB, int Ip, int q, bool DYN, int STEP, class ACC>
__device__ __forceinline__ void lu_prep_step_s(const ACC& acc, LuPrepS<NB>& s, float (&w)[NB], float& wm,
                                               float* rdiag, float* xbuf, int f, const LuLaneS& ln, int dbg = 0) {
  constexpr int L = NB - Ip;
  constexpr int SD = tile_of<NB>(Ip, Ip);
  constexpr int p0 = 16 * Ip + 4 * q;
  auto sel = [](bool p, float a, float b) { return p ? a : b; };
  auto rsq = [](float d) { return __builtin_amdgcn_rsqf(d); };
  auto bperm = [](int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
  };
  const bool v1 = !DYN || p0 + 1 < f, v2 = !DYN || p0 + 2 < f, v3 = !DYN || p0 + 3 < f;
  if constexpr (STEP == 0) {
    // Round 6: the panel's raw rows reach the other lane groups through LDS -- every lane stores the four rows it holds of
    // every live block (one 16-byte store per block: [lane group][block][column][row r]), every lane then loads the four rows
    // of lane group q at its column (one 16-byte load per block, steps 1 .. L; the 16 lanes of a column read one address) --
    // instead of four ds_bpermute per block: 2 L LDS instructions per panel instead of 4 L (LDS operations of one wave
    // execute in order: no barrier).
    // (Stores by the 16 lanes of lane group q alone -- EXEC narrowed inside one asm block -- measured the same: the stores'
    // cost is not in the lanes that carry nothing useful, profiles/r06/ab_side_rows.txt.)
    static_for<L>([&](auto bc) {
      constexpr int b = Ip + decltype(bc)::value;
      *reinterpret_cast<f32x4*>(xbuf + ((ln.kk * NB + b) * 16 + ln.c) * 4) = acc[tile_of<NB>(Ip, b)];
    });
    s.n0 = acc[SD][0], s.n1 = acc[SD][1], s.n2 = acc[SD][2], s.n3 = acc[SD][3];
    s.rs0 = rsq(-quad_bcast<0>(s.n0));
    s.t0 = s.n0 * (s.rs0 * s.rs0);  // quad lane i: m_i0 = N_0i / u_00
    s.n1 = fma_quad_bcast<1>(s.t0, s.n0, s.n1);
    s.n2 = fma_quad_bcast<2>(s.t0, s.n0, s.n2);
    s.n3 = fma_quad_bcast<3>(s.t0, s.n0, s.n3);
  } else if constexpr (STEP <= L) {
    constexpr int b = Ip + STEP - 1;
#if CUMF_ABLATE
    // profiling build: 4096 / 8192 = the panel rows of the blocks right of the diagonal tile without their exchange
    if ((dbg & (4096 | 8192)) && b > Ip) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s.R[b][r] = acc[tile_of<NB>(Ip, b)][r];
    } else
#endif
    {
      const f32x4 rr = *reinterpret_cast<const f32x4*>(xbuf + ((q * NB + b) * 16 + ln.c) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) s.R[b][r] = rr[r];
    }
  } else if constexpr (STEP == L + 1) {
    const float d = -quad_bcast<1>(s.n1);
    s.rs1 = rsq(DYN ? sel(v1, d, 1.0f) : d);
    s.t1 = s.n1 * (s.rs1 * s.rs1);
    s.n2 = fma_quad_bcast<2>(s.t1, s.n1, s.n2);
    s.n3 = fma_quad_bcast<3>(s.t1, s.n1, s.n3);
  } else if constexpr (STEP == L + 2) {
    const float d = -quad_bcast<2>(s.n2);
    s.rs2 = rsq(DYN ? sel(v2, d, 1.0f) : d);
    s.t2 = s.n2 * (s.rs2 * s.rs2);
    s.n3 = fma_quad_bcast<3>(s.t2, s.n2, s.n3);
  } else if constexpr (STEP == L + 3) {
    const float d = -quad_bcast<3>(s.n3);
    s.rs3 = rsq(DYN ? sel(v3, d, 1.0f) : d);
    // multipliers that exist: m_i0 for i >= 1, m_i1 for i >= 2, m_32
    s.t0 = sel(ln.j1, s.t0, 0.f);
    s.t1 = sel(ln.j2, s.t1, 0.f);
    s.t2 = sel(ln.j3, s.t2, 0.f);
  } else if constexpr (STEP == L + 4) {
    // rows of E = (unit lower triangle of the panel)^-1 in the quad: lane kk holds E[kk][0..2]
    const float a = fma_quad_bcast<1>(s.t0, s.t1, s.t0);           // lane 1: m10, lane 2: e20, lane 3: m31 m10 + m30
    s.e0 = sel(ln.j0, 1.0f, fma_quad_bcast<2>(a, s.t2, a));          // lane 3: m32 e20 + m31 m10 + m30 = e30
    s.e1 = fma_quad_bcast<2>(s.t1, s.t2, s.t1) + ln.d1;              // lane 1: 1, lane 2: m21, lane 3: m32 m21 + m31 = e31
  } else if constexpr (STEP == L + 5) {
    s.e2 = s.t2 + ln.d2;                                              // lane 2: 1, lane 3: m32
    float rsk = sel(ln.j3, s.rs3, sel(ln.j2, s.rs2, sel(ln.j1, s.rs1, s.rs0)));
    if constexpr (DYN) rsk = sel(p0 + (ln.c & 3) < f, rsk, 0.f);  // a pivot past f (short last panel) eliminates nothing
    s.rsk = rsk;
  } else if constexpr (STEP == L + 6) {
    // quad lane kk of lane group q -> every lane of lane group kk
    const int src = 4 * (20 * q + ln.kk);
    s.e0 = bperm(src, s.e0);
    s.e1 = bperm(src, s.e1);
    s.e2 = bperm(src, s.e2);
    s.rsk = bperm(src, s.rsk);
  } else if constexpr (STEP == L + 7) {
    // the back substitution runs on the rows of -U that stay in the accumulators: it wants 1 / (-u_kk) (all 16 lanes of
    // a group write the same value to the same word; p0 + kk < (f + 3) & ~3 always, a pivot past f is never read).
    // Rows of a fourth panel above the last block row are read as rows of W = -U / sqrt(u_kk) instead (lu_wave_blocked: the side
    // rows), whose diagonal is -sqrt(u_kk) = -1 / rsk.
    // (One writer per lane group -- the other lanes on words of their own -- measured the same: ab_side_rows.txt.)
    rdiag[p0 + ln.kk] = (q == 3 && Ip < NB - 1) ? -s.rsk : -(s.rsk * s.rsk);
  } else {
    constexpr int b = Ip + STEP - (L + 8);
#if CUMF_ABLATE
    // profiling build: 8192 = the eliminated rows of the blocks right of the diagonal tile by ONE fp32 MFMA on the
    // accumulator registers (what a stride-4 pivot order would issue; wrong values here)
    if ((dbg & 8192) && b > Ip) {
      const f32x4 z = __builtin_amdgcn_mfma_f32_16x16x4f32(s.e0 * s.rsk, acc[tile_of<NB>(Ip, b)][q], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      w[b] = z[0];
      return;
    }
#endif
    const float un = fmaf(ln.e3c, s.R[b][3], fmaf(s.e2, s.R[b][2], fmaf(s.e1, s.R[b][1], s.e0 * s.R[b][0])));  // -u_k at block b
    w[b] = un * s.rsk;  // the sign is immaterial: w meets itself
    if constexpr (b == Ip) wm = sel(ln.c > 4 * q + ln.kk, w[b], 0.f);  // rows at or above the pivot stay
  }
}

// exact three-way split of (a, b) into packed bf16 pairs (a in the low half), as split_micro does for the gathered rows
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& H, unsigned& M, unsigned& Lw) {
  H = pack_bf16(a, b);
  const float ra = sub_bf16_lo(a, H), rb = sub_bf16_hi(b, H);
  M = pack_bf16(ra, rb);
  const float ta = sub_bf16_lo(ra, M), tb = sub_bf16_hi(rb, M);
  Lw = pack_bf16(ta, tb);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16_k16(u32x2 a, u32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4, a), __builtin_bit_cast(bf16x4, b), c, 0, 0, 0);
}

}  // namespace cumf
