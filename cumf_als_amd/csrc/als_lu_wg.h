// Shared by als_kernels.hip (four wave roles per workgroup) and als_wave.hip (two waves per item):
// the LU on MFMA accumulator tiles dealt round-robin to NW wave roles.
#pragma once
#include "als_device.h"
#include "als_lu_blocked.h"

namespace cumf {


// Tile t of the upper triangle lives in accumulator slot t / NW of wave role t % NW: every role keeps a
// similar share of live tiles all through the elimination.
template <int NB, int NW>
struct LuGeo {
  static constexpr int NT = NB * (NB + 1) / 2;
  static constexpr int TPW = (NT + NW - 1) / NW;
  __host__ __device__ static constexpr int tile(int W, int s) { return W + NW * s; }
};
template <int NB, int NW>
using LuAcc = f32x4[LuGeo<NB, NW>::TPW];  // the accumulator tiles of one wave role

// ----------------------------------------------------------------------------------
// LU directly on the MFMA accumulators (the fused kernels' LU path).
//
// After the Gram pass wave role W of NW (4 in the workgroup kernels of als_kernels.hip, 2 in the
// two-wave kernel of als_wave.hip) holds the tiles t = W + NW s of the upper triangle of [A | b] in
// the 16x16x4 C/D layout (lane (kk, c) = (l >> 4, l & 15), register r: element
// (16 I + 4 kk + r, 16 J + c)).  The elimination keeps them there and applies FOUR pivots per
// step as one rank-4 MFMA per tile:
//   1. the lanes holding the panel rows p0 .. p0+3 (block row Ip, lane group kk = q) publish them
//      raw -- updated by all earlier panels -- into a 4-row exchange buffer (double-buffered by panel
//      parity); ONE barrier;
//   2. every wave reads the 4x4 pivot block and eliminates it redundantly (multipliers m_kq,
//      reciprocals 1/u_kk), then per live feature block b >= Ip reads the four raw rows at its
//      column and forms the eliminated row of ITS lane group, ub[b] = U'[kk][16 b + c];
//   3. A operand of tile (I, J) = -ub[I] / u_kk masked to rows below the pivot (by symmetry
//      a_i,pk = u'_k,i), B operand = ub[J]:  acc -= L21 * U12  in one v_mfma_f32_16x16x4_f32;
//   4. the eliminated rows stay where they are: the masked update never touches a row at or above
//      its pivot, so after the last panel the accumulators hold U, and the back substitution reads it
//      through a 16-column LDS window (back_substitute_tiles_wg).
// Compared with lu_solve_reg: no hand-over of the tiles through LDS, a quarter of the barriers,
// the trailing update on the otherwise idle matrix pipe.  Unpivoted Gaussian elimination as
// before (the content of getrfBatched(Pivot = NULL) + getrs); operation order differs from the
// oracle's, parity is by tolerance (tests/test_gpu_parity.py).
// ----------------------------------------------------------------------------------
// Does wave W need the eliminated panel row at feature block b while block row Ip is being
// eliminated?  Yes if one of its live tiles (I >= Ip) has b as its row or column block.
template <int NB, int NW>
__host__ __device__ constexpr bool lu_needs_block(int W, int Ip, int b) {
  constexpr int NT = LuGeo<NB, NW>::NT, TPW = LuGeo<NB, NW>::TPW;
  if (b < Ip) return false;
  for (int s = 0; s < TPW; ++s) {
    const int t = LuGeo<NB, NW>::tile(W, s);
    if (t < NT && tile_I<NB>(t) >= Ip && (tile_I<NB>(t) == b || tile_J<NB>(t) == b)) return true;
  }
  return false;
}
template <int NB, int NW>
__host__ __device__ constexpr bool lu_wave_live(int W, int Ip) {
  for (int b = 0; b < NB; ++b)
    if (lu_needs_block<NB, NW>(W, Ip, b)) return true;
  return false;
}

// Role that holds the diagonal tile of block row Ip.
template <int NB, int NW>
__host__ __device__ constexpr int lu_diag_owner(int Ip) {
  constexpr int NT = LuGeo<NB, NW>::NT, TPW = LuGeo<NB, NW>::TPW;
  for (int W = 0; W < NW; ++W)
    for (int s = 0; s < TPW; ++s) {
      const int t = LuGeo<NB, NW>::tile(W, s);
      if (t < NT && tile_I<NB>(t) == Ip && tile_J<NB>(t) == Ip) return W;
    }
  return 0;
}

// Accumulator slot of the diagonal tile (Ip, Ip) in role W.
template <int NB, int NW>
__host__ __device__ constexpr int lu_diag_slot(int W, int Ip) {
  constexpr int NT = LuGeo<NB, NW>::NT, TPW = LuGeo<NB, NW>::TPW;
  for (int s = 0; s < TPW; ++s) {
    const int t = LuGeo<NB, NW>::tile(W, s);
    if (t < NT && tile_I<NB>(t) == Ip && tile_J<NB>(t) == Ip) return s;
  }
  return 0;
}

// LDS of lu_solve_mfma (round 2): [exchange: 2 x 4 panel rows x 16 NB][window: 16 NB rows x 17]
// [pivot reciprocals][2 x 16 multipliers][16 zeros].  f = 200: 24 KB instead of the 95 KB packed row
// store of round 1 (one workgroup per CU); f = 100: 12.6 KB instead of 29.5 KB.
template <int NB>
struct LuLds {
  static constexpr int kXRow = 16 * NB;            // one published panel row
  static constexpr int kX = 2 * 4 * kXRow;         // double-buffered by panel parity
  static constexpr int kPitch = 17;                // window pitch (odd: lane = row reads are conflict-free)
  static constexpr int kT = 16 * NB * kPitch;
};
// Round 6, lu_solve_blocked_wg (below): the h | m | l bf16 planes of the w of one block row, one u32x2 per lane, plane and
// feature block; the back-substitution window aliases them (it is written after the last trailing update).
#ifndef CUMF_WG_LU_BLOCKED_NB
#define CUMF_WG_LU_BLOCKED_NB 8  // systems of NB >= this many feature blocks take the blocked elimination (99: none)
#endif
__host__ __device__ constexpr bool lu_wg_blocked(int nb) { return nb >= CUMF_WG_LU_BLOCKED_NB; }
template <int NB>
struct LuBlkLds {
  static constexpr int kPlane = 128;                                      // dwords of one plane of one block: 64 lanes x u32x2
  static constexpr int kPL = 3 * kPlane * NB;
  static constexpr int kU = kPL > LuLds<NB>::kT ? kPL : LuLds<NB>::kT;  // planes / window
};
template <int NB>
__host__ __device__ constexpr size_t lu_wg_lds_floats(int f) {
  // + pivot reciprocals, 2 x 16 multipliers, 16 zeros, 16 spare (train SSE)
  return (size_t)LuLds<NB>::kX + (lu_wg_blocked(NB) ? LuBlkLds<NB>::kU : LuLds<NB>::kT) + ((f + 3) & ~3) + 64;
}

// Back substitution U x = y by the workgroup, straight from the accumulator tiles of the four roles
// through the LDS window (same recurrence as back_substitute_zeroed): per 16-pivot block column kb every
// role writes its tiles (I, kb), I <= kb, to the window (entries at and left of the diagonal as zeros),
// barrier, wave 0 (lane i = rows i, i + 64, ...) reads the 16 entries of its rows and runs the 16 steps,
// barrier.  2 (NB) barriers instead of a 95 KB row store.
template <int NB, int W, int NQ, int NW>
__device__ __forceinline__ float back_substitute_tiles_wg(const LuAcc<NB, NW>& acc, float* __restrict__ T,
                                                         const float* __restrict__ rdiag,
                                                         const float* __restrict__ zpad, int f,
                                                         float* __restrict__ x_global, int lane) {
  constexpr int NT = LuGeo<NB, NW>::NT, TPW = LuGeo<NB, NW>::TPW, P = LuLds<NB>::kPitch;
  const int c = lane & 15, g = lane >> 4;
  const int top = f - 1;
  float z[NQ], rdl[NQ];
  const float* rowp[NQ];
  int ib[NQ];
  if constexpr (W == 0) {
    static_for<NQ>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int i = lane + 64 * q;
      const int ic = i < f ? i : f - 1;
      ib[q] = i < f ? (i >> 4) : 1 << 20;  // rows past f never take part
      rowp[q] = T + ic * P;
      rdl[q] = i < f ? rdiag[ic] : 0.f;
      z[q] = 0.f;
    });
  }
  static_for<NB>([&](auto bc) {
    constexpr int kb = NB - 1 - decltype(bc)::value;
    constexpr int Q = kb >> 2;  // pivots of this block live in z[Q]
    // this role's tiles of block column kb -> window
    static_for<TPW>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int t = LuGeo<NB, NW>::tile(W, s);
      if constexpr (t < NT) {
        if constexpr (tile_J<NB>(t) == kb) {
          constexpr int I = tile_I<NB>(t);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[s][r];
            if constexpr (I == kb) v = (c > 4 * g + r) ? v : 0.f;
            T[(16 * I + 4 * g + r) * P + c] = v;
          }
        }
      }
    });
    __syncthreads();
    if constexpr (W == 0 && Q < NQ) {
      if constexpr (kb == NB - 1) {  // y = column f of the last block column
        static_for<NQ>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          z[q] = rowp[q][f - 16 * (NB - 1)] * rdl[q];
        });
      }
      if (16 * kb <= top) {  // uniform: the last block column may hold nothing but y
        float col[16][Q + 1];
        static_for<Q + 1>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const float* base = (ib[q] > kb) ? zpad : rowp[q];
          static_for<16>([&](auto jc) { col[decltype(jc)::value][q] = base[decltype(jc)::value]; });
        });
        static_for<16>([&](auto jc) {
          constexpr int j = 15 - decltype(jc)::value;
          const int k = 16 * kb + j;
          if (k <= top) {  // uniform; only the last block can be short
            const float xk = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[Q]), k & 63));
            static_for<Q + 1>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              z[q] = fmaf(-(col[j][q] * rdl[q]), xk, z[q]);
            });
          }
        });
      }
    }
    if constexpr (kb > 0) __syncthreads();  // the window is rewritten for the next block column
  });
  float ssq = 0.f;  // wave 0: this lane's share of ||x||^2 (rows past f hold zeros)
  if constexpr (W == 0) {
    static_for<NQ>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if (lane + 64 * q < f) x_global[lane + 64 * q] = z[q];
      ssq = fmaf(z[q], z[q], ssq);
    });
  }
  return ssq;
}

// sse_bins != nullptr: the train SSE of the row for free, as in lu_wave_blocked (als_wave.hip, wave_tile_ff): entry (f, f)
// of the eliminated system is the Schur complement sum r^2 + reg - b^T A^-1 b, so SSE = (f, f) - reg (1 + |x|^2); the role
// that owns the last diagonal tile hands (f, f) to wave 0 through LDS.
template <int NB, int W, int NW = 4>
__device__ __forceinline__ void lu_solve_mfma(LuAcc<NB, NW>& acc, float* __restrict__ lds, int f, float reg,
                                              float* __restrict__ x_global, int tid, double* sse_bins = nullptr,
                                              int rowlen = 0) {
  constexpr int NT = LuGeo<NB, NW>::NT, TPW = LuGeo<NB, NW>::TPW;
  const int lane = tid & 63, c = lane & 15, kk = lane >> 4;
  float* X = lds;                                 // published raw panel rows, [parity][column 16 J + c][r]: the four rows of a column side by side (one 16-byte access)
  float* Twin = lds + LuLds<NB>::kX;              // back-substitution window
  float* rdiag = Twin + LuLds<NB>::kT;            // pivot reciprocals
  // lambda * n_u on the diagonal (als.cu:545-557)
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = LuGeo<NB, NW>::tile(W, s);
    if constexpr (t < NT) {
      if constexpr (tile_I<NB>(t) == tile_J<NB>(t)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * kk + r == c) acc[s][r] += reg;
      }
    }
  });
  float* ctab = rdiag + ((f + 3) & ~3);  // 2 x 16 floats, 16-byte aligned
  float* zpad = ctab + 32;               // 16 zeros (back substitution: rows outside a pivot block read these)
  if constexpr (W == 0) {
    if (lane < 16) zpad[lane] = 0.f;
  }
  static_for<NB>([&](auto ipc) {
    constexpr int Ip = decltype(ipc)::value;
    for (int q = 0; q < 4; ++q) {
      const int p0 = 16 * Ip + 4 * q;
      if (p0 >= f) break;
      // 1. publish the raw panel rows (tiles of block row Ip, lane group q)
      static_for<TPW>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int t = LuGeo<NB, NW>::tile(W, s);
        if constexpr (t < NT) {
          if constexpr (tile_I<NB>(t) == Ip) {
            constexpr int J = tile_J<NB>(t);
            if (kk == q)  // the four rows of a column side by side: ONE 16-byte store here, one 16-byte load per reader
              *reinterpret_cast<f32x4*>(X + ((p0 >> 2) & 1) * 4 * LuLds<NB>::kXRow + 4 * (16 * J + c)) = acc[s];
          }
        }
      });
      // 2a. the role that owns the diagonal tile of block row Ip has just written the 4x4 pivot
      // block: it alone eliminates it (four dependent reciprocals, no division) and leaves, per
      // lane group kk, the composite multipliers (c0, c1, c2) of panel row kk and -1/u_kk in a
      // 16-float table (double-buffered by panel parity) -- the other roles just read their line.
      // Row k of the panel after the elimination is raw_k + sum_{q<k} e_kq raw_q with e = the rows
      // of the inverse of the panel's unit lower triangle.
      const bool vk = p0 + kk < f;  // this lane group's pivot exists (short last panel otherwise)
      float* tab = ctab + 16 * ((p0 >> 2) & 1);
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, nrp = 0.f;
      constexpr int OWNER = lu_diag_owner<NB, NW>(Ip);
      if constexpr (W == OWNER) {
        // the 4 x 4 pivot block sits in this role's diagonal tile: rows = registers 0..3 of lane group q,
        // columns = lanes 4q .. 4q+3 of that group: fetch it with v_readlane, no LDS round trip
        constexpr int SD = lu_diag_slot<NB, NW>(OWNER, Ip);
        const int l0 = 20 * q;  // lane of (lane group q, column 4 q)
        auto rl = [&](float v, int l) {
          return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
        };
        const bool v1 = p0 + 1 < f, v2 = p0 + 2 < f, v3 = p0 + 3 < f;
        float P00 = rl(acc[SD][0], l0), P01 = rl(acc[SD][0], l0 + 1), P02 = rl(acc[SD][0], l0 + 2),
              P03 = rl(acc[SD][0], l0 + 3);
        float P11 = rl(acc[SD][1], l0 + 1), P12 = rl(acc[SD][1], l0 + 2), P13 = rl(acc[SD][1], l0 + 3);
        float P22 = rl(acc[SD][2], l0 + 2), P23 = rl(acc[SD][2], l0 + 3);
        float P33 = rl(acc[SD][3], l0 + 3);
        // v_rcp_f32 is accurate to 1 ulp; the four reciprocals are a dependent chain, so no Newton step
        auto recip = [](float d) { return __builtin_amdgcn_rcpf(d); };
        const float rp0 = recip(P00);
        const float m10 = -P01 * rp0, m20 = -P02 * rp0, m30 = -P03 * rp0;  // -(multiplier of row k w.r.t. pivot 0)
        P11 = fmaf(m10, P01, P11);
        P12 = fmaf(m10, P02, P12);
        P13 = fmaf(m10, P03, P13);
        P22 = fmaf(m20, P02, P22);
        P23 = fmaf(m20, P03, P23);
        P33 = fmaf(m30, P03, P33);
        const float rp1 = recip(v1 ? P11 : 1.0f);
        const float m21 = -P12 * rp1, m31 = -P13 * rp1;
        P22 = fmaf(m21, P12, P22);
        P23 = fmaf(m21, P13, P23);
        P33 = fmaf(m31, P13, P33);
        const float rp2 = recip(v2 ? P22 : 1.0f);
        const float m32 = -P23 * rp2;
        P33 = fmaf(m32, P23, P33);
        const float rp3 = recip(v3 ? P33 : 1.0f);
        const float e20 = fmaf(m21, m10, m20);
        const float e31 = fmaf(m32, m21, m31);
        const float e30 = fmaf(m32, e20, fmaf(m31, m10, m30));
        const float rpk = kk == 0 ? rp0 : (kk == 1 ? rp1 : (kk == 2 ? rp2 : rp3));
        c0 = kk == 1 ? m10 : (kk == 2 ? e20 : (kk == 3 ? e30 : 0.f));
        c1 = kk == 2 ? m21 : (kk == 3 ? e31 : 0.f);
        c2 = kk == 3 ? m32 : 0.f;
        nrp = vk ? -rpk : 0.f;
        if (c < 4) tab[4 * kk + c] = c == 0 ? c0 : (c == 1 ? c1 : (c == 2 ? c2 : nrp));
        if (c == 4 && vk) rdiag[p0 + kk] = rpk;
      }
      __syncthreads();
      if constexpr (lu_wave_live<NB, NW>(W, Ip)) {
      const float* xb = X + ((p0 >> 2) & 1) * 4 * LuLds<NB>::kXRow + 4 * c;
      if constexpr (W != OWNER) {
        const f32x4 line = *reinterpret_cast<const f32x4*>(tab + 4 * kk);
        c0 = line[0];
        c1 = line[1];
        c2 = line[2];
        nrp = line[3];
      }
      // 2b. eliminated panel row of this lane group at every live block: three FMAs per block.  Round 5: the four raw
      // rows of a column arrive in ONE ds_read_b128 (4 LDS cycles, against 4 x 2 for four ds_read_b32 and a quarter of the
      // instructions: this LU is bound by the CU's LDS pipe -- 4 660 LDS instructions per 200 x 200 system, round 3)
      const bool k1 = kk == 1, k2 = kk == 2, k3 = kk == 3;
      float ub[NB];
      static_for<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (lu_needs_block<NB, NW>(W, Ip, b)) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(xb + 64 * b);
          const float a0 = rr[0], a1 = rr[1], a2 = rr[2];
          float own = a0;  // this lane group's own raw row (flat selects)
          own = k1 ? a1 : own;
          own = k2 ? a2 : own;
          own = k3 ? rr[3] : own;
          // lanes of a pivot past f (short last panel) keep a finite dummy: their A operand is 0
          ub[b] = fmaf(c2, a2, fmaf(c1, a1, fmaf(c0, a0, own)));
        }
      });
      // 3. rank-4 update of the live tiles
      static_for<TPW>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int t = LuGeo<NB, NW>::tile(W, s);
        if constexpr (t < NT) {
          constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
          if constexpr (I >= Ip) {
            float la = ub[I] * nrp;
            if constexpr (I == Ip) la = (c > 4 * q + kk) ? la : 0.f;  // rows at or above the pivot stay
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(la, ub[J], acc[s], 0, 0, 0);
          }
        }
      });
      }  // lu_wave_live
    }
  });
  // the eliminated rows stay in the accumulators (the masked update leaves rows at and above a pivot alone)
  float* ff_slot = zpad + 16;
  if constexpr ((NT - 1) % NW == W) {
    if (sse_bins != nullptr) {
      const float ff = wave_tile_ff<NB>(acc[(NT - 1) / NW], f);
      if (lane == 0) ff_slot[0] = ff;
    }
  }
  __syncthreads();  // rdiag is complete
  const float ssq = back_substitute_tiles_wg<NB, W, (16 * NB + 63) / 64, NW>(acc, Twin, rdiag, zpad, f, x_global, lane);
  if constexpr (W == 0) {
    if (sse_bins != nullptr) {
      const float tt = wave_sum_uniform(ssq);
      if (lane == 0 && rowlen > 0)
        atomicAdd(sse_bins + (blockIdx.x & (kSseBins - 1)), (double)ff_slot[0] - (double)reg * (1.0 + (double)tt));
    }
  }
}

// ----------------------------------------------------------------------------------
// Round 6: the same elimination BLOCKED by block rows, as lu_wave_blocked (als_wave.hip) runs it inside one wave -- here over
// NW wave roles.  lu_solve_mfma applies every four-pivot panel to ALL live tiles with fp32 MFMAs (36 cycles each, nothing
// issues beside them): 1 820 per 200 x 200 system, and every role forms the eliminated panel row at every feature block of
// its tiles, panel after panel (13 x 7 VALU + 13 LDS reads per panel and role at the start).  Blocked:
//   per panel       only the roles that hold a tile of block row Ip work: eliminated row of their lane group at the diagonal
//                   block and at their own blocks, scaled to w = u / sqrt(u_kk), ONE fp32 MFMA per tile of the block row
//                   (364 instead of 1 820 at NB = 13); an odd panel splits the w of its pair (q - 1, q) exactly into three
//                   bf16 terms and leaves them in LDS in MFMA operand layout (K slot (lane group g, element e) = pivot
//                   4 e + g: where w of panel e sits, no data movement);
//   per block row   one barrier, then every tile below the block row takes its rank-16 update as six
//                   v_mfma_f32_16x16x16_bf16 (the six products of the Gram pass), operands straight from LDS.
// The accumulators hold -A (the update is the positive product w w^T); the rows of -U stay in them for the back substitution.
// Parity by tolerance, as for lu_wave_blocked.  LDS: planes 1.5 KB per feature block (aliased by the window later).
// ----------------------------------------------------------------------------------
template <int NB, int NW>
__host__ __device__ constexpr int lu_row_slot(int W, int Ip, int b) {  // slot of tile (Ip, b) in role W; -1: another role's
  if (b < Ip || b >= NB) return -1;
  const int t = tile_of<NB>(Ip, b);
  return t % NW == W ? t / NW : -1;
}
template <int NB, int NW>
__host__ __device__ constexpr bool lu_role_in_row(int W, int Ip) {
  for (int b = Ip; b < NB; ++b)
    if (lu_row_slot<NB, NW>(W, Ip, b) >= 0) return true;
  return false;
}

template <int NB, int W, int NW = 4, bool NEGATED = false>
__device__ __forceinline__ void lu_solve_blocked_wg(LuAcc<NB, NW>& acc, float* __restrict__ lds, int f, float reg,
                                                    float* __restrict__ x_global, int tid, double* sse_bins = nullptr,
                                                    int rowlen = 0) {
  constexpr int NT = LuGeo<NB, NW>::NT, TPW = LuGeo<NB, NW>::TPW;
  const int lane = tid & 63, c = lane & 15, kk = lane >> 4;
  float* X = lds;                                     // published raw panel rows: [parity][column 16 J + c][r]
  float* Twin = lds + LuLds<NB>::kX;                  // back-substitution window ...
  unsigned* PL = reinterpret_cast<unsigned*>(Twin);   // ... and before it the planes: [block][plane][lane][2]
  float* rdiag = Twin + LuBlkLds<NB>::kU;             // 1 / (-u_kk)
  float* ctab = rdiag + ((f + 3) & ~3);               // 2 x 16 floats, 16-byte aligned
  float* zpad = ctab + 32;
  // the system, negated: -(A + lambda n_u I) (als.cu:545-557 for the diagonal term); NEGATED: the caller hands over -A
  static_for<TPW>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int t = LuGeo<NB, NW>::tile(W, s);
    if constexpr (t < NT) {
      constexpr bool diag = tile_I<NB>(t) == tile_J<NB>(t);
      if constexpr (diag || !NEGATED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = NEGATED ? -acc[s][r] : acc[s][r];
          if constexpr (diag) v = (4 * kk + r == c) ? v + reg : v;
          acc[s][r] = -v;
        }
      }
    }
  });
  if constexpr (W == 0) {
    if (lane < 16) zpad[lane] = 0.f;
  }
  const bool k1 = kk == 1, k2 = kk == 2, k3 = kk == 3;
  static_for<NB>([&](auto ipc) {
    constexpr int Ip = decltype(ipc)::value;
    constexpr int L = NB - Ip;
    constexpr bool IN_ROW = lu_role_in_row<NB, NW>(W, Ip);
    constexpr int OWNER = lu_diag_owner<NB, NW>(Ip);
    float wprev[NB];  // w of the even panel of a pair at this role's blocks
#pragma unroll
    for (int b = 0; b < NB; ++b) wprev[b] = 0.f;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int p0 = 16 * Ip + 4 * q;
      if (p0 >= f) break;  // uniform (only the last block row can be short; nothing lies below it)
      float* Xp = X + (q & 1) * 4 * LuLds<NB>::kXRow;
      // 1. publish the raw panel rows (this role's tiles of block row Ip, lane group q): one 16-byte store per tile
      static_for<L>([&](auto bc) {
        constexpr int b = Ip + decltype(bc)::value;
        constexpr int s = lu_row_slot<NB, NW>(W, Ip, b);
        if constexpr (s >= 0) {
          if (kk == q) *reinterpret_cast<f32x4*>(Xp + 4 * (16 * b + c)) = acc[s];
        }
      });
      // 2a. the owner of the diagonal tile eliminates the 4 x 4 pivot block (of -A: all ratios are those of A) and leaves,
      // per lane group kk, row kk of the inverse of the panel's unit lower triangle times 1 / sqrt(u_kk)
      const bool vk = p0 + kk < f;
      float* tab = ctab + 16 * (q & 1);
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
      if constexpr (W == OWNER) {
        constexpr int SD = lu_diag_slot<NB, NW>(OWNER, Ip);
        const int l0 = 20 * q;  // lane of (lane group q, column 4 q)
        auto rl = [&](float v, int l) {
          return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
        };
        auto rsq = [](float d) { return __builtin_amdgcn_rsqf(d); };
        const bool v1 = p0 + 1 < f, v2 = p0 + 2 < f, v3 = p0 + 3 < f;
        float P00 = rl(acc[SD][0], l0), P01 = rl(acc[SD][0], l0 + 1), P02 = rl(acc[SD][0], l0 + 2),
              P03 = rl(acc[SD][0], l0 + 3);
        float P11 = rl(acc[SD][1], l0 + 1), P12 = rl(acc[SD][1], l0 + 2), P13 = rl(acc[SD][1], l0 + 3);
        float P22 = rl(acc[SD][2], l0 + 2), P23 = rl(acc[SD][2], l0 + 3);
        float P33 = rl(acc[SD][3], l0 + 3);
        const float rs0 = rsq(-P00), rp0 = -(rs0 * rs0);  // 1 / P00
        const float m10 = -P01 * rp0, m20 = -P02 * rp0, m30 = -P03 * rp0;
        P11 = fmaf(m10, P01, P11);
        P12 = fmaf(m10, P02, P12);
        P13 = fmaf(m10, P03, P13);
        P22 = fmaf(m20, P02, P22);
        P23 = fmaf(m20, P03, P23);
        P33 = fmaf(m30, P03, P33);
        const float rs1 = rsq(v1 ? -P11 : 1.0f), rp1 = -(rs1 * rs1);
        const float m21 = -P12 * rp1, m31 = -P13 * rp1;
        P22 = fmaf(m21, P12, P22);
        P23 = fmaf(m21, P13, P23);
        P33 = fmaf(m31, P13, P33);
        const float rs2 = rsq(v2 ? -P22 : 1.0f), rp2 = -(rs2 * rs2);
        const float m32 = -P23 * rp2;
        P33 = fmaf(m32, P23, P33);
        const float rs3 = rsq(v3 ? -P33 : 1.0f);
        const float e20 = fmaf(m21, m10, m20);
        const float e31 = fmaf(m32, m21, m31);
        const float e30 = fmaf(m32, e20, fmaf(m31, m10, m30));
        const float rs = k1 ? rs1 : (k2 ? rs2 : (k3 ? rs3 : rs0));
        const float sc = vk ? rs : 0.f;  // a pivot past f (short last panel) eliminates nothing
        // row kk of E = (unit lower triangle of the panel)^-1, scaled by 1 / sqrt(u_kk): w = sum_r e_r raw_r
        e0 = (k1 ? m10 : (k2 ? e20 : (k3 ? e30 : 1.0f))) * sc;
        e1 = (k1 ? 1.0f : (k2 ? m21 : (k3 ? e31 : 0.f))) * sc;
        e2 = (k2 ? 1.0f : (k3 ? m32 : 0.f)) * sc;
        e3 = k3 ? sc : 0.f;
        if (c < 4) tab[4 * kk + c] = c == 0 ? e0 : (c == 1 ? e1 : (c == 2 ? e2 : e3));
        if (c == 4 && vk) rdiag[p0 + kk] = -(rs * rs);  // the back substitution runs on the rows of -U
      }
      __syncthreads();
      if constexpr (IN_ROW) {
        const float* xb = Xp + 4 * c;
        if constexpr (W != OWNER) {
          const f32x4 line = *reinterpret_cast<const f32x4*>(tab + 4 * kk);
          e0 = line[0];
          e1 = line[1];
          e2 = line[2];
          e3 = line[3];
        }
        // 2b. w of this lane group's pivot at the diagonal block and at this role's blocks
        float wv[NB];
        static_for<L>([&](auto bc) {
          constexpr int b = Ip + decltype(bc)::value;
          if constexpr (b == Ip || lu_row_slot<NB, NW>(W, Ip, b) >= 0) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(xb + 64 * b);
            wv[b] = fmaf(e3, rr[3], fmaf(e2, rr[2], fmaf(e1, rr[1], e0 * rr[0])));
          }
        });
        // 3. the block row's own tiles: what its next panel reads (rows at or above the pivot stay)
        const float wm = (c > 4 * q + kk) ? wv[Ip] : 0.f;
        static_for<L>([&](auto bc) {
          constexpr int b = Ip + decltype(bc)::value;
          constexpr int s = lu_row_slot<NB, NW>(W, Ip, b);
          if constexpr (s >= 0) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm, wv[b], acc[s], 0, 0, 0);
        });
        // 4. planes of the pair (q - 1, q) for the trailing update
        if constexpr (L > 1) {
          if (q & 1) {
            static_for<L - 1>([&](auto bc) {
              constexpr int b = Ip + 1 + decltype(bc)::value;
              if constexpr (lu_row_slot<NB, NW>(W, Ip, b) >= 0) {
                unsigned H, M, Lw;
                split3_pair(wprev[b], wv[b], H, M, Lw);
                unsigned* pl = PL + (3 * b * 64 + lane) * 2 + (q >> 1);
                pl[0] = H;
                pl[128] = M;
                pl[256] = Lw;
              }
            });
          } else {
            static_for<L - 1>([&](auto bc) {
              constexpr int b = Ip + 1 + decltype(bc)::value;
              if constexpr (lu_row_slot<NB, NW>(W, Ip, b) >= 0) wprev[b] = wv[b];
            });
          }
        }
      }
    }
    // 5. rank-16 update of every tile below the block row
    if constexpr (L > 1) {
      __syncthreads();
      auto plane = [&](int b, int p) { return *reinterpret_cast<const u32x2*>(PL + ((3 * b + p) * 64 + lane) * 2); };
      u32x2 hI = {0u, 0u}, mI = {0u, 0u}, lI = {0u, 0u};
      static_for<TPW>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int t = LuGeo<NB, NW>::tile(W, s);
        if constexpr (t < NT) {
          constexpr int I = tile_I<NB>(t), J = tile_J<NB>(t);
          if constexpr (I > Ip) {
            constexpr bool new_row = s == 0 || tile_I<NB>(t - (s == 0 ? 0 : NW)) != I;
            if constexpr (new_row) {
              hI = plane(I, 0);
              mI = plane(I, 1);
              lI = plane(I, 2);
            }
            u32x2 hJ = hI, mJ = mI, lJ = lI;
            if constexpr (J != I) {
              hJ = plane(J, 0);
              mJ = plane(J, 1);
              lJ = plane(J, 2);
            }
            acc[s] = mfma_bf16_k16(lI, hJ, acc[s]);  // small terms first, as in the Gram pass
            acc[s] = mfma_bf16_k16(hI, lJ, acc[s]);
            acc[s] = mfma_bf16_k16(mI, mJ, acc[s]);
            acc[s] = mfma_bf16_k16(mI, hJ, acc[s]);
            acc[s] = mfma_bf16_k16(hI, mJ, acc[s]);
            acc[s] = mfma_bf16_k16(hI, hJ, acc[s]);
#ifndef CUMF_WG_LU_NO_SCHED  // keeps the plane reads of a tile next to its MFMAs: 43 spilled registers instead of 233 at 128 (four workgroups per CU)
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        }
      });
    }
  });
  float* ff_slot = zpad + 16;
  if constexpr ((NT - 1) % NW == W) {
    if (sse_bins != nullptr) {
      const float ff = -wave_tile_ff<NB>(acc[(NT - 1) / NW], f);  // the tiles hold the negated system
      if (lane == 0) ff_slot[0] = ff;
    }
  }
  __syncthreads();  // rdiag is complete; the planes are dead (the window aliases them)
  const float ssq = back_substitute_tiles_wg<NB, W, (16 * NB + 63) / 64, NW>(acc, Twin, rdiag, zpad, f, x_global, lane);
  if constexpr (W == 0) {
    if (sse_bins != nullptr) {
      const float tt = wave_sum_uniform(ssq);
      if (lane == 0 && rowlen > 0)
        atomicAdd(sse_bins + (blockIdx.x & (kSseBins - 1)), (double)ff_slot[0] - (double)reg * (1.0 + (double)tt));
    }
  }
}

// the LU of the workgroup kernels: blocked from CUMF_WG_LU_BLOCKED_NB feature blocks on
template <int NB, int W, int NW = 4, bool NEGATED = false>
__device__ __forceinline__ void lu_solve_wg(LuAcc<NB, NW>& acc, float* __restrict__ lds, int f, float reg,
                                            float* __restrict__ x_global, int tid, double* sse_bins = nullptr, int rowlen = 0) {
  static_assert(!NEGATED || lu_wg_blocked(NB), "only the blocked elimination takes the negated system");
  if constexpr (lu_wg_blocked(NB))
    lu_solve_blocked_wg<NB, W, NW, NEGATED>(acc, lds, f, reg, x_global, tid, sse_bins, rowlen);
  else
    lu_solve_mfma<NB, W, NW>(acc, lds, f, reg, x_global, tid, sse_bins, rowlen);
}

}  // namespace cumf
