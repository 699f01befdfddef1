// als_internal.h -- declarations shared by the HIP kernels and the host side of libALS.so.
#ifndef CUMF_ALS_INTERNAL_H_
#define CUMF_ALS_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstddef>

// Profiling build (make ablate -> libALS_ablate.so): the kernels additionally honour KernelArgs::dbg, switches that
// make the results WRONG on purpose (no solve, no Gram pass, ...) to time parts of a kernel.  The production
// library compiles none of it and exports no way to set it.
#ifndef CUMF_ABLATE
#define CUMF_ABLATE 0
#endif

namespace cumf {

constexpr int kThreads = 256;   // 4 waves of 64
constexpr int kStage = 32;      // gathered factor rows per LDS stage
constexpr int kVecLd = 128;     // pitch of the CG vectors in LDS (fused solve needs f <= 128)
constexpr int kMaxFusedNB = 9;  // f <= 128  ->  NB = f / 16 + 1 <= 9
constexpr int kMaxF = 207;      // NB <= 13: the range of the tile kernels (fused paths, MFMA Gram)
constexpr int kMaxFAny = 512;   // round 6: above kMaxF the reference's unfused data flow on two plain kernels (als_generic.hip)
constexpr int kMaxWaveNB = 7;   // wave-per-item kernels (als_wave.hip): f <= 111; register budget of one wave
// two-wave kernel: LU in place up to this NB (CG: always).  Round 5 measured 13 (f = 200 solved in place by the two Gram
// waves): Theta side 82.4 vs 74.4 ms through the tile buffer (profiles/r05/lu_rows_ab.txt)
constexpr int kMaxFusedLuWaveNB = 9;

enum { kModeCG = 0, kModeLU = 1, kModeMaterialize = 2, kModeLUExact = 3, kModeCGHalf = 4 };  // CGHalf: A stored as fp16

// Feature blocks of 16 including the slot that carries the rating value (RHS).
__host__ __device__ constexpr int nb_for_f(int f) { return f / 16 + 1; }
// LDS system matrix G: f rows, column f = RHS.  CG reads rows with 16-byte loads (pitch a
// multiple of 4 floats); the LU paths walk columns (odd pitch = conflict-free).
__host__ __device__ constexpr int solve_ldg(int f, int mode) { return mode == kModeCG ? ((f + 1 + 3) & ~3) : (f + 1); }
__host__ __device__ constexpr size_t solve_g_floats(int f, int mode) {
  return ((size_t)f * solve_ldg(f, mode) + 3) & ~(size_t)3;
}
// Register LU (fast path): packed upper-triangular row store (als_kernels.hip: lu_row_off) + f
// pivot reciprocals.  f = 100: 29 520 B, below the 32 256 B of the stage buffers it aliases
// -> 5 workgroups per CU.
__host__ __device__ constexpr size_t lu_packed_floats(int nb) { return (size_t)256 * nb * (nb + 1) / 2 + 16 * nb; }
// + 2 x 16 multipliers of the panel owner + 16 zeros (accumulator LU)
__host__ __device__ constexpr size_t lu_lds_floats(int nb, int f) { return lu_packed_floats(nb) + (size_t)((f + 3) & ~3) + 48; }
constexpr int kCgExtraFloats = 12 * kVecLd;  // 4 per-wave operand copies + 2 x 4 partial mat-vecs
// whole LDS footprint of a solve on a full G: G + CG exchange buffers | G (exact-order LU)
__host__ __device__ constexpr size_t solve_lds_floats(int f, int mode) {
  return solve_g_floats(f, mode) + (mode == kModeCG ? (size_t)kCgExtraFloats : 0);
}

// Which (f, solver) pairs the fused Gram+solve kernel covers.
__host__ __device__ constexpr bool fused_supported(int f, int mode) {
  // CG keeps x, r, p, Ap as two registers per lane (elements lane and lane + 64): f <= 128 exactly, not
  // "NB <= 9" (f = 130..143 has NB = 9 too; found by tests/test_gpu_parity.py::test_doals_large_f)
  return mode == kModeLU ? nb_for_f(f) <= nb_for_f(kMaxF) : (f <= kVecLd && nb_for_f(f) <= kMaxFusedNB);
}

struct KernelArgs {
  // plan items (one workgroup each)
  const int* item_row;
  const long long* item_begin;
  const int* item_len;
  const int* item_slot;
  const int* item_rowlen;
  // chunked rows (reduce kernel)
  const int* mrow_row;
  const int* mrow_slot0;
  const int* mrow_nslots;
  const int* mrow_rowlen;
  float* part;
  // matrix + factors
  const int* colidx;
  const float* val;
  const float* gather;
  float* update;
  // materialise outputs
  float* tt;       // f x f Gram per row; holds _Float16 when tt_half (CUMF_TT_FP16 of als.cu:335-441)
  float* rhs;
  int tt_half;
  int tt_packed;   // tt receives the packed upper triangle, f (f + 1) / 2 floats per row (cumf_get_hermitian_packed)
  // dense_slots: item i (wave kernel) / row i (reduce kernel) of this launch uses slot i of `part` and
  // every item is dumped, none solved in place (batched "Gram -> tiles -> solver kernel" path)
  int dense_slots;
  int whole_only;  // no item of the launch has a slot (the plan has no chunked rows): the LU wave kernel without its dump exit
  long long row_begin;
  int f;
  float lambda;
  int cg_iters;
  int dbg;  // ablation switches; read only by the kernels of the profiling build (-DCUMF_ABLATE=1, libALS_ablate.so)
  // gram mode "fast": `gather` points at the pre-split (h, l) f16 words of the factor table
  // (presplit_f16x2_kernel) and range violations are OR-ed into *fast_flag (bit 0: table, bit 1: ratings)
  int fast_words;
  int* fast_flag;
  // round 6 (kArithPre): `gather` points at the pre-split bf16 h | m | l planes of the factor table (presplit_bf16x3_kernel),
  // rows of pre_pitch bytes.  1: the production form (packed last block, kArithPrePk), 2: the verification form (kArithPre,
  // bit-identical to the in-kernel split)
  int pre_words;
  unsigned pre_pitch;
  // fused train SSE (als.cu:979-991 folded into the Theta update): when not null, every whole-row item of a wave-kernel
  // launch adds sum_u (r_uv - x_u . theta_v)^2 of its row to sse_bins[item % kSseBins] (fp64 atomics)
  double* sse_bins;
  // the fp32 gather table itself (`gather` is replaced by the pre-split planes / f16 words of a launch that uses them): what
  // the Gram-free CG of short rows reads (als_short.hip)
  const float* gather_f32;
  // the six-product forms throughout (CUMF_PRESPLIT_OFF / CUMF_PRESPLIT_VERIFY: what the bit-identity tests compare): no packed
  // last block in the in-kernel split either (kArithSplitPk, als_wave.hip)
  int no_pack;
};
constexpr int kSseBins = 1024;

// Work lists of a plan as the launchers see them (device arrays of als_plan.cpp).
struct PlanLists {
  // all items, longest first (fused kernels)
  long n_items, n_mrows;
  // chunk items only (slot >= 0) and whole-row items only (slot < 0), each longest first
  long n_citems, n_witems;
  const int *c_row, *c_len, *c_slot, *c_rowlen;
  const long long* c_begin;
  const int *w_row, *w_len, *w_rowlen;
  const long long* w_begin;
  float* part2;      // dense-slot tile buffer of the batched path (lazily allocated), part2_rows slots
  long part2_rows;
  double chunk_share;  // share of the plan's ratings that sits in chunked rows
  long n_short;        // whole rows of at most kShortRow ratings: the LAST n_short items of the full list (and of the w list)
};
// Rows this short have a CG of their own that never forms the Gram matrix (als_short.hip); plans list them last.
constexpr int kShortRow = 32;
bool short_cg_available(int f);
hipError_t launch_short_cg(const KernelArgs& a, long n_items, hipStream_t stream);
hipError_t launch_half_iteration(const KernelArgs& a, int mode, long n_items, long n_mrows, hipStream_t stream,
                                 const PlanLists* lists = nullptr);
hipError_t launch_solve_batched(const float* A, const float* b, float* x, long batch, int f, int mode, int cg_iters,
                                hipStream_t stream);
// Gram arithmetic of the fused / materialising passes.
//   kGramAuto:  fp32 values split exactly into three bf16 terms, six bf16 MFMA products per fp32
//               product, fp32 accumulation (als_wave.hip; fp32-class error, not bit-identical to a
//               fmaf chain); used where the wave kernels exist (all solvers and materialise, 16 <= f <= 207),
//   kGramExact: v_mfma_f32_16x16x4_f32, bit-identical to the reference thread's fmaf chain.
//   kGramFast (opt-in): the factor table pre-split into (h, l) f16 pairs of 4096 x, three f16 MFMA
//               products per fp32 product (22 significand bits); fused LU / CG passes of the wave
//               kernels only, everything else as kGramAuto; values must stay below 15.99 in magnitude.
enum { kGramAuto = 0, kGramExact = 1, kGramFast = 2 };
void set_gram_mode(int mode);
int gram_mode();
bool wave_path_available(int f, int mode);
// the f >= 112 path: two waves per item form the Gram; whole rows are solved in place (CG always, LU up to NB = 9),
// larger LUs and the materialising pass go Gram -> tiles (dense slots of the pooled tile buffer) -> reduce kernel
bool wave_batched_path(int f, int mode);
// unpack = 0: full (batch x f x f) -> packed (batch x f(f+1)/2); unpack = 1: `full` is the packed input,
// `packed` receives the mirrored full matrices
hipError_t launch_presplit(const float* src, unsigned* dst, size_t n, int* flag, hipStream_t stream);
// f > kMaxF (als_generic.hip): the fp32 f x f Gram batch + right-hand sides of the plan's items (whole rows), and the batched
// unpivoted LU in global memory (A is overwritten with the factors, as cublasSgetrfBatched does)
hipError_t launch_gram_generic(const KernelArgs& a, long n_items, hipStream_t stream);
hipError_t launch_lu_global(float* A, const float* b, float* x, long batch, int f, hipStream_t stream);
// kArithPre: which (f, NB) have kernels on the pre-split bf16x3 table (als_wave.hip: CUMF_WAVE_PRE, presplit_shape_ok), the
// table's row pitch in bytes, and the kernel that writes it
// One-wave kernels (NB = 5, 7): strip of f % 16 in {0, 4} features; two-wave kernels (NB = 8 .. 13): f % 16 in {0, 4, 8}.
__host__ __device__ constexpr bool presplit_supported(int f) {
  return ((nb_for_f(f) == 7 || nb_for_f(f) == 5) && ((f & 15) == 0 || (f & 15) == 4)) ||
         (nb_for_f(f) > kMaxWaveNB && nb_for_f(f) <= nb_for_f(kMaxF) && (f & 3) == 0 && (f & 15) <= 8);
}
// ... and where CUMF_PRESPLIT_AUTO uses them: NB = 7 only.  At NB = 5 (f = 64) the 19 KB stage image costs the third wave per
// SIMD that the 160-register kernel otherwise gets (10 KB of dword chunks): measured SLOWER, Netflix f = 64 LU Theta side
// 5.6-5.9 -> 6.4-6.5 ms (profiles/r06/ab_presplit_f64_lu.txt); the kernels stay for CUMF_PRESPLIT_ON and the tests.
__host__ __device__ constexpr bool presplit_pays(int f) { return presplit_supported(f) && nb_for_f(f) >= 7; }
// ... and regardless of the table's size from NB = 10 on (f >= 144): there the two-wave kernel is bound by its matrix-pipe and
// split work, not by the gather -- Netflix f = 200, X side (a 384 MB table, 584 MB of planes): 28.8 -> 26.3 ms (CG), 29.5 ->
// 26.8 (LU); at f = 128 (NB = 9) the X side gains nothing (profiles/r06/ab_presplit_multi.txt)
__host__ __device__ constexpr bool presplit_pays_any_size(int f) { return presplit_supported(f) && nb_for_f(f) >= 10; }
__host__ __device__ constexpr unsigned presplit_pitch(int f) {
  return 96u * (f / 16) + ((f & 15) ? (nb_for_f(f) > kMaxWaveNB ? 64u : 32u) : 0u);
}
hipError_t launch_presplit3(const float* src, void* dst, long long rows, int f, hipStream_t stream);
hipError_t launch_pack_upper(const float* full, float* packed, long batch, int f, int unpack, hipStream_t stream);
void set_last_error(int code);  // read (and cleared) by cumf_last_error
void set_kernel_timing(bool on);
void note_item_kernel(const void* host_function);  // called by the launchers of the Gram(+solve) kernels
const void* last_item_kernel();
hipError_t last_kernel_ms(float* item_ms, float* reduce_ms);
// sums over every timed launch sequence since the previous call (or since timing was switched on), then resets
hipError_t kernel_ms_since_reset(float* item_ms, float* reduce_ms, int* launches);
hipError_t launch_quadratic_terms(const float* A, const float* b, const float* x, const float* reg, long batch, int f,
                                  double* out, hipStream_t stream);
hipError_t launch_sse(const float* val, const int* row, const int* col, const float* thetaT, const float* XT,
                      long count, int f, int surpass_nan, double* out, hipStream_t stream);

}  // namespace cumf

#define CUMF_HIP_CHECK(call)                                                                          \
  do {                                                                                                \
    hipError_t err__ = (call);                                                                        \
    if (err__ != hipSuccess) {                                                                        \
      fprintf(stderr, "HIP Error:\nFile = %s\nLine = %d\nReason = %s\n", __FILE__, __LINE__,          \
              hipGetErrorString(err__));                                                              \
      return (int)err__;                                                                              \
    }                                                                                                 \
  } while (0)

#endif  // CUMF_ALS_INTERNAL_H_
