/*
 * cumf_als_capi.h -- C ABI of libALS.so, the MI355X-native ALS solve path.
 *
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 * Every entry point names the reference interface (file:line under the
 * cuMF/cumf_als tree) it replaces.  Device pointers are HIP device pointers of
 * the calling process; `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  All entry points return 0 on success and a non-zero HIP
 * error code on failure after printing file/line to stderr; they never fall
 * back to a CPU path.
 */
#ifndef CUMF_ALS_CAPI_H_
#define CUMF_ALS_CAPI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Solver selection: the reference chooses at compile time with `#define USE_CG`
 * (als.cu:28); here it is a run-time argument. */
enum { CUMF_SOLVER_CG = 0, CUMF_SOLVER_LU = 1 };

/*
 * extern "C" alias of `float doALS(...)` (als.h:676-681, als.cu:662-1035).
 * Same 22 arguments, same meaning, same in/out behaviour of thetaTHost / XTHost,
 * same stdout lines (als.cu:670,830,991,1019).  HOST pointers.  Solver, CG
 * iteration count and RMSE compat flags come from the environment
 * (CUMF_ALS_SOLVER=cg|lu, CUMF_ALS_CG_ITERS, CUMF_ALS_EXACT_TEST_GRID,
 * CUMF_ALS_SURPASS_NAN, CUMF_ALS_PATH=fused|unfused) -- see INTEGRATION.md.
 */
float cumf_doALS(const int* csrRowIndexHostPtr, const int* csrColIndexHostPtr, const float* csrValHostPtr,
                 const int* cscRowIndexHostPtr, const int* cscColIndexHostPtr, const float* cscValHostPtr,
                 const int* cooRowIndexHostPtr, float* thetaTHost, float* XTHost,
                 const int* cooRowIndexTestHostPtr, const int* cooColIndexTestHostPtr,
                 const float* cooValHostTestPtr, const int m, const int n, const int f, const long nnz,
                 const long nnz_test, const float lambda, const int ITERS, const int X_BATCH,
                 const int THETA_BATCH, const int DEVICEID);

/* doALS with the compile-time switches of als.cu:25-33 exposed as arguments.
 * rmse_log (may be NULL): 2*ITERS floats, (train, test) per iteration. */
float cumf_doALS_ex(const int* csrRowIndexHostPtr, const int* csrColIndexHostPtr, const float* csrValHostPtr,
                    const int* cscRowIndexHostPtr, const int* cscColIndexHostPtr, const float* cscValHostPtr,
                    const int* cooRowIndexHostPtr, float* thetaTHost, float* XTHost,
                    const int* cooRowIndexTestHostPtr, const int* cooColIndexTestHostPtr,
                    const float* cooValHostTestPtr, int m, int n, int f, long nnz, long nnz_test,
                    float lambda, int ITERS, int X_BATCH, int THETA_BATCH, int DEVICEID,
                    int solver, int cg_iters, int fused, int exact_test_grid, int surpass_nan,
                    int quiet, float* rmse_log);

/* ------------------------------------------------------------------------
 * Half-iteration plan: the static work decomposition of one side of the
 * factorisation (update X over CSR rows, or update Theta over CSC columns).
 * Replaces the per-iteration launch bookkeeping of als.cu:760-855 / 876-964:
 * rows are cut into chunks of at most `chunk` ratings so that a heavy row
 * (Netflix X side: up to ~230k ratings) is spread over many workgroups instead
 * of one CUDA block per row (als.cu:449).
 * ------------------------------------------------------------------------ */
typedef struct cumf_plan cumf_plan_t;

/* rowptr_host: HOST row pointers of the rows [0, rows] (int32, or int64 when
 * rowptr_is_64 != 0 -- hugewiki.cu:2266 keeps them unsigned for nnz > 2^31).
 * Only rows [row_begin, row_end) are planned (the X_BATCH / THETA_BATCH slices of
 * als.cu:768-777, 881-890).  chunk <= 0 picks the default for f.  Any even f <= 512: up to f = 207 the tile kernels
 * (fused half-iterations, MFMA Gram); above that -- the reference's generic kernel takes every f % 10 == 0,
 * als.cu:575-659 -- the reference's own data flow on two plain kernels: cumf_get_hermitian (fp32 f x f batch) +
 * cumf_cg_solve_batched / cumf_lu_solve_batched, which is what doALS then runs (slow, correct, in the reference's
 * operation order: Gram and LU are bit-identical to a sequential evaluation). */
int cumf_plan_create(cumf_plan_t** plan, const void* rowptr_host, int rowptr_is_64, long rows,
                     long row_begin, long row_end, int f, int chunk);
int cumf_plan_destroy(cumf_plan_t* plan);
/* counts for tests/diagnostics: [0]=items, [1]=partial slots, [2]=multi-chunk rows, [3]=chunk */
int cumf_plan_info(const cumf_plan_t* plan, long info[4]);

/*
 * Fused half-iteration over the planned rows: RHS + Gram + solve in one pass,
 * Gram never written to HBM.  Replaces, for one batch:
 *   cusparseScsrmm2 + cublasSgeam            (als.cu:750-757 / 867-874)
 *   get_hermitian100 / get_hermitianT10      (als.cu:788-817 / 900-924)
 *   updateXWithCGHost | updateX/updateTheta  (als.cu:831,839 / 941,948)
 * colidx/val: DEVICE CSR arrays of the whole matrix (indexed by the plan's row
 * pointers); gather: DEVICE factors gathered from (cols x f); update: DEVICE
 * factors being solved (rows x f), read as the CG warm start and overwritten.
 * The wave kernels (gram modes auto / fast, 16 <= f <= 207) address `gather` with 64-bit lane addresses: no
 * size limit.  The workgroup kernels (f <= 14, gram mode exact) use 32-bit byte offsets and need a table
 * below 4 GiB; cumf_check_gather_table tells which applies.
 */
/* 1 when cumf_als_update_fused can handle (f, solver) in the current gram mode. */
int cumf_fused_available(int f, int solver);
int cumf_als_update_fused(const cumf_plan_t* plan, const int* colidx, const float* val,
                          const float* gather, float* update, int f, float lambda, int solver,
                          int cg_iters, void* stream);
/*
 * The same half-iteration + the train SSE of the updated rows at no extra pass (replaces the RMSE kernel launch of
 * als.cu:979-991 for the train set when it is the Theta update: the errors r - x_u . theta_v of a column only involve
 * that column's new theta_v and the X it gathered).  The rating rides in slot f of the gathered rows, so the Gram
 * pass also accumulates sum r^2 in entry (f, f) of the augmented system; LU turns that entry into its Schur complement
 * sum r^2 + lambda n - b^T A^-1 b, CG supplies x.b, x.r, |x|^2: sum_u (r - x_u . t)^2 = sum r^2 - 2 t.b + t^T G t
 * follows without reading a rating or a factor row again.  sse_bins: CUMF_SSE_BINS doubles in DEVICE
 * memory, zeroed by the caller, ADDED to (fp64 atomics); the SSE is their sum.  Only when
 * cumf_fused_sse_available(plan, solver) -- wherever the wave kernels' solvers run: 16 <= f <= 207, gram mode not
 * "exact"; not for chunked rows solved by the older workgroup solvers (LU below f = 96, CG at f = 112 .. 128) --
 * otherwise an error, and cumf_sse is the way.  Accuracy: fp32 per column, absolute error <~ 1e-7 of the columns' sum r^2
 * (measured 2e-9 .. 6e-8), i.e. relative error <~ 1e-7 (sum r^2 / SSE): callers that may meet near-perfect fits compare
 * the result with sum r^2 and fall back on cumf_sse below ~1e-3 of it, as doALS does.
 */
#define CUMF_SSE_BINS 1024
int cumf_fused_sse_available(const cumf_plan_t* plan, int solver);
/* The same identity on MATERIALISED systems (the multi-GPU `reduce` scheme: Gram batches reduced across GPUs and solved
 * by cumf_*_solve_batched): *sse_terms (device, fp64) += sum over the batch of 2 x.b - x^T A x + reg[v] |x|^2, so that the
 * squared error over the batch's ratings is (their sum r^2, a constant of the data) minus it.  A: batch x f x f fp32 with
 * reg[v] = lambda n_v on the diagonal; a system WITHOUT ratings (its solution is NaN) is marked by reg[v] < 0 and skipped
 * (reg[v] == 0 is a valid system: lambda = 0). */
int cumf_quadratic_sse_terms(const float* A, const float* b, const float* x, const float* reg, long batch, int f,
                             double* sse_terms, void* stream);
int cumf_als_update_fused_sse(const cumf_plan_t* plan, const int* colidx, const float* val,
                              const float* gather, float* update, int f, float lambda, int solver,
                              int cg_iters, double* sse_bins, void* stream);

/*
 * Materialising Gram + RHS (the reference's data flow): tt receives
 * (row_end-row_begin) x f x f fp32, row-major, both triangles, lambda*n_u on the
 * diagonal -- exactly the `tt` / `xx` buffers of als.cu:782,897 -- and rhs (may be
 * NULL) receives f-contiguous b_u at rhs[(u - row_begin) * f] (ythetaT, als.cu:756).
 * With a slab-local CSC (hugewiki.cu:2332-2340) n_u is the slab-local count, so the
 * partial Grams of several GPUs sum to the full system including lambda * n_u
 * (hugewiki.cu:1187-1687, reduction at hugewiki.cu:2703-2730).
 */
int cumf_get_hermitian(const cumf_plan_t* plan, const int* colidx, const float* val,
                       const float* gather, float* tt, float* rhs, int f, float lambda, void* stream);
/* The same batch as PACKED upper triangles (row i keeps columns i .. f-1: f (f + 1) / 2 floats per system, the
 * layout of cumf_pack_upper), written straight from the accumulators: the payload of the multi-GPU partial-Gram
 * reduction (hugewiki.cu:2703-2717 copies and adds full f x f matrices) without materialising f x f first. */
int cumf_get_hermitian_packed(const cumf_plan_t* plan, const int* colidx, const float* val,
                              const float* gather, float* packed, float* rhs, int f, float lambda, void* stream);

/*
 * fp16 storage of the Gram batch for the solver (the reference's compile-time switch CUMF_TT_FP16 /
 * CUMF_XX_FP16: get_hermitian100_tt_fp16, als.cu:335-441; updateXWithCGKernel3, cg.cu:235-429;
 * host updateXWithCGHost_tt_fp16, cg.h:32).  The Gram is accumulated in fp32 exactly as
 * cumf_get_hermitian does and rounded to nearest-even fp16 on the store (__float2half_rn); the CG reads
 * halves and computes in fp32.  Halves the Gram bytes of the unfused path; opt-in because it changes
 * the numerics (the reference's author notes convergence problems with it).  tt_half / A_half:
 * DEVICE buffers of (rows x f x f) 2-byte elements.
 */
int cumf_get_hermitian_fp16(const cumf_plan_t* plan, const int* colidx, const float* val, const float* gather,
                            void* tt_half, float* rhs, int f, float lambda, void* stream);
int cumf_cg_solve_batched_fp16(const void* A_half, float* x, const float* b, long batch, int f, int cg_iters,
                               void* stream);
/* doALS-level switch (process-wide; environment CUMF_ALS_TT_FP16=1): with the CG solver, doALS runs
 * cumf_get_hermitian_fp16 + cumf_cg_solve_batched_fp16 per batch and prints "\tCG solver with fp16."
 * (als.cu:826,936). */
int cumf_set_tt_fp16(int enable);
int cumf_get_tt_fp16(void);

/* Batched CG on materialised systems; C alias of updateXWithCGHost (cg.h:30,
 * cg.cu:682-686): A batch x f x f, x batch x f (warm start in, solution out),
 * b batch x f, all DEVICE pointers.  Asynchronous on `stream`. */
int cumf_cg_solve_batched(const float* A, float* x, const float* b, long batch, int f, int cg_iters,
                          void* stream);

/* Batched unpivoted LU + 1 RHS (cublasSgetrfBatched + cublasSgetrsBatched with
 * PivotArray = NULL, als.cu:77,98 / 146,166).  A (batch x f x f) is NOT modified;
 * x (batch x f) receives the solution; b batch x f. */
int cumf_lu_solve_batched(const float* A, const float* b, float* x, long batch, int f, void* stream);

/*
 * Packed upper triangle of a batch of symmetric Grams: row i keeps columns i .. f-1, f (f + 1) / 2
 * floats per system.  It is what the multi-GPU Theta phase sums across GPUs: the reference copies and
 * adds the full f x f per GPU (hugewiki.cu:2703-2717), half of which is redundant.  DEVICE pointers.
 */
int cumf_pack_upper(const float* full, float* packed, long batch, int f, void* stream);
int cumf_unpack_upper(const float* packed, float* full, long batch, int f, void* stream);

/*
 * Sum of squared errors over `count` ratings (RMSE kernel + cublasSasum,
 * als.cu:191-219, 979-991, 1006-1019): sse_out is one DEVICE double.
 * surpass_nan reproduces `#define SURPASS_NAN` (als.cu:201-211).
 */
int cumf_sse(const float* val, const int* row, const int* col, const float* thetaT, const float* XT,
             long count, int f, int surpass_nan, double* sse_out, void* stream);

/*
 * Arithmetic of the Gram pass (the reference chooses its variants at compile time too:
 * `#define CUMF_USE_HALF` / CUMF_TT_FP16, als.cu:25-33).
 *   CUMF_GRAM_AUTO  (default) fp32 evaluated on the bf16 matrix pipe where a kernel for it exists
 *                   (LU, CG and the materialising pass, 16 <= f <= 207): every gathered fp32 value is
 *                   split EXACTLY into three bf16 terms and each product is formed from six bf16
 *                   products with fp32 accumulation; the dropped terms are < 2^-23 of a product.
 *                   fp32-class error against an fp64 Gram, not bit-identical to get_hermitian's
 *                   fmaf chain (als.h:39-143).  Everything else runs CUMF_GRAM_EXACT.
 *   CUMF_GRAM_EXACT v_mfma_f32_16x16x4_f32: bit-identical to the reference thread's fmaf chain.
 * Process-wide; also settable with the environment variable CUMF_ALS_GRAM=split|exact|fast read at the
 * first half-iteration.
 */
enum { CUMF_GRAM_AUTO = 0, CUMF_GRAM_EXACT = 1, CUMF_GRAM_FAST = 2 };
/*
 *   CUMF_GRAM_FAST  (opt-in; CUMF_ALS_GRAM=fast) the fused LU / CG passes of the wave kernels read a
 *                   PRE-SPLIT copy of the factor table, one (h, l) pair of f16 per value with
 *                   4096 x ~ h + l to 2^-22, and form each product from three f16 products (hh + hl + lh,
 *                   fp32 accumulation): 22 significand bits instead of 24, half the matrix-pipe work of
 *                   CUMF_GRAM_AUTO.  Needs the row count of the gather table (cumf_plan_set_gather_rows)
 *                   and factors and ratings below 15.99 in magnitude; violations are detected on the
 *                   device and reported by cumf_gram_fast_status (bit 0: a factor, bit 1: a rating).
 *                   Everything else (materialise, f <= 14) runs as CUMF_GRAM_AUTO.
 */
int cumf_plan_set_gather_rows(cumf_plan_t* plan, long gather_rows);
int cumf_gram_fast_status(int* flags);
/* 0 when a table of gather_rows x f floats can be gathered by the kernels that (solver, f,
 * materialize) select; an error (message on stderr) when that path addresses the table with 32-bit
 * byte offsets and the table is 4 GiB or larger.  doALS and the Python wrappers call it. */
int cumf_check_gather_table(long gather_rows, int f, int solver, int materialize);
int cumf_set_gram_mode(int mode);
int cumf_get_gram_mode(void);
/*
 * Round 6: where the gather table of a fused call lives in the caches (the Netflix Theta side gathers X: 7 MB; the
 * hugewiki X side gathers Theta: 16 MB) CUMF_GRAM_AUTO rewrites it per call as bf16 h | m | l planes -- the exact three-way
 * split the kernels otherwise perform on every gathered value, the same bits -- and the Gram stage reads its MFMA operands
 * from those planes (16-byte LDS-DMA + transposing LDS reads; als_wave.hip, kArithPrePk); the last, mostly empty feature
 * block is multiplied as ONE packed operand (three products instead of six).  Same arithmetic class as the in-kernel split
 * (every plane product of the last block column is kept, the rest is the same operation for operation).  Needs
 * cumf_plan_set_gather_rows; f = 64..79 or 96..111 with f % 16 in {0, 4}.
 *   cumf_set_presplit(mode)   CUMF_PRESPLIT_AUTO (default): tables whose planes take at most CUMF_ALS_PRESPLIT_MB (64) MB;
 *                             CUMF_PRESPLIT_OFF / CUMF_PRESPLIT_ON: never / whenever the shape allows;
 *                             CUMF_PRESPLIT_VERIFY: like ON, with the last block unpacked -- the same operands in the same
 *                             MFMA slots as the in-kernel split, BIT-IDENTICAL results (what the tests compare).  Also the
 *                             environment variable CUMF_ALS_PRESPLIT=0|1|2 read at the first half-iteration.
 *                             OFF and VERIFY select the six-product forms THROUGHOUT: AUTO / ON also let the in-kernel split
 *                             multiply a last feature block that holds nothing but the rating slot (f % 16 == 0, f <= 96) as
 *                             one packed operand (same error class as the packed pre-split form; no gather of that block).
 *   cumf_presplit_table       writes the planes of a rows x f table (device pointers; rows of cumf_presplit_pitch(f) bytes:
 *                             [h: 16 FB bf16][m][l][strip of f % 16 features: h, m, l, 8 zero bytes], FB = f / 16); what the
 *                             fused calls do internally -- exported so that callers and tests can check the planes.
 */
enum { CUMF_PRESPLIT_AUTO = -1, CUMF_PRESPLIT_OFF = 0, CUMF_PRESPLIT_ON = 1, CUMF_PRESPLIT_VERIFY = 2 };
int cumf_set_presplit(int mode);
int cumf_get_presplit(void);
long cumf_presplit_pitch(int f);
int cumf_presplit_table(const float* table, void* planes, long rows, int f, void* stream);

/*
 * Measurement hooks (no reference counterpart; the reference times phases with
 * gettimeofday under #ifdef DEBUG, als.cu:728-732,821,845).  When enabled, the two
 * kernels of a half-iteration (per-item Gram[+solve] kernel, chunked-row reduce kernel)
 * are bracketed by HIP events recorded on the launch stream; cumf_last_kernel_ms waits
 * for the last half-iteration and returns their durations in milliseconds.
 */
int cumf_set_kernel_timing(int enable);
int cumf_last_kernel_ms(float* item_kernel_ms, float* reduce_kernel_ms);
/* The same two durations SUMMED over every half-iteration launch sequence since the previous call (or since timing
 * was enabled), and their count: a half-iteration made of several launches (X_BATCH / THETA_BATCH plans, the
 * pipeline pieces of the multi-GPU gather scheme) is read with one call.  Waits for those launches; resets the sum. */
int cumf_kernel_ms_since_reset(float* item_kernel_ms, float* reduce_kernel_ms, int* launches);
/* Demangled name of the Gram(+solve) kernel the last half-iteration dispatched, as rocprofv3 prints it
 * (e.g. "cumf::als_wave_kernel<7, 1, 100, 0>"); buf receives a NUL-terminated string ("" before any launch). */
int cumf_last_kernel_name(char* buf, int cap);

/* Error state of the entry points that return a value instead of a code: cumf_doALS_ex / cumf_doALS / doALS
 * return NaN and set it when the opt-in gram mode "fast" meets data outside its range (the reference's own
 * convention -- print and exit, als.h:628-665 -- is kept for HIP failures only).  Reading clears it. */
enum { CUMF_ERR_FAST_RANGE = 10001 };
int cumf_last_error(void);
/* Frees the scratch the library keeps between calls on the CURRENT device (tile buffers of the f >= 144 LU,
 * pre-split tables of gram mode "fast"; one per device and stream, grow-only).  Safe against other host threads:
 * it waits for entry points that are mid-sequence on this device and leaves other devices' buffers alone.
 * doALS calls it before returning. */
int cumf_release_scratch(void);

/* Row pointers of a matrix with 2^31 or more ratings handed over as 4-byte values (doALS takes `const int*`; the
 * reference's own 3.1 G-rating run reads its row-pointer files as unsigned, hugewiki.cu:1973,1984): rowptr32 is read
 * as uint32 and 2^32 is added at every wrap (row pointers are non-decreasing), out64[0 .. rows] receives the result.
 * Host pointers.  Returns 0, or hipErrorInvalidValue when the result does not end at nnz.  cumf_doALS_ex does this
 * itself when nnz > 2^31 - 1. */
int cumf_widen_rowptr(const int* rowptr32, long rows, long nnz, long long* out64);

/* Library/version probe used by the loaders' "fail loudly" checks. */
/* Factor initialisation of the reference's hosts: a[k] = scale * ((float)rand() / (float)RAND_MAX)
 * for k in order, libc rand() (main.cpp:72-76 with scale 0.2 after srand(0); als_tf.cc:121-123
 * with scale 0.1, unseeded).  Host pointer.  seed < 0: continue the current rand() stream. */
void cumf_rand_init(float* a, long count, float scale, long seed);

int cumf_als_version(void);
const char* cumf_als_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* CUMF_ALS_CAPI_H_ */
