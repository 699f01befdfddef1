/*
 * als_oracle_impl.h -- body of the CPU oracle, included twice by als_oracle.c
 * (once with REAL=float / SUF=_f32, once with REAL=double / SUF=_f64).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (cumf_als_amd/,
 * include/, libALS.so) may include, link or call this.  See als_oracle.c header.
 *
 * Every function cites the reference lines (under /root/reference) whose
 * arithmetic it restates.  The restatement is sequential, single-accumulator
 * arithmetic in CSR order: that is what one reference thread does for one
 * output element (one register `tempNN` per Gram entry, accumulated rating by
 * rating with an FMA -- als.h:39-143, als.cu:524-534).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* fused multiply-add in REAL: nvcc contracts `t += a*b` to one FMA by default,
 * so a single reference thread's chain is an fma chain. */
static inline REAL FN(fma_)(REAL a, REAL b, REAL c) {
#ifdef REAL_IS_DOUBLE
  return fma(a, b, c);
#else
  return fmaf(a, b, c);
#endif
}

/*
 * Gram + RHS for rows [row_begin, row_end) of a CSR matrix.
 *   A_u = sum_{v in Omega_u} theta_v theta_v^T + lambda * n_u * I     (als.cu:443-569, 575-659)
 *   b_u = sum_{v in Omega_u} r_uv theta_v                              (als.cu:750-757, 867-874)
 * tt  : (row_end-row_begin) * f * f, row-major f x f, both triangles  (als.cu:559-566)
 * b   : (row_end-row_begin) * f,  f contiguous per row                (ythetaT[f*u+k], als.cu:756)
 * The lambda term is `(end - start) * lambda` evaluated in fp32 (int -> float
 * product) and added to the diagonal after the accumulation (als.cu:545-557, 653-656).
 * An empty row yields an all-zero A and b (als.cu:455: iterations == 1, no loads).
 * Either output may be NULL.
 */
static void FN(gram_rhs_row)(const int *rowptr, const int *colidx, const float *val,
                             const float *factors, int f, float lambda, long u,
                             REAL *A, REAL *bu) {
  long start = rowptr[u], end = rowptr[u + 1];
  if (A) {
    for (int i = 0; i < f * f; i++) A[i] = 0;
    /* rating-outer loop: every A[i][j] is still ONE sequential FMA chain in CSR
     * order (the reference thread's `tempNN += th[x]*th[y]`, als.h:39-143). */
    for (long k = start; k < end; k++) {
      const float *th = factors + (size_t)colidx[k] * f;
      for (int i = 0; i < f; i++) {
        REAL ti = (REAL)th[i];
        REAL *Ai = A + (size_t)i * f;
        for (int j = i; j < f; j++) Ai[j] = FN(fma_)(ti, (REAL)th[j], Ai[j]);
      }
    }
    for (int i = 0; i < f; i++)
      for (int j = i + 1; j < f; j++) A[j * f + i] = A[i * f + j];
    /* als.cu:547: float temp = (end - start) * lambda;  (int * float in fp32) */
#ifdef REAL_IS_DOUBLE
    REAL reg = (double)(end - start) * (double)lambda;
#else
    REAL reg = (float)(end - start) * lambda;
#endif
    for (int i = 0; i < f; i++) A[i * f + i] += reg;
  }
  if (bu) {
    for (int i = 0; i < f; i++) bu[i] = 0;
    for (long k = start; k < end; k++) {
      const float *th = factors + (size_t)colidx[k] * f;
      REAL r = (REAL)val[k];
      for (int i = 0; i < f; i++) bu[i] = FN(fma_)(r, (REAL)th[i], bu[i]);
    }
  }
}

void FN(oracle_gram_rhs)(const int *rowptr, const int *colidx, const float *val,
                         const float *factors, int f, float lambda,
                         long row_begin, long row_end, REAL *tt, REAL *b) {
  long u;
#pragma omp parallel for schedule(dynamic, 16)
  for (u = row_begin; u < row_end; u++) {
    long lu = u - row_begin;
    FN(gram_rhs_row)(rowptr, colidx, val, factors, f, lambda, u,
                     tt ? tt + (size_t)lu * f * f : NULL, b ? b + (size_t)lu * f : NULL);
  }
}

/*
 * Batched conjugate gradient, one system per "block" (cg.cu:36-231).
 *   x <- warm start (cg.cu:48); r = b - A x (cg.cu:52-56); p = r; rsold = r.r (cg.cu:60-67)
 *   loop iter < cgIter (cg.cu:85): ap = A p (cg.cu:88-92); alpha = rsold / p.ap (cg.cu:119-128);
 *   x += alpha p; r -= alpha ap (cg.cu:142-146); rsnew = r.r (cg.cu:174-177);
 *   if (rsnew < CG_ERROR) break (cg.cu:31,195 -- float promoted to double vs the
 *   double literal 1e-4); beta = rsnew / rsold; rsold = rsnew (cg.cu:197-200);
 *   p = r + beta p (cg.cu:203-204).  x written back (cg.cu:230).
 * The mat-vec reads A[f*i + tid], i.e. column tid == row tid for symmetric A (cg.cu:55,91).
 * Dot products: the reference sums with warp shuffles + smem atomicAdd in
 * non-deterministic order (device_utilities.h:36-48); the oracle uses index order.
 * cgIter is a float in the reference signature (cg.h:30).
 */
static void FN(cg_one)(const REAL *As, REAL *xs, const REAL *bs, int f, float cgIter) {
  REAL r[ORACLE_MAX_F], p[ORACLE_MAX_F], ap[ORACLE_MAX_F], xx[ORACLE_MAX_F];
  for (int t = 0; t < f; t++) xx[t] = xs[t];
  for (int t = 0; t < f; t++) {
    REAL temp = 0;
    for (int i = 0; i < f; i++) temp = FN(fma_)(As[f * i + t], xx[i], temp);
    r[t] = bs[t] - temp;
    p[t] = r[t];
  }
  REAL rsold = 0;
  for (int t = 0; t < f; t++) rsold = FN(fma_)(r[t], r[t], rsold);
  for (int iter = 0; iter < cgIter; iter++) {
    for (int t = 0; t < f; t++) {
      REAL temp = 0;
      for (int i = 0; i < f; i++) temp = FN(fma_)(As[f * i + t], p[i], temp);
      ap[t] = temp;
    }
    REAL pap = 0;
    for (int t = 0; t < f; t++) pap = FN(fma_)(p[t], ap[t], pap);
    REAL alpha = rsold / pap;
    for (int t = 0; t < f; t++) {
      xx[t] = FN(fma_)(alpha, p[t], xx[t]);
      r[t] = FN(fma_)(-alpha, ap[t], r[t]);
    }
    REAL rsnew = 0;
    for (int t = 0; t < f; t++) rsnew = FN(fma_)(r[t], r[t], rsnew);
    if ((double)rsnew < 1e-4) break;
    REAL beta = rsnew / rsold;
    rsold = rsnew;
    for (int t = 0; t < f; t++) p[t] = FN(fma_)(beta, p[t], r[t]);
  }
  for (int t = 0; t < f; t++) xs[t] = xx[t];
}

void FN(oracle_cg)(const REAL *A, REAL *x, const REAL *b, long batch, int f, float cgIter) {
  long s;
#pragma omp parallel for schedule(dynamic, 16)
  for (s = 0; s < batch; s++)
    FN(cg_one)(A + (size_t)s * f * f, x + (size_t)s * f, b + (size_t)s * f, f, cgIter);
}

/*
 * Batched UNPIVOTED LU + one right-hand side, in place (als.cu:58-122, 124-189):
 *   cublasSgetrfBatched(handle, f, A[], lda=f, PivotArray=NULL, info, batch)   (als.cu:77,146)
 *   cublasSgetrsBatched(handle, OP_N, f, nrhs=1, A[], f, NULL, b[], f, ...)    (als.cu:98,166)
 * cuBLAS is closed source and unpinned: the mathematical definition is
 * the spec -- Doolittle LU without row exchanges (column-major view of a symmetric
 * matrix == row-major view), then L y = b, U x = y.  b is overwritten with x.
 * A is overwritten with the factors (unit-lower L below the diagonal, U on/above).
 */
static void FN(lu_one)(REAL *As, REAL *bs, int f) {
  for (int k = 0; k < f; k++) {
    REAL piv = As[k * f + k];
    for (int i = k + 1; i < f; i++) {
      REAL l = As[i * f + k] / piv;
      As[i * f + k] = l;
      for (int j = k + 1; j < f; j++)
        As[i * f + j] = FN(fma_)(-l, As[k * f + j], As[i * f + j]);
    }
  }
  for (int i = 0; i < f; i++) {
    REAL acc = bs[i];
    for (int j = 0; j < i; j++) acc = FN(fma_)(-As[i * f + j], bs[j], acc);
    bs[i] = acc;
  }
  /* U x = y, column-oriented like the reference BLAS strsv that getrs performs
   * (upper, no-transpose): x_j is final, then every y_i (i < j) loses U_ij x_j -- so
   * row i accumulates its terms in DESCENDING j. */
  for (int i = f - 1; i >= 0; i--) {
    REAL acc = bs[i];
    for (int j = f - 1; j > i; j--) acc = FN(fma_)(-As[i * f + j], bs[j], acc);
    bs[i] = acc / As[i * f + i];
  }
}

void FN(oracle_lu)(REAL *A, REAL *b, long batch, int f) {
  long s;
#pragma omp parallel for schedule(dynamic, 16)
  for (s = 0; s < batch; s++) FN(lu_one)(A + (size_t)s * f * f, b + (size_t)s * f, f);
}

/*
 * RMSE kernel + Sasum (als.cu:191-219, 979-991, 1006-1019).
 *   e_i = val[i] - sum_k thetaT[f*col+k] * XT[f*row+k]  (sequential FMA, als.cu:199-213)
 *   error[i % error_size] += e_i^2  (atomicAdd, als.cu:216), error_size = 1000 (als.cu:966)
 *   returns sum(error) ; caller takes sqrt(sum / nnz)  (als.cu:988-991)
 * `count` is the number of ratings the grid covers: nnz for train ((nnz-1)/256+1 blocks,
 * als.cu:979) but ((nnz_test-1)/256)*256 for test (missing +1, als.cu:1006).
 * surpass_nan: SURPASS_NAN variant (als.cu:201-211): stop the dot product at the
 * first NaN factor entry and keep the partial e.
 */
double FN(oracle_sse)(const float *val, const int *row, const int *col,
                      const REAL *thetaT, const REAL *XT, long count, int f,
                      int surpass_nan) {
  enum { ERROR_SIZE = 1000 };
  REAL bins[ERROR_SIZE];
  for (int i = 0; i < ERROR_SIZE; i++) bins[i] = 0;
  /* Bin b receives the ratings i = b, b + 1000, ... in increasing i (the order a sequential walk over i gives it):
   * the bins are independent, so walking them in parallel changes no bit of any of them. */
  int b;
#pragma omp parallel for schedule(dynamic, 8)
  for (b = 0; b < ERROR_SIZE; b++) {
    REAL acc = 0;
    for (long i = b; i < count; i += ERROR_SIZE) {
      REAL e = (REAL)val[i];
      const REAL *th = thetaT + (size_t)col[i] * f;
      const REAL *xr = XT + (size_t)row[i] * f;
      for (int k = 0; k < f; k++) {
        REAL a = th[k], bb = xr[k];
        if (surpass_nan && (a != a || bb != bb)) break;
        e = FN(fma_)(-a, bb, e);
      }
      acc += e * e;
    }
    bins[b] = acc;
  }
  REAL sum = 0;
  for (int i = 0; i < ERROR_SIZE; i++) sum += (bins[i] < 0 ? -bins[i] : bins[i]); /* Sasum */
  return (double)sum;
}

/*
 * One half-iteration: update the factors of rows [0, rows) from `gather`
 * (als.cu:727-855 for X with CSR, als.cu:857-964 for Theta with CSC-as-CSR).
 * Rows are processed in `nbatch` batches exactly as als.cu:768-777 / 881-890
 * (batch_size = rows / nbatch, last batch takes the remainder); results do not
 * depend on nbatch.  solver: 0 = CG (USE_CG, als.cu:28,831,941), 1 = LU (als.cu:839,948).
 */
void FN(oracle_half_iteration)(const int *rowptr, const int *colidx, const float *val,
                               const float *gather, float *update, long rows, int f,
                               float lambda, int nbatch, int solver, int cg_iters) {
  for (int batch_id = 0; batch_id < nbatch; batch_id++) {
    long batch_size = (batch_id != nbatch - 1) ? rows / nbatch : rows - batch_id * (rows / nbatch);
    long batch_offset = (long)batch_id * (rows / nbatch);
    long u;
    /* Rows are independent (one CUDA block per row, als.cu:449): the Gram batch
     * buffer of the reference (als.cu:782,897) is not materialised here; each
     * row's f x f system lives in per-thread scratch. */
#pragma omp parallel
    {
      REAL *A = (REAL *)malloc((size_t)f * f * sizeof(REAL));
      REAL bu[ORACLE_MAX_F], xu[ORACLE_MAX_F];
#pragma omp for schedule(dynamic, 8)
      for (u = batch_offset; u < batch_offset + batch_size; u++) {
        FN(gram_rhs_row)(rowptr, colidx, val, gather, f, lambda, u, A, bu);
        float *out = update + (size_t)u * f;
        if (solver == 0) {
          for (int i = 0; i < f; i++) xu[i] = (REAL)out[i];
          FN(cg_one)(A, xu, bu, f, (float)cg_iters);
          for (int i = 0; i < f; i++) out[i] = (float)xu[i];
        } else {
          FN(lu_one)(A, bu, f);
          for (int i = 0; i < f; i++) out[i] = (float)bu[i];
        }
      }
      free(A);
    }
  }
}

/*
 * doALS restated (als.cu:662-1035): ITERS x (update X, update Theta, train RMSE,
 * test RMSE).  Factors are kept in fp32 between half-iterations (they are the
 * reference's device arrays thetaT / XT); with REAL=double only the per-row
 * arithmetic is widened, which bounds the fp32 path's rounding error.
 * rmse_log (may be NULL) receives 2*ITERS doubles: train, test per iteration.
 * test_grid_compat != 0 reproduces the truncated test grid of als.cu:1006.
 * Returns the final test RMSE (als.cu:1018,1034).
 */
float FN(oracle_doALS)(const int *csrRow, const int *csrCol, const float *csrVal,
                       const int *cscRow, const int *cscCol, const float *cscVal,
                       const int *cooRow, float *thetaT, float *XT,
                       const int *cooRowTest, const int *cooColTest, const float *cooValTest,
                       int m, int n, int f, long nnz, long nnz_test, float lambda,
                       int ITERS, int X_BATCH, int THETA_BATCH,
                       int solver, int cg_iters, int test_grid_compat, int surpass_nan,
                       double *rmse_log) {
  float final_rmse = 0;
  REAL *th = (REAL *)malloc((size_t)n * f * sizeof(REAL));
  REAL *xr = (REAL *)malloc((size_t)m * f * sizeof(REAL));
  for (int iter = 0; iter < ITERS; iter++) {
    FN(oracle_half_iteration)(csrRow, csrCol, csrVal, thetaT, XT, m, f, lambda, X_BATCH, solver, cg_iters);
    /* als.cu:867-869: CSC arrays passed as the CSR of R^T (colptr, rowidx) */
    FN(oracle_half_iteration)(cscCol, cscRow, cscVal, XT, thetaT, n, f, lambda, THETA_BATCH, solver, cg_iters);
    for (size_t i = 0; i < (size_t)n * f; i++) th[i] = (REAL)thetaT[i];
    for (size_t i = 0; i < (size_t)m * f; i++) xr[i] = (REAL)XT[i];
    double sse_train = FN(oracle_sse)(csrVal, cooRow, csrCol, th, xr, nnz, f, surpass_nan);
    long count_test = test_grid_compat ? ((nnz_test - 1) / 256) * 256 : nnz_test;
    if (count_test < 0) count_test = 0;
    double sse_test = FN(oracle_sse)(cooValTest, cooRowTest, cooColTest, th, xr, count_test, f, surpass_nan);
    double rmse_train = sqrt(sse_train / (double)nnz);
    final_rmse = (float)sqrt((float)sse_test / (float)nnz_test);
    if (rmse_log) {
      rmse_log[2 * iter] = rmse_train;
      rmse_log[2 * iter + 1] = (double)final_rmse;
    }
  }
  free(th);
  free(xr);
  return final_rmse;
}

#undef FN
#undef CAT
#undef CAT_
