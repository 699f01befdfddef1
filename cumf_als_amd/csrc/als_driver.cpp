// als_driver.cpp -- `doALS`, the drop-in boundary (als.h:676-681, als.cu:662-1035).
//
// Same 22 host-pointer arguments, same in/out factors, same return value (final test
// RMSE) and same stdout lines as the reference.  What changes underneath:
//   * everything is uploaded ONCE and stays resident in HBM (the reference re-mallocs
//     and re-uploads CSR, COO and the test set every iteration: als.cu:734-739,
//     972-977, 999-1004);
//   * a half-iteration is one fused pass (RHS + Gram + solve, cumf_als_update_fused);
//     the reference's data flow (Gram batch in device memory, separate solver) is kept
//     as the "unfused" path and is what f > 128 uses;
//   * X_BATCH / THETA_BATCH keep their meaning (als.cu:768-777, 881-890): rows are
//     processed in that many slices; results do not depend on them.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "als.h"
#include "als_internal.h"
#include "cumf_als_capi.h"

namespace {

#define DRV_CHECK(call)                                                                                         \
  do {                                                                                                          \
    hipError_t err__ = (hipError_t)(call);                                                                      \
    if (err__ != hipSuccess) {                                                                                  \
      /* error convention of als.h:628-639: message to stderr, then exit */                                    \
      fprintf(stderr, "HIP Error:\nFile = %s\nLine = %d\nReason = %s\n", __FILE__, __LINE__,                    \
              hipGetErrorString(err__));                                                                        \
      exit(EXIT_FAILURE);                                                                                       \
    }                                                                                                           \
  } while (0)

template <typename T>
T* to_device(const T* host, size_t count) {
  T* d = nullptr;
  if (count == 0) count = 1;
  DRV_CHECK(hipMalloc(reinterpret_cast<void**>(&d), count * sizeof(T)));
  if (host) DRV_CHECK(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

struct Side {
  // one side of the factorisation: rows x cols CSR (CSC of R passed as CSR of R^T, als.cu:867-869)
  const int* rowptr_host;
  const long long* rowptr64;  // the same widened to 64 bits when nnz > 2^31 - 1 (else nullptr)
  const int* d_colidx;
  const float* d_val;
  long rows;
  int nbatch;
  std::vector<cumf_plan_t*> plans;
  std::vector<long> offset, size;
};

void make_side(Side& s, int f, long gather_rows) {
  for (int b = 0; b < s.nbatch; ++b) {
    // als.cu:768-777
    const long bs = (b != s.nbatch - 1) ? s.rows / s.nbatch : s.rows - (long)b * (s.rows / s.nbatch);
    const long off = (long)b * (s.rows / s.nbatch);
    cumf_plan_t* p = nullptr;
    DRV_CHECK(s.rowptr64 ? cumf_plan_create(&p, s.rowptr64, 1, s.rows, off, off + bs, f, 0)
                         : cumf_plan_create(&p, s.rowptr_host, 0, s.rows, off, off + bs, f, 0));
    DRV_CHECK(cumf_plan_set_gather_rows(p, gather_rows));  // gram mode "fast" pre-splits the gather table
    s.plans.push_back(p);
    s.offset.push_back(off);
    s.size.push_back(bs);
  }
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int g_tt_fp16 = -1;  // fp16 Gram storage for the CG solver: `#define CUMF_TT_FP16` / CUMF_XX_FP16 (als.cu:25-33)

}  // namespace

extern "C" int cumf_widen_rowptr(const int* rowptr32, long rows, long nnz, long long* out64) {
  if (!rowptr32 || !out64 || rows < 0) return (int)hipErrorInvalidValue;
  long long hi = 0;
  unsigned prev = static_cast<unsigned>(rowptr32[0]);
  out64[0] = prev;
  for (long i = 1; i <= rows; ++i) {
    const unsigned cur = static_cast<unsigned>(rowptr32[i]);
    if (cur < prev) hi += 1ll << 32;  // non-decreasing row pointers: a smaller value is a wrap
    out64[i] = hi + cur;
    prev = cur;
  }
  if (out64[rows] != (long long)nnz) {
    fprintf(stderr, "cumf_widen_rowptr: the row pointer ends at %lld, not at nnz = %ld\n", out64[rows], nnz);
    return (int)hipErrorInvalidValue;
  }
  return 0;
}

extern "C" int cumf_set_tt_fp16(int enable) {
  g_tt_fp16 = enable != 0;
  return 0;
}
extern "C" int cumf_get_tt_fp16(void) {
  if (g_tt_fp16 < 0) g_tt_fp16 = env_int("CUMF_ALS_TT_FP16", 0) != 0;
  return g_tt_fp16;
}

extern "C" float cumf_doALS_ex(const int* csrRowIndexHostPtr, const int* csrColIndexHostPtr,
                               const float* csrValHostPtr, const int* cscRowIndexHostPtr,
                               const int* cscColIndexHostPtr, const float* cscValHostPtr,
                               const int* cooRowIndexHostPtr, float* thetaTHost, float* XTHost,
                               const int* cooRowIndexTestHostPtr, const int* cooColIndexTestHostPtr,
                               const float* cooValHostTestPtr, int m, int n, int f, long nnz, long nnz_test,
                               float lambda, int ITERS, int X_BATCH, int THETA_BATCH, int DEVICEID, int solver,
                               int cg_iters, int fused, int exact_test_grid, int surpass_nan, int quiet,
                               float* rmse_log) {
  // CUMF_ALS_TIMING=1: wall-clock phases of the call on stderr (the reference times its phases under #ifdef DEBUG, als.cu:728-732)
  const bool phase_timing = env_int("CUMF_ALS_TIMING", 0) != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!phase_timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "doALS phase %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  DRV_CHECK(hipSetDevice(DEVICEID));
  phase("hipSetDevice");
  if (!quiet) printf("*******parameters: m: %d, n:  %d, f: %d, nnz: %ld \n", m, n, f, nnz);
  if (X_BATCH < 1) X_BATCH = 1;
  if (THETA_BATCH < 1) THETA_BATCH = 1;
  if (!cumf_fused_available(f, solver)) fused = 0;
  // fp16 storage of the Gram batch (als.cu:779-783, 893-895): the reference's data flow by construction
  // (Gram batch in device memory + updateXWithCGHost_tt_fp16); CG only, like the reference
  const int tt_fp16 = cumf_get_tt_fp16() && solver == CUMF_SOLVER_CG;
  if (tt_fp16) fused = 0;

  // both factor tables are gathered from (Theta by the X update, X by the Theta update)
  DRV_CHECK(cumf_check_gather_table(n, f, solver, !fused));
  DRV_CHECK(cumf_check_gather_table(m, f, solver, !fused));
  if (!quiet) printf("*******start allocating memory on GPU...\n");
  // The uploads (2 GB at the Netflix shape) run on a stream of their own while the host builds the plans below
  // (sorting ~500 k work items per side); pinned callers (main.cpp:50-69) get true overlap, pageable ones (the TF op) the
  // staged copy they would get anyway.  Non-blocking: the plans' own small synchronous copies do not wait for it.
  hipStream_t up = nullptr;
  DRV_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
  auto to_device_async = [&](auto* host, size_t count) {
    using T = std::remove_cv_t<std::remove_pointer_t<decltype(host)>>;
    T* d = nullptr;
    DRV_CHECK(hipMalloc(reinterpret_cast<void**>(&d), (count ? count : 1) * sizeof(T)));
    if (host && count) DRV_CHECK(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, up));
    return d;
  };
  int* csrColIndex = to_device_async(csrColIndexHostPtr, (size_t)nnz);
  float* csrVal = to_device_async(csrValHostPtr, (size_t)nnz);
  int* cscRowIndex = to_device_async(cscRowIndexHostPtr, (size_t)nnz);
  float* cscVal = to_device_async(cscValHostPtr, (size_t)nnz);
  int* cooRowIndex = nullptr;  // only the RMSE kernel reads it (als.cu:972-977): uploaded below if that kernel runs
  int* cooRowIndex_test = to_device_async(cooRowIndexTestHostPtr, (size_t)nnz_test);
  int* cooColIndex_test = to_device_async(cooColIndexTestHostPtr, (size_t)nnz_test);
  float* cooVal_test = to_device_async(cooValHostTestPtr, (size_t)nnz_test);
  if (!quiet) printf("*******start copying memory to GPU...\n");
  float* thetaT = to_device_async(thetaTHost, (size_t)n * f);
  float* XT = to_device_async(XTHost, (size_t)m * f);
  double* d_sse = to_device<double>(nullptr, 2);
  double* d_bins = to_device<double>(nullptr, CUMF_SSE_BINS);  // fused train SSE of the Theta update

  // 2^31 or more ratings: the 4-byte row pointers have wrapped (hugewiki.cu:1973 reads them as unsigned)
  std::vector<long long> csr64, csc64;
  if (nnz > 0x7fffffffL) {
    csr64.resize((size_t)m + 1);
    csc64.resize((size_t)n + 1);
    DRV_CHECK(cumf_widen_rowptr(csrRowIndexHostPtr, m, nnz, csr64.data()));
    DRV_CHECK(cumf_widen_rowptr(cscColIndexHostPtr, n, nnz, csc64.data()));
  }
  phase("hipMalloc + async uploads issued");
  Side sx{csrRowIndexHostPtr, csr64.empty() ? nullptr : csr64.data(), csrColIndex, csrVal, m, X_BATCH, {}, {}, {}};
  Side st{cscColIndexHostPtr, csc64.empty() ? nullptr : csc64.data(), cscRowIndex, cscVal, n, THETA_BATCH, {}, {}, {}};
  make_side(sx, f, n);  // X rows gather from thetaT (n rows)
  make_side(st, f, m);  // Theta rows gather from XT (m rows)
  phase("plans (host) under the uploads");

  // unfused path: Gram batch `tt` (als.cu:782,897) + RHS (ythetaT / yTXT, als.cu:746,864)
  float *tt = nullptr, *rhs = nullptr;
  if (!fused) {
    long maxb = 0;
    for (long s : sx.size) maxb = s > maxb ? s : maxb;
    for (long s : st.size) maxb = s > maxb ? s : maxb;
    tt = to_device<float>(nullptr, tt_fp16 ? ((size_t)maxb * f * f + 1) / 2 : (size_t)maxb * f * f);
    rhs = to_device<float>(nullptr, (size_t)maxb * f);
  }

  // Train RMSE (als.cu:979-991) without its own pass: the Theta update delivers sum (r - x_u . theta_v)^2 of its
  // columns from the systems it has just solved (cumf_als_update_fused_sse).  Off -- the RMSE kernel runs as in the
  // reference -- with CUMF_ALS_RMSE=kernel, with SURPASS_NAN semantics (als.cu:201-211 skip NaN factor entries inside
  // a dot product: not expressible on the system level), and when a Theta plan has rows the wave kernel does not solve.
  bool fuse_rmse = fused && !surpass_nan;
  {
    const char* e = getenv("CUMF_ALS_RMSE");
    if (e && strcmp(e, "kernel") == 0) fuse_rmse = false;
    for (cumf_plan_t* p : st.plans) fuse_rmse = fuse_rmse && cumf_fused_sse_available(p, solver);
  }
  if (!fuse_rmse) cooRowIndex = to_device_async(cooRowIndexHostPtr, (size_t)nnz);
  // The fused value is S - 2 t.b + t^T G t evaluated in fp32 per column: absolute error <~ 1e-7 of S = sum r^2 (measured
  // 2e-9 .. 6e-8).  Harmless while the fit leaves a few per cent of S (Netflix: SSE / S ~ 1/15); when the fit is
  // near-perfect the value would be mostly cancellation noise, so S is taken once (the SSE kernel with f = 0
  // sums val^2; the index arrays are only dereferenced) and an iteration whose fused SSE falls below 1e-3 S -- where the
  // relative error of the RMSE could pass 5e-5 -- is re-evaluated by the RMSE kernel.
  double sum_r2 = 0.0;
  const double kFusedSseFloor = 1e-3;
  DRV_CHECK(hipStreamSynchronize(up));
  DRV_CHECK(hipStreamDestroy(up));
  phase("uploads complete");
  if (fuse_rmse) {
    DRV_CHECK(cumf_sse(csrVal, csrColIndex, csrColIndex, thetaT, XT, nnz, 0, 0, d_sse, nullptr));
    DRV_CHECK(hipMemcpy(&sum_r2, d_sse, sizeof(double), hipMemcpyDeviceToHost));
  }

  auto half_iteration = [&](Side& s, const float* gather, float* update, double* sse_bins) {
    for (int b = 0; b < s.nbatch; ++b) {
      if (fused) {
        if (solver == CUMF_SOLVER_CG && !quiet) printf("\tCG solver with fp32.\n");
        if (sse_bins)
          DRV_CHECK(cumf_als_update_fused_sse(s.plans[b], s.d_colidx, s.d_val, gather, update, f, lambda, solver,
                                              cg_iters, sse_bins, nullptr));
        else
          DRV_CHECK(cumf_als_update_fused(s.plans[b], s.d_colidx, s.d_val, gather, update, f, lambda, solver,
                                          cg_iters, nullptr));
      } else {
        float* xb = update + (size_t)s.offset[b] * f;
        if (tt_fp16) {
          DRV_CHECK(cumf_get_hermitian_fp16(s.plans[b], s.d_colidx, s.d_val, gather, tt, rhs, f, lambda, nullptr));
          if (!quiet) printf("\tCG solver with fp16.\n");  // als.cu:826,936
          DRV_CHECK(cumf_cg_solve_batched_fp16(tt, xb, rhs, s.size[b], f, cg_iters, nullptr));
          continue;
        }
        DRV_CHECK(cumf_get_hermitian(s.plans[b], s.d_colidx, s.d_val, gather, tt, rhs, f, lambda, nullptr));
        if (solver == CUMF_SOLVER_CG) {
          if (!quiet) printf("\tCG solver with fp32.\n");
          DRV_CHECK(cumf_cg_solve_batched(tt, xb, rhs, s.size[b], f, cg_iters, nullptr));
        } else {
          DRV_CHECK(cumf_lu_solve_batched(tt, rhs, xb, s.size[b], f, nullptr));
        }
      }
    }
  };

  float final_rmse = 0;
  bool range_error = false;
  if (!quiet) printf("*******start iterations...\n");
  for (int iter = 0; iter < ITERS; iter++) {
    half_iteration(sx, thetaT, XT, nullptr);  // update X      (als.cu:727-855)
    if (fuse_rmse) DRV_CHECK(hipMemsetAsync(d_bins, 0, CUMF_SSE_BINS * sizeof(double), nullptr));
    half_iteration(st, XT, thetaT, fuse_rmse ? d_bins : nullptr);  // update Theta  (als.cu:857-964)
    if (cumf_get_gram_mode() == CUMF_GRAM_FAST) {
      int flags = 0;
      DRV_CHECK(cumf_gram_fast_status(&flags));
      if (flags) {
        // not a device failure: the opt-in arithmetic cannot represent this data.  Report and return NaN
        // (cumf_last_error() == CUMF_ERR_FAST_RANGE); the factors are not copied back.
        fprintf(stderr, "doALS: gram mode \"fast\": %s beyond the f16 range (|value| >= 15.99) in iteration %d; "
                        "use CUMF_ALS_GRAM=split\n", (flags & 1) ? "a factor" : "a rating", iter);
        cumf::set_last_error(CUMF_ERR_FAST_RANGE);
        range_error = true;
        break;
      }
    }

    // RMSE (als.cu:966-1020).  The test grid of the reference is (nnz_test-1)/256 blocks
    // -- one short of covering the set (als.cu:1006) -- yet divides by nnz_test.
    long count_test = exact_test_grid ? nnz_test : ((nnz_test - 1) / 256) * 256;
    if (count_test < 0) count_test = 0;
    if (!fuse_rmse)
      DRV_CHECK(cumf_sse(csrVal, cooRowIndex, csrColIndex, thetaT, XT, nnz, f, surpass_nan, d_sse, nullptr));
    DRV_CHECK(cumf_sse(cooVal_test, cooRowIndex_test, cooColIndex_test, thetaT, XT, count_test, f, surpass_nan,
                       d_sse + 1, nullptr));
    double sse[2];
    DRV_CHECK(hipMemcpy(sse, d_sse, sizeof(sse), hipMemcpyDeviceToHost));
    if (fuse_rmse) {
      double bins[CUMF_SSE_BINS];
      DRV_CHECK(hipMemcpy(bins, d_bins, sizeof(bins), hipMemcpyDeviceToHost));
      sse[0] = 0.0;
      for (int i = 0; i < CUMF_SSE_BINS; ++i) sse[0] += bins[i];  // fixed order (als.cu:988: cublasSasum over the bins)
      if (!(sse[0] >= kFusedSseFloor * sum_r2)) {  // near-perfect fit (or NaN): the direct evaluation of this iteration
        if (!cooRowIndex) {
          DRV_CHECK(hipMalloc(reinterpret_cast<void**>(&cooRowIndex), (size_t)nnz * sizeof(int)));
          DRV_CHECK(hipMemcpy(cooRowIndex, cooRowIndexHostPtr, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
        }
        DRV_CHECK(cumf_sse(csrVal, cooRowIndex, csrColIndex, thetaT, XT, nnz, f, surpass_nan, d_sse, nullptr));
        DRV_CHECK(hipMemcpy(sse, d_sse, sizeof(double), hipMemcpyDeviceToHost));
      }
    }
    const float rmse_train = (float)sqrt(sse[0] / (double)nnz);
    final_rmse = (float)sqrt(sse[1] / (double)nnz_test);
    if (!quiet) {
      printf("--------- Train RMSE in iter %d: %f\n", iter, rmse_train);
      printf("--------- Test RMSE in iter %d: %f\n", iter, final_rmse);
    }
    if (rmse_log) {
      rmse_log[2 * iter] = rmse_train;
      rmse_log[2 * iter + 1] = final_rmse;
    }
  }
  DRV_CHECK(hipDeviceSynchronize());
  phase("iterations");
  if (!range_error) {
    // copy feature vectors back to host (als.cu:1024-1025)
    DRV_CHECK(hipMemcpy(thetaTHost, thetaT, (size_t)n * f * sizeof(float), hipMemcpyDeviceToHost));
    DRV_CHECK(hipMemcpy(XTHost, XT, (size_t)m * f * sizeof(float), hipMemcpyDeviceToHost));
  }

  for (cumf_plan_t* p : sx.plans) cumf_plan_destroy(p);
  for (cumf_plan_t* p : st.plans) cumf_plan_destroy(p);
  void* bufs[] = {csrColIndex, csrVal, cscRowIndex, cscVal, cooRowIndex, cooRowIndex_test, cooColIndex_test,
                  cooVal_test, thetaT,  XT,      d_sse,  tt,          rhs, d_bins};
  for (void* q : bufs)
    if (q) DRV_CHECK(hipFree(q));
  DRV_CHECK(cumf_release_scratch());  // pooled tile buffers / pre-split tables of the plans above
  phase("factors back + teardown");
  // the device is NOT reset here (als.cu:1031-1033: "WARN: do not call cudaDeviceReset inside ALS()")
  return range_error ? nanf("") : final_rmse;
}

extern "C" float cumf_doALS(const int* csrRowIndexHostPtr, const int* csrColIndexHostPtr, const float* csrValHostPtr,
                            const int* cscRowIndexHostPtr, const int* cscColIndexHostPtr, const float* cscValHostPtr,
                            const int* cooRowIndexHostPtr, float* thetaTHost, float* XTHost,
                            const int* cooRowIndexTestHostPtr, const int* cooColIndexTestHostPtr,
                            const float* cooValHostTestPtr, const int m, const int n, const int f, const long nnz,
                            const long nnz_test, const float lambda, const int ITERS, const int X_BATCH,
                            const int THETA_BATCH, const int DEVICEID) {
  // run-time versions of the compile-time switches of als.cu:25-33
  const char* s = getenv("CUMF_ALS_SOLVER");
  const int solver = (s && (strcmp(s, "lu") == 0 || strcmp(s, "LU") == 0)) ? CUMF_SOLVER_LU : CUMF_SOLVER_CG;
  const char* pth = getenv("CUMF_ALS_PATH");
  const int fused = !(pth && strcmp(pth, "unfused") == 0);
  return cumf_doALS_ex(csrRowIndexHostPtr, csrColIndexHostPtr, csrValHostPtr, cscRowIndexHostPtr,
                       cscColIndexHostPtr, cscValHostPtr, cooRowIndexHostPtr, thetaTHost, XTHost,
                       cooRowIndexTestHostPtr, cooColIndexTestHostPtr, cooValHostTestPtr, m, n, f, nnz, nnz_test,
                       lambda, ITERS, X_BATCH, THETA_BATCH, DEVICEID, solver, env_int("CUMF_ALS_CG_ITERS", 6), fused,
                       env_int("CUMF_ALS_EXACT_TEST_GRID", 0), env_int("CUMF_ALS_SURPASS_NAN", 0),
                       env_int("CUMF_ALS_QUIET", 0), nullptr);
}

// C++ linkage, the reference's own symbol (_Z5doALSPKiS0_PKfS0_S0_S2_S0_PfS3_S0_S0_S2_iiillfiiii).
float doALS(const int* csrRowIndexHostPtr, const int* csrColIndexHostPtr, const float* csrValHostPtr,
            const int* cscRowIndexHostPtr, const int* cscColIndexHostPtr, const float* cscValHostPtr,
            const int* cooRowIndexHostPtr, float* thetaTHost, float* XTHost, const int* cooRowIndexTestHostPtr,
            const int* cooColIndexTestHostPtr, const float* cooValHostTestPtr, const int m, const int n, const int f,
            const long nnz, const long nnz_test, const float lambda, const int ITERS, const int X_BATCH,
            const int THETA_BATCH, const int DEVICEID) {
  return cumf_doALS(csrRowIndexHostPtr, csrColIndexHostPtr, csrValHostPtr, cscRowIndexHostPtr, cscColIndexHostPtr,
                    cscValHostPtr, cooRowIndexHostPtr, thetaTHost, XTHost, cooRowIndexTestHostPtr,
                    cooColIndexTestHostPtr, cooValHostTestPtr, m, n, f, nnz, nnz_test, lambda, ITERS, X_BATCH,
                    THETA_BATCH, DEVICEID);
}
