/*
 * als.h -- drop-in declaration of the cuMF/cumf_als ALS entry point.
 *
 * Replaces the reference's als.h:676-681.  The symbol is exported from libALS.so
 * with C++ linkage under the same Itanium-mangled name the reference's own
 * callers bind (`main.cpp:141-146`, `tensorflow/als_tf.cc:33-38`):
 *     _Z5doALSPKiS0_PKfS0_S0_S2_S0_PfS3_S0_S0_S2_iiillfiiii
 * plus the `extern "C"` alias `cumf_doALS` in cumf_als_capi.h.
 *
 * Contract: all pointers are caller-owned HOST pointers (pinned or
 * pageable); thetaTHost (n*f) and XTHost (m*f) are in/out, row-contiguous
 * f-vectors; returns the final test RMSE; prints the reference's stdout lines;
 * owns and frees all device memory; never resets the device.
 *
 * T10 (als.h:37) is kept because main.cpp:33 checks `f % T10`.
 */
#ifndef ALS_H_
#define ALS_H_

#define T10 10

float doALS(const int* csrRowIndexHostPtr, const int* csrColIndexHostPtr, const float* csrValHostPtr,
            const int* cscRowIndexHostPtr, const int* cscColIndexHostPtr, const float* cscValHostPtr,
            const int* cooRowIndexHostPtr, float* thetaTHost, float* XTHost,
            const int* cooRowIndexTestHostPtr, const int* cooColIndexTestHostPtr,
            const float* cooValHostTestPtr, const int m, const int n, const int f, const long nnz,
            const long nnz_test, const float lambda, const int ITERS, const int X_BATCH,
            const int THETA_BATCH, const int DEVICEID);

#endif /* ALS_H_ */
