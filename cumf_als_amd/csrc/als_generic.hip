// als_generic.hip -- the solve path for ANY even f above the tile kernels' range (f > 207; round 6).
//
// The reference's generic kernel takes every f % 10 == 0 (get_hermitianT10, als.cu:575-659; launch shape als.cu:766-767;
// CLI check main.cpp:32-36), e.g. `./main ... 250 ...`.  The tile kernels of this library stop at NB = 13 feature blocks
// (f <= 207); above that the reference's own data flow runs -- the Gram batch in HBM, then a batched solver
// (als.cu:782-831) -- on two plain kernels.  Slow on purpose-built hardware terms (no MFMA, the rows re-gathered once per
// 4 096 Gram entries), correct for every f, and in the reference's own operation order:
//   gram_generic_kernel   one workgroup per row; every Gram entry (and the right-hand side, als.cu:750-757) is ONE fmaf chain
//                         over the row's ratings in their order -- the chain one reference thread evaluates (als.h:39-143) --
//                         written to both triangles from the upper entry, lambda * n_u on the diagonal (als.cu:545-557);
//   lu_global_kernel      unpivoted Doolittle LU + the two triangular solves (cublasSgetrfBatched(Pivot = NULL) +
//                         cublasSgetrsBatched, als.cu:77,98 / 146,166) in global memory, the elimination's (i, j) updates
//                         spread over the workgroup, every element's own operation sequence unchanged.
// CG above f = 207 is cg_global_kernel (als_kernels.hip).
#include <hip/hip_runtime.h>

#include "als_internal.h"

namespace cumf {

namespace {

constexpr int kGenThreads = 256;
constexpr int kGenAcc = 16;                      // Gram entries per thread and pass
constexpr int kGenPass = kGenThreads * kGenAcc;  // entries per pass over the row
constexpr int kGenWindow = 16;                   // ratings staged per window

// entry e of the list [upper triangle row by row | right-hand side] -> (i, j); j == f marks the right-hand side
__device__ __forceinline__ void gen_entry(long e, int f, int& i, int& j) {
  const long tri = (long)f * (f + 1) / 2;
  if (e >= tri) {
    i = (int)(e - tri);
    j = f;
    return;
  }
  // row i starts at s(i) = i f - i (i - 1) / 2; solve s(i) <= e by the quadratic, then correct
  const double ff = (double)f + 0.5;
  int r = (int)(ff - sqrt(ff * ff - 2.0 * (double)e));
  if (r < 0) r = 0;
  if (r > f - 1) r = f - 1;
  auto start = [&](int q) { return (long)q * f - (long)q * (q - 1) / 2; };
  while (r > 0 && start(r) > e) --r;
  while (r + 1 < f && start(r + 1) <= e) ++r;
  i = r;
  j = r + (int)(e - start(r));
}

__global__ __launch_bounds__(kGenThreads) void gram_generic_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // kGenWindow x (f + 1): factor row | rating
  const int f = a.f, ld = f + 1;
  const int item = blockIdx.x;
  const int row = a.item_row[item];
  const long long begin = a.item_begin[item];
  const int len = a.item_len[item];
  const float reg = (float)a.item_rowlen[item] * a.lambda;  // als.cu:547: (end - start) * lambda
  float* tt = a.tt + (size_t)(row - a.row_begin) * f * f;
  float* rhs = a.rhs ? a.rhs + (size_t)(row - a.row_begin) * f : nullptr;
  const long entries = (long)f * (f + 1) / 2 + f;
  for (long base = 0; base < entries; base += kGenPass) {
    int ei[kGenAcc], ej[kGenAcc];
    float acc[kGenAcc];
#pragma unroll
    for (int q = 0; q < kGenAcc; ++q) {
      const long e = base + (long)q * kGenThreads + threadIdx.x;
      acc[q] = 0.f;
      if (e < entries)
        gen_entry(e, f, ei[q], ej[q]);
      else
        ei[q] = ej[q] = -1;
    }
    for (int w0 = 0; w0 < len; w0 += kGenWindow) {
      const int wn = len - w0 < kGenWindow ? len - w0 : kGenWindow;
      __syncthreads();  // the previous window has been consumed
      for (int t = threadIdx.x; t < wn * ld; t += kGenThreads) {
        const int k = t / ld, c = t - k * ld;
        const long long pos = begin + w0 + k;
        smem[t] = c < f ? a.gather[(size_t)(unsigned)a.colidx[pos] * f + c] : (a.val ? a.val[pos] : 0.f);
      }
      __syncthreads();
      for (int k = 0; k < wn; ++k) {  // the ratings in their order: one fmaf chain per entry
        const float* th = smem + k * ld;
#pragma unroll
        for (int q = 0; q < kGenAcc; ++q)
          if (ei[q] >= 0) acc[q] = fmaf(th[ei[q]], th[ej[q]], acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < kGenAcc; ++q) {
      const int i = ei[q], j = ej[q];
      if (i < 0) continue;
      if (j == f) {
        if (rhs) rhs[i] = acc[q];
      } else {
        const float v = i == j ? acc[q] + reg : acc[q];
        tt[(size_t)i * f + j] = v;
        if (i != j) tt[(size_t)j * f + i] = v;  // both triangles from one accumulator (als.h:39-143)
      }
    }
  }
}

// One workgroup per system; A (f x f, row-major, overwritten with the factors) and b -> x, all in global memory.
__global__ __launch_bounds__(kGenThreads) void lu_global_kernel(float* __restrict__ A, const float* __restrict__ b,
                                                                float* __restrict__ x, int f) {
  extern __shared__ __attribute__((aligned(16))) float y[];  // f: the running right-hand side
  float* As = A + (size_t)blockIdx.x * f * f;
  const float* bs = b + (size_t)blockIdx.x * f;
  float* xs = x + (size_t)blockIdx.x * f;
  const int tid = threadIdx.x;
  for (int k = 0; k < f; ++k) {
    const float piv = As[(size_t)k * f + k];
    // multipliers of column k (rows i > k), then the trailing update, each (i, j) by one thread
    for (int i = k + 1 + tid; i < f; i += kGenThreads) As[(size_t)i * f + k] = As[(size_t)i * f + k] / piv;
    __syncthreads();
    const int n = f - k - 1;
    for (long t = tid; t < (long)n * n; t += kGenThreads) {
      const int i = k + 1 + (int)(t / n), j = k + 1 + (int)(t % n);
      As[(size_t)i * f + j] = fmaf(-As[(size_t)i * f + k], As[(size_t)k * f + j], As[(size_t)i * f + j]);
    }
    __syncthreads();
  }
  for (int i = tid; i < f; i += kGenThreads) y[i] = bs[i];
  __syncthreads();
  // L y = b: row i takes its terms in ascending j
  for (int j = 0; j < f; ++j) {
    const float yj = y[j];
    for (int i = j + 1 + tid; i < f; i += kGenThreads) y[i] = fmaf(-As[(size_t)i * f + j], yj, y[i]);
    __syncthreads();
  }
  // U x = y, column-oriented: x_j is final, then every y_i (i < j) loses U_ij x_j -- row i accumulates in descending j
  for (int j = f - 1; j >= 0; --j) {
    if (tid == 0) y[j] = y[j] / As[(size_t)j * f + j];
    __syncthreads();
    const float xj = y[j];
    for (int i = tid; i < j; i += kGenThreads) y[i] = fmaf(-As[(size_t)i * f + j], xj, y[i]);
    __syncthreads();
  }
  for (int i = tid; i < f; i += kGenThreads) xs[i] = y[i];
}

}  // namespace

hipError_t launch_gram_generic(const KernelArgs& a, long n_items, hipStream_t stream) {
  if (n_items <= 0) return hipSuccess;
  if (a.tt == nullptr || a.tt_half || a.tt_packed) return hipErrorInvalidValue;  // the fp32 f x f batch only
  const size_t lds = (size_t)kGenWindow * (a.f + 1) * sizeof(float);
  hipLaunchKernelGGL(gram_generic_kernel, dim3((unsigned)n_items), dim3(kGenThreads), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_lu_global(float* A, const float* b, float* x, long batch, int f, hipStream_t stream) {
  if (batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(lu_global_kernel, dim3((unsigned)batch), dim3(kGenThreads), (size_t)f * sizeof(float), stream, A, b, x, f);
  return hipGetLastError();
}

}  // namespace cumf
