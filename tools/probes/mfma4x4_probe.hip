// Layout / exactness probe of v_mfma_f32_4x4x1_16B_f32 on gfx950 (hipcc --offload-arch=gfx950).
// Prints, for lane l and register r, which (block, row, col) product D holds, by feeding
// A[l] = 1000 + l, B[l] = 1 + l/1000.f patterns and decoding.  Expected (CDNA3 ISA):
// D[lane = 4*b + j][reg = i] = A[4*b + i] * B[4*b + j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d, int reps) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l + 64 * r], b[l + 64 * r], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[4 * l + r] = acc[r];
}
int main() {
  const int reps = 3;
  float ha[64 * reps], hb[64 * reps], hd[256];
  for (int i = 0; i < 64 * reps; ++i) { ha[i] = 1.0f + 0.37f * (i % 61) + 0.001f * (i / 64); hb[i] = 2.0f - 0.11f * (i % 53); }
  float *a, *b, *d;
  hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&d, sizeof(hd));
  hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d, reps);
  hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l / 4, j = l % 4, i = r;
      float want = 0.f;
      for (int k = 0; k < reps; ++k) want = __builtin_fmaf(ha[64 * k + 4 * blk + i], hb[64 * k + 4 * blk + j], want);
      if (want != hd[4 * l + r]) { if (bad < 8) printf("lane %d reg %d got %.9g want %.9g\n", l, r, hd[4 * l + r], want); ++bad; }
    }
  printf("mismatches: %d of 256 (0 = layout D[4b+j][i] = sum_k fma(A_k[4b+i], B_k[4b+j]) confirmed, bit-exact)\n", bad);
  return 0;
}
