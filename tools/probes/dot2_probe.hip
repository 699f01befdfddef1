// Is v_dot2c_f32_bf16 usable for the exact residual x - bf16(x) of the three-plane split?
// build: hipcc --offload-arch=gfx950 -O3 -w tools/probes/dot2_probe.hip -o tools/_bin/dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  v2 r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return __builtin_bit_cast(unsigned, r);
}
__global__ void k(const float* x, float* ref, float* got, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float a = x[2 * i], b = x[2 * i + 1];
  unsigned H = pack_bf16(a, b);
  float ta = __builtin_bit_cast(float, H << 16), tb = __builtin_bit_cast(float, H & 0xffff0000u);
  ref[2 * i] = a - ta;
  ref[2 * i + 1] = b - tb;
  bf16x2 hv = __builtin_bit_cast(bf16x2, H);
  unsigned slo = 0x0000BF80u, shi = 0xBF800000u;  // (-1, 0) and (0, -1) as bf16 pairs, kept in registers
  asm volatile("" : "+v"(slo), "+v"(shi));
  got[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2, slo), a, false);
  got[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2, shi), b, false);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
    if (i % 3 == 0) {  // moderate magnitudes
      float f = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
      memcpy(&u, &f, 4);
    }
    unsigned e = (u >> 23) & 0xff;
    if (e == 0xff) u &= 0x7f7fffffu;  // no inf / nan
    memcpy(&h[i], &u, 4);
  }
  float *x, *r, *g;
  hipMalloc(&x, n * 4); hipMalloc(&r, n * 4); hipMalloc(&g, n * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(x, r, g, n);
  std::vector<float> hr(n), hg(n);
  hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hg.data(), g, n * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0, bad_odd = 0;
  for (int i = 0; i < n; ++i) {
    if (memcmp(&hr[i], &hg[i], 4)) {
      ++bad;
      bad_odd += i & 1;
      unsigned u; memcpy(&u, &h[i], 4);
      const unsigned e = (u >> 23) & 0xff;
      if (e > 30 && e < 250) {
        if (bad_normal++ < 8) printf("x %.9g (exp %u) ref %.9g got %.9g\n", h[i], e, hr[i], hg[i]);
      }
    }
  }
  printf("mismatches %ld of %d (odd slots %ld); with a mid-range exponent: %ld\n", bad, n, bad_odd, bad_normal);
  return 0;
}
