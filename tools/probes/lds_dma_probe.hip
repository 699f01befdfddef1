// Probe of global_load_lds_dword semantics on gfx950 (tools only): where does lane l's dword land
// for a given LDS base pointer and instruction offset?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out, int stride) {
  extern __shared__ float lds[];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  const float* p = in + (size_t)l * stride;  // lane l gathers from its own row
  // load A: base = lds + 64 floats, imm = 64 bytes ; load B: base = lds + 256 floats, imm = 0
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                   (__attribute__((address_space(3))) void*)(lds + 64), 4, 64, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 1),
                                   (__attribute__((address_space(3))) void*)(lds + 256), 4, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  const int stride = 100;
  std::vector<float> h(64 * stride);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float *din, *dout;
  hipMalloc(&din, h.size() * 4);
  hipMalloc(&dout, 1024 * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, din, dout, stride);
  std::vector<float> o(1024);
  hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
  // expected if imm applies to both: A lands at float 64 + 16 + l with value in[l*stride + 16]
  int okA = 0, okB = 0;
  for (int l = 0; l < 64; ++l) {
    okA += o[80 + l] == (float)(l * stride + 16);
    okB += o[256 + l] == (float)(l * stride + 1);
  }
  printf("LDS_DMA_PROBE A(imm both)=%d/64 B(plain)=%d/64\n", okA, okB);
  for (int i = 0; i < 1024; ++i)
    if (o[i] != -1.f && !((i >= 80 && i < 144) || (i >= 256 && i < 320))) { printf("unexpected write at %d = %g\n", i, o[i]); break; }
  return 0;
}
