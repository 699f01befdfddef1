// tr16_dma_probe.hip -- pins the two gfx950 instructions the pre-split Gram stage (als_wave.hip, kArithPre) is built on:
//   (1) global_load_lds_dwordx4: lane l's 16 bytes land at M0 + instruction offset + 16 l, exec-masked lanes write
//       nothing, the instruction offset moves BOTH the global and the LDS address (as the dword form does);
//   (2) ds_read_b64_tr_b16: in every 16-lane group, lane 4 a + b receives as element j the 16-bit element b of the
//       8-byte piece addressed by lane 4 j + a of the same group (a 4 x 4 transpose of 16-bit elements across lanes).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_dma_probe.hip -o tools/_bin/tr16_dma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr;
typedef const __attribute__((address_space(1))) void* gptr;

__global__ void dma_probe(const unsigned* g, const int* lane_src, unsigned* out, int masked_from) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) smem[i] = 0xdeadbeefu;
  __syncthreads();
  // lane l fetches the 16 bytes at g + 4 * lane_src[l] (+ instruction offset 32 B), LDS base smem + 64 dwords
  if (l < masked_from)
    __builtin_amdgcn_global_load_lds((gptr)(g + 4 * lane_src[l]), (lptr)(smem + 64), 16, 32, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = l; i < 1024; i += 64) out[i] = smem[i];
}

__global__ void tr_probe(const int* lane_piece, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int l = threadIdx.x;
  unsigned short* h = reinterpret_cast<unsigned short*>(smem);
  for (int i = l; i < 2048; i += 64) h[i] = (unsigned short)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(smem) + lane_piece[l]);
  for (int e = 0; e < 4; ++e) out[4 * l + e] = (unsigned short)v[e];
}

// throughput of the transposing read against ds_read_b64 on the same addresses (cycles per wave instruction)
template <bool TR>
__global__ void tr_rate(const int* lane_piece, unsigned* out, long long* cycles, int stride_pieces) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int l = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) smem[i] = i;
  __syncthreads();
  auto* base = (__attribute__((address_space(3))) s16x4*)(smem) + lane_piece[l];
  s16x4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s16x4 v;
      if constexpr (TR)
        v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(base + k * stride_pieces);
      else
        v = *(base + k * stride_pieces);
      acc += v;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = (unsigned short)acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

#define CK(x)                                                             \
  do {                                                                    \
    hipError_t e = (x);                                                   \
    if (e != hipSuccess) {                                                \
      printf("%s: %s\n", #x, hipGetErrorString(e));                       \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main() {
  int fails = 0;
  {  // (1) LDS-DMA dwordx4
    std::vector<unsigned> g(4096);
    for (int i = 0; i < 4096; ++i) g[i] = 0x10000u + i;
    std::vector<int> src(64);
    for (int l = 0; l < 64; ++l) src[l] = (l * 37 + 11) % 200;  // scattered 16-byte pieces
    unsigned *dg, *dout;
    int* dsrc;
    CK(hipMalloc(&dg, 4096 * 4));
    CK(hipMalloc(&dout, 1024 * 4));
    CK(hipMalloc(&dsrc, 64 * 4));
    CK(hipMemcpy(dg, g.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsrc, src.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 4096, 0, dg, dsrc, dout, 48);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> out(1024);
    CK(hipMemcpy(out.data(), dout, 1024 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
      unsigned want = 0xdeadbeefu;
      const int rel = i - 64 - 8;  // LDS base + 64 dwords, + the instruction offset of 32 bytes
      if (rel >= 0 && rel < 4 * 48) {
        const int l = rel / 4, e = rel % 4;
        want = 0x10000u + 4 * src[l] + 8 + e;  // global side moved by the same 32 bytes
      }
      if (out[i] != want) {
        if (bad < 8) printf("  dma: lds dword %d = %08x, expected %08x\n", i, out[i], want);
        ++bad;
      }
    }
    printf("global_load_lds_dwordx4: lane l -> M0 + offset + 16 l, masked lanes silent, offset moves both: %s (%d mismatches)\n",
           bad ? "NO" : "yes", bad);
    fails += bad != 0;
  }
  {  // (2) ds_read_b64_tr_b16
    std::vector<int> piece(64);
    for (int l = 0; l < 64; ++l) piece[l] = (l * 29 + 5) % 256;  // scattered 8-byte pieces
    int* dp;
    unsigned* dout;
    CK(hipMalloc(&dp, 64 * 4));
    CK(hipMalloc(&dout, 256 * 4));
    CK(hipMemcpy(dp, piece.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 4096, 0, dp, dout);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> out(256);
    CK(hipMemcpy(out.data(), dout, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, a = (l & 15) >> 2, b = l & 3;
        const unsigned want = 4 * piece[16 * g + 4 * j + a] + b;
        if (out[4 * l + j] != want) {
          if (bad < 8) printf("  tr: lane %d elem %d = %u, expected %u\n", l, j, out[4 * l + j], want);
          ++bad;
        }
      }
    printf("ds_read_b64_tr_b16: lane 4a+b elem j <- element b of the piece of lane 4j+a (same 16-lane group): %s (%d mismatches)\n",
           bad ? "NO" : "yes", bad);
    fails += bad != 0;
  }
  {  // (3) rate of the transposing read in the layouts the Gram stage uses
    struct Case { const char* name; int pitch_rating; int skew; };
    // lane (g, 4 j + a): piece address = chunk(E = 2 g) + pitch * j + 8 a [+ 32-byte skew for odd g]
    const Case cases[] = {{"192-B rating pitch, no skew", 192, 0}, {"192-B rating pitch, 32-B skew for g odd", 192, 32},
                          {"contiguous (8 l)", 0, 0}};
    for (const Case& c : cases) {
      std::vector<int> piece(64);
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, j = (l & 15) >> 2, a = l & 3;
        int bytes = c.pitch_rating ? (2 * g) * 2304 + c.pitch_rating * j + 8 * a + ((g & 1) ? c.skew : 0) : 8 * l;
        piece[l] = bytes / 8;
      }
      int* dp;
      unsigned* dout;
      long long* dcy;
      CK(hipMalloc(&dp, 64 * 4));
      CK(hipMalloc(&dout, 256 * 4));
      CK(hipMalloc(&dcy, 8 * 1024));
      CK(hipMemcpy(dp, piece.data(), 64 * 4, hipMemcpyHostToDevice));
      for (int tr = 0; tr < 2; ++tr) {
        for (int waves : {1, 4, 8}) {
          if (tr)
            hipLaunchKernelGGL(tr_rate<true>, dim3(256), dim3(64 * waves), 32768, 0, dp, dout, dcy, 4);
          else
            hipLaunchKernelGGL(tr_rate<false>, dim3(256), dim3(64 * waves), 32768, 0, dp, dout, dcy, 4);
          CK(hipDeviceSynchronize());
          std::vector<long long> cy(256);
          CK(hipMemcpy(cy.data(), dcy, 256 * 8, hipMemcpyDeviceToHost));
          double s = 0;
          for (long long v : cy) s += (double)v;
          printf("  %-42s %-18s %d wave(s)/WG: %.1f clock ticks per read per wave\n", c.name,
                 tr ? "ds_read_b64_tr_b16" : "ds_read_b64", waves, s / 256 / (256.0 * 16));
        }
      }
    }
  }
  printf(fails ? "PROBE FAILED\n" : "PROBE OK\n");
  return fails;
}
