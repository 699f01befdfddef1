// Is it safe to write a register of the 128-bit B operand of v_mfma_f32_16x16x32_bf16 straight BEHIND the instruction
// (write-after-read) or straight IN FRONT of it (read-after-write)?  Round 5 met wrong tiles when the compiler's builtin let
// the VALU moves that assemble the next operand quad follow the MFMA directly (profiles/r05/mfma_k32_operand_hazard.txt).
// One wave; operands = small integers in bf16, so every product and sum is exact; the MFMA and the v_mov_b32 next to it sit
// in ONE inline-asm block on fixed physical registers (acc v[0:3], A v[4:7], B v[8:11], junk v12) with N wait states
// between them.  Prints, per variant, how many of the 256 accumulator entries differ from the reference (no move nearby).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_k32_hazard_probe.hip -o tools/_bin/mfma_k32_hazard_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16pair(int lo, int hi) {  // small integers are exact in bf16
  return (__builtin_bit_cast(unsigned, (float)lo) >> 16) | (__builtin_bit_cast(unsigned, (float)hi) & 0xffff0000u);
}

#define LOAD                                                                                                           \
  "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"                                        \
  "v_mov_b32 v4, %4\n\tv_mov_b32 v5, %5\n\tv_mov_b32 v6, %6\n\tv_mov_b32 v7, %7\n\t"                                    \
  "v_mov_b32 v8, %8\n\tv_mov_b32 v9, %9\n\tv_mov_b32 v10, %10\n\tv_mov_b32 v11, %11\n\tv_mov_b32 v12, %12\n\t"          \
  "s_nop 7\n\ts_nop 7\n\t"
#define STORE "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3"
#define MFMA "v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\t"
#define OPERANDS                                                                                                       \
  : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])                                                                     \
  : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(junk)                  \
  : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12"

// VARIANT 0: reference.  1x: junk written to B dword D, N wait states BEHIND the MFMA (the result must not see it).
// 2x: B dword D holds junk and the right value arrives N wait states IN FRONT of the MFMA (the result must see it).
#define WAR(D, NOPS) asm volatile(LOAD MFMA NOPS "v_mov_b32 v" #D ", v12\n\t" STORE OPERANDS)
#define RAW(D, SRC, NOPS) asm volatile(LOAD "v_mov_b32 v" #D ", v12\n\ts_nop 7\n\tv_mov_b32 v" #D ", %" #SRC "\n\t" NOPS MFMA STORE OPERANDS)

template <int VARIANT>
__global__ void probe(float* out) {
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  unsigned a[4], b[4], junk = bf16pair(7, 9);
  float r[4];
  for (int e = 0; e < 4; ++e) {
    a[e] = bf16pair(1 + ((c + 2 * e + g) & 3), 1 + ((c + 2 * e + 1 + g) & 3));
    b[e] = bf16pair(1 + ((c * 3 + 2 * e + g) & 3), 1 + ((c * 3 + 2 * e + 1) & 3));
  }
  if constexpr (VARIANT == 0) asm volatile(LOAD MFMA STORE OPERANDS);
  if constexpr (VARIANT == 108) WAR(8, "");
  if constexpr (VARIANT == 109) WAR(9, "");
  if constexpr (VARIANT == 110) WAR(10, "");
  if constexpr (VARIANT == 111) WAR(11, "");
  if constexpr (VARIANT == 1080) RAW(8, 8, "");
  if constexpr (VARIANT == 1081) RAW(8, 8, "s_nop 0\n\t");
  if constexpr (VARIANT == 1082) RAW(8, 8, "s_nop 1\n\t");
  if constexpr (VARIANT == 1083) RAW(8, 8, "s_nop 2\n\t");
  if constexpr (VARIANT == 1084) RAW(8, 8, "s_nop 3\n\t");
  if constexpr (VARIANT == 1085) RAW(8, 8, "s_nop 4\n\t");
  if constexpr (VARIANT == 1090) RAW(9, 9, "");
  if constexpr (VARIANT == 1091) RAW(9, 9, "s_nop 0\n\t");
  if constexpr (VARIANT == 1092) RAW(9, 9, "s_nop 1\n\t");
  if constexpr (VARIANT == 1093) RAW(9, 9, "s_nop 2\n\t");
  if constexpr (VARIANT == 1094) RAW(9, 9, "s_nop 3\n\t");
  if constexpr (VARIANT == 1095) RAW(9, 9, "s_nop 4\n\t");
  if constexpr (VARIANT == 1100) RAW(10, 10, "");
  if constexpr (VARIANT == 1101) RAW(10, 10, "s_nop 0\n\t");
  if constexpr (VARIANT == 1102) RAW(10, 10, "s_nop 1\n\t");
  if constexpr (VARIANT == 1103) RAW(10, 10, "s_nop 2\n\t");
  if constexpr (VARIANT == 1104) RAW(10, 10, "s_nop 3\n\t");
  if constexpr (VARIANT == 1105) RAW(10, 10, "s_nop 4\n\t");
  if constexpr (VARIANT == 1110) RAW(11, 11, "");
  if constexpr (VARIANT == 1111) RAW(11, 11, "s_nop 0\n\t");
  if constexpr (VARIANT == 1112) RAW(11, 11, "s_nop 1\n\t");
  if constexpr (VARIANT == 1113) RAW(11, 11, "s_nop 2\n\t");
  if constexpr (VARIANT == 1114) RAW(11, 11, "s_nop 3\n\t");
  if constexpr (VARIANT == 1115) RAW(11, 11, "s_nop 4\n\t");
  if constexpr (VARIANT == 1040) RAW(4, 4, "");
  if constexpr (VARIANT == 1041) RAW(4, 4, "s_nop 0\n\t");
  if constexpr (VARIANT == 1042) RAW(4, 4, "s_nop 1\n\t");
  if constexpr (VARIANT == 1043) RAW(4, 4, "s_nop 2\n\t");
  if constexpr (VARIANT == 1044) RAW(4, 4, "s_nop 3\n\t");
  if constexpr (VARIANT == 1045) RAW(4, 4, "s_nop 4\n\t");
  if constexpr (VARIANT == 1050) RAW(5, 5, "");
  if constexpr (VARIANT == 1051) RAW(5, 5, "s_nop 0\n\t");
  if constexpr (VARIANT == 1052) RAW(5, 5, "s_nop 1\n\t");
  if constexpr (VARIANT == 1053) RAW(5, 5, "s_nop 2\n\t");
  if constexpr (VARIANT == 1054) RAW(5, 5, "s_nop 3\n\t");
  if constexpr (VARIANT == 1055) RAW(5, 5, "s_nop 4\n\t");
  if constexpr (VARIANT == 1060) RAW(6, 6, "");
  if constexpr (VARIANT == 1061) RAW(6, 6, "s_nop 0\n\t");
  if constexpr (VARIANT == 1062) RAW(6, 6, "s_nop 1\n\t");
  if constexpr (VARIANT == 1063) RAW(6, 6, "s_nop 2\n\t");
  if constexpr (VARIANT == 1064) RAW(6, 6, "s_nop 3\n\t");
  if constexpr (VARIANT == 1065) RAW(6, 6, "s_nop 4\n\t");
  if constexpr (VARIANT == 1070) RAW(7, 7, "");
  if constexpr (VARIANT == 1071) RAW(7, 7, "s_nop 0\n\t");
  if constexpr (VARIANT == 1072) RAW(7, 7, "s_nop 1\n\t");
  if constexpr (VARIANT == 1073) RAW(7, 7, "s_nop 2\n\t");
  if constexpr (VARIANT == 1074) RAW(7, 7, "s_nop 3\n\t");
  if constexpr (VARIANT == 1075) RAW(7, 7, "s_nop 4\n\t");
  for (int k = 0; k < 4; ++k) out[(4 * g + k) * 16 + c] = r[k];
}

template <int V>
static void run(const char* what, const std::vector<float>& ref, float* d_out, std::vector<float>* keep = nullptr) {
  std::vector<float> h(256);
  int worst = 0;
  for (int rep = 0; rep < 16; ++rep) {  // many launches: a timing-dependent hazard need not show every time
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(64), 0, 0, d_out);
    (void)hipMemcpy(h.data(), d_out, 256 * sizeof(float), hipMemcpyDeviceToHost);
    int bad = 0;
    if (!ref.empty())
      for (int i = 0; i < 256; ++i) bad += h[i] != ref[i];
    worst = bad > worst ? bad : worst;
  }
  if (keep) *keep = h;
  printf("%-70s wrong entries (worst of 16 launches): %d / 256\n", what, worst);
}

int main() {
  float* d_out;
  (void)hipMalloc(&d_out, 256 * sizeof(float));
  std::vector<float> ref;
  run<0>("reference (no move near the MFMA)", ref, d_out, &ref);
  run<108>("WAR: junk into B dword 0 straight behind the MFMA (0 wait states)", ref, d_out);
  run<109>("WAR: junk into B dword 1 straight behind the MFMA (0 wait states)", ref, d_out);
  run<110>("WAR: junk into B dword 2 straight behind the MFMA (0 wait states)", ref, d_out);
  run<111>("WAR: junk into B dword 3 straight behind the MFMA (0 wait states)", ref, d_out);
  run<1080>("RAW: B dword 0 written 0 wait states in front of the MFMA", ref, d_out);
  run<1081>("RAW: B dword 0 written 1 wait states in front of the MFMA", ref, d_out);
  run<1082>("RAW: B dword 0 written 2 wait states in front of the MFMA", ref, d_out);
  run<1083>("RAW: B dword 0 written 3 wait states in front of the MFMA", ref, d_out);
  run<1084>("RAW: B dword 0 written 4 wait states in front of the MFMA", ref, d_out);
  run<1085>("RAW: B dword 0 written 5 wait states in front of the MFMA", ref, d_out);
  run<1090>("RAW: B dword 1 written 0 wait states in front of the MFMA", ref, d_out);
  run<1091>("RAW: B dword 1 written 1 wait states in front of the MFMA", ref, d_out);
  run<1092>("RAW: B dword 1 written 2 wait states in front of the MFMA", ref, d_out);
  run<1093>("RAW: B dword 1 written 3 wait states in front of the MFMA", ref, d_out);
  run<1094>("RAW: B dword 1 written 4 wait states in front of the MFMA", ref, d_out);
  run<1095>("RAW: B dword 1 written 5 wait states in front of the MFMA", ref, d_out);
  run<1100>("RAW: B dword 2 written 0 wait states in front of the MFMA", ref, d_out);
  run<1101>("RAW: B dword 2 written 1 wait states in front of the MFMA", ref, d_out);
  run<1102>("RAW: B dword 2 written 2 wait states in front of the MFMA", ref, d_out);
  run<1103>("RAW: B dword 2 written 3 wait states in front of the MFMA", ref, d_out);
  run<1104>("RAW: B dword 2 written 4 wait states in front of the MFMA", ref, d_out);
  run<1105>("RAW: B dword 2 written 5 wait states in front of the MFMA", ref, d_out);
  run<1110>("RAW: B dword 3 written 0 wait states in front of the MFMA", ref, d_out);
  run<1111>("RAW: B dword 3 written 1 wait states in front of the MFMA", ref, d_out);
  run<1112>("RAW: B dword 3 written 2 wait states in front of the MFMA", ref, d_out);
  run<1113>("RAW: B dword 3 written 3 wait states in front of the MFMA", ref, d_out);
  run<1114>("RAW: B dword 3 written 4 wait states in front of the MFMA", ref, d_out);
  run<1115>("RAW: B dword 3 written 5 wait states in front of the MFMA", ref, d_out);
  run<1040>("RAW: A dword 0 written 0 wait states in front of the MFMA", ref, d_out);
  run<1041>("RAW: A dword 0 written 1 wait states in front of the MFMA", ref, d_out);
  run<1042>("RAW: A dword 0 written 2 wait states in front of the MFMA", ref, d_out);
  run<1043>("RAW: A dword 0 written 3 wait states in front of the MFMA", ref, d_out);
  run<1044>("RAW: A dword 0 written 4 wait states in front of the MFMA", ref, d_out);
  run<1045>("RAW: A dword 0 written 5 wait states in front of the MFMA", ref, d_out);
  run<1050>("RAW: A dword 1 written 0 wait states in front of the MFMA", ref, d_out);
  run<1051>("RAW: A dword 1 written 1 wait states in front of the MFMA", ref, d_out);
  run<1052>("RAW: A dword 1 written 2 wait states in front of the MFMA", ref, d_out);
  run<1053>("RAW: A dword 1 written 3 wait states in front of the MFMA", ref, d_out);
  run<1054>("RAW: A dword 1 written 4 wait states in front of the MFMA", ref, d_out);
  run<1055>("RAW: A dword 1 written 5 wait states in front of the MFMA", ref, d_out);
  run<1060>("RAW: A dword 2 written 0 wait states in front of the MFMA", ref, d_out);
  run<1061>("RAW: A dword 2 written 1 wait states in front of the MFMA", ref, d_out);
  run<1062>("RAW: A dword 2 written 2 wait states in front of the MFMA", ref, d_out);
  run<1063>("RAW: A dword 2 written 3 wait states in front of the MFMA", ref, d_out);
  run<1064>("RAW: A dword 2 written 4 wait states in front of the MFMA", ref, d_out);
  run<1065>("RAW: A dword 2 written 5 wait states in front of the MFMA", ref, d_out);
  run<1070>("RAW: A dword 3 written 0 wait states in front of the MFMA", ref, d_out);
  run<1071>("RAW: A dword 3 written 1 wait states in front of the MFMA", ref, d_out);
  run<1072>("RAW: A dword 3 written 2 wait states in front of the MFMA", ref, d_out);
  run<1073>("RAW: A dword 3 written 3 wait states in front of the MFMA", ref, d_out);
  run<1074>("RAW: A dword 3 written 4 wait states in front of the MFMA", ref, d_out);
  run<1075>("RAW: A dword 3 written 5 wait states in front of the MFMA", ref, d_out);
  return 0;
}
