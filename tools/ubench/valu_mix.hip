// valu_mix.hip — does a "2-cycle" VALU op (v_and/v_mov/v_add_u32) issued between 4-cycle ops (v_mad_u64_u32)
// cost 2 or 4 cycles of a SIMD?  Wall-derived cycles per wave-instruction at 2, 4 and 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 4096;
#define KERN(NAME, BODY, NINSTR)                                                             \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                  \
  uint64_t r[4]; uint32_t s[4];                                                              \
  for (int i = 0; i < 4; ++i) { r[i] = (uint64_t)(seed + threadIdx.x + i) * 0x9E3779B97F4A7C15ull; s[i] = seed + i + threadIdx.x; } \
  uint32_t a = seed | 0x10001u;                                                              \
  for (int it = 0; it < ITER; ++it) {                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile(BODY : "+v"(r[i]), "+v"(s[i]) : "v"(a) : "vcc"); \
  }                                                                                          \
  uint64_t x = 0; for (int i = 0; i < 4; ++i) x ^= r[i] + s[i];                              \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x ^ (x >> 32));                    \
}                                                                                            \
static const int NAME##_n = NINSTR;
KERN(k_mad2,      "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0", 2)
KERN(k_mad_and,   "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_and_b32 %1, %2, %1", 2)
KERN(k_mad_mov,   "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mov_b32 %1, %2", 2)
KERN(k_mad_add,   "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_add_u32 %1, %2, %1", 2)
KERN(k_and2,      "v_and_b32 %1, %2, %1\n\tv_xor_b32 %1, %2, %1", 2)
KERN(k_mad_2and,  "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_and_b32 %1, %2, %1\n\tv_xor_b32 %1, %2, %1", 3)
KERN(k_mad_nop,   "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\ts_nop 0", 2)
KERN(k_mad_lsh64, "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_lshl_add_u64 %0, %0, 1, %0", 2)
typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  uint32_t* out; CHECK(hipMalloc(&out, sizeof(uint32_t) * cus * 8 * 4 * 256));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct { const char* n; kern_t k; int ni; } es[] = {
    {"mad,mad", k_mad2, k_mad2_n}, {"mad,and", k_mad_and, k_mad_and_n}, {"mad,mov", k_mad_mov, k_mad_mov_n},
    {"mad,add_u32", k_mad_add, k_mad_add_n}, {"and,xor", k_and2, k_and2_n}, {"mad,and,xor", k_mad_2and, k_mad_2and_n},
    {"mad,s_nop0", k_mad_nop, k_mad_nop_n}, {"mad,lshl_add_u64", k_mad_lsh64, k_mad_lsh64_n}};
  printf("%-18s %12s %12s %12s   (ns-derived cycles per GROUP of instructions per SIMD at 2.25 GHz; waves/SIMD = 2, 4, 8)\n", "pattern", "2w", "4w", "8w");
  for (auto& e : es) {
    printf("%-18s", e.n);
    for (int bpc : {2, 4, 8}) {
      int blocks = cus * bpc * 4;           // 4 rounds of resident blocks
      e.k<<<blocks, 256>>>(out, 7u); CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0)); e.k<<<blocks, 256>>>(out, 7u); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
      }
      // groups executed per SIMD = (blocks*4 waves / (cus*4 SIMDs)) * ITER*4 ; note with bpc<8 only bpc waves are co-resident per SIMD only if limited -> use LDS? (approximation: launch bounds only)
      double groups_per_simd = (double)blocks * 4 / (cus * 4) * ITER * 4;
      printf(" %12.2f", best * 1e-3 * 2.25e9 / groups_per_simd);
    }
    printf("\n");
  }
  return 0;
}
