// valu_rates.hip — gfx950 issue-rate probe for the integer instructions the
// modular-multiply datapath is built from.  Design input only (not shipped in
// the library): the modmul formulation in tiny_ntt_amd/csrc is chosen from the
// numbers this prints (see DESIGN.md "Instruction budget").
//
// Each kernel runs ITER iterations of 8 independent dependency chains of ONE
// instruction (inline asm, so the compiler cannot fold or reorder it away).
// Reported: cycles per wave-instruction per SIMD at 1, 2, 4 and 8 waves/SIMD,
// using the in-kernel s_memtime delta of the slowest wave.
//
// build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cstddef>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int ITER = 4096;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- 32-bit one-dest two-src form: v_op d, a, d
#define K32(NAME, ASMSTR)                                                        \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint64_t* cyc, uint32_t seed) { \
  asm volatile("s_mov_b32 s4, 0x12345\n\ts_mov_b64 s[10:11], 0" ::: "s4", "s10", "s11");           \
  uint32_t r[8];                                                                 \
  for (int i = 0; i < 8; ++i) r[i] = seed + threadIdx.x * 8 + i;                 \
  uint32_t a = seed | 0x10001u;                                                  \
  uint64_t t0 = __builtin_amdgcn_s_memtime();                                    \
  for (int it = 0; it < ITER; ++it) {                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(a) : "vcc", "s10", "s11");                                \
  }                                                                              \
  uint64_t t1 = __builtin_amdgcn_s_memtime();                                    \
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;\
}

K32(add_u32,      "v_add_u32 %0, %1, %0")
K32(and_b32,      "v_and_b32 %0, %1, %0")
K32(mul_lo_u32,   "v_mul_lo_u32 %0, %1, %0")
K32(mul_hi_u32,   "v_mul_hi_u32 %0, %1, %0")
K32(mul_u32_u24,  "v_mul_u32_u24 %0, %1, %0")
K32(mul_hi_u24,   "v_mul_hi_u32_u24 %0, %1, %0")
K32(mad_u32_u24,  "v_mad_u32_u24 %0, %1, %0, %0")
K32(add3_u32,     "v_add3_u32 %0, %1, %0, %0")
K32(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
K32(alignbit,     "v_alignbit_b32 %0, %1, %0, 28")
K32(fma_f32,      "v_fma_f32 %0, %1, %0, %0")
K32(addc_pair,    "v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc")
K32(cndmask,      "v_cndmask_b32 %0, %1, %0, vcc")
K32(bfe_u32,      "v_bfe_u32 %0, %0, 3, 28")
K32(dpp_mov,      "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
K32(dpp_add,      "v_add_u32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf")
K32(bpermute,     "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")
K32(swizzle,      "ds_swizzle_b32 %0, %0 offset:0x041F\n\ts_waitcnt lgkmcnt(0)")

K32(sub_u32,      "v_sub_u32 %0, %1, %0")
K32(or_b32,       "v_or_b32 %0, %1, %0")
K32(xor_b32,      "v_xor_b32 %0, %1, %0")
K32(lshlrev_b32,  "v_lshlrev_b32 %0, 3, %0")
K32(lshrrev_b32,  "v_lshrrev_b32 %0, 3, %0")
K32(mov_b32,      "v_mov_b32 %0, %1")
K32(min_u32,      "v_min_u32 %0, %1, %0")
K32(add_co_only,  "v_add_co_u32 %0, vcc, %1, %0")
K32(addc_only,    "v_addc_co_u32 %0, vcc, %0, %1, vcc")
K32(add_co_sgpr,  "v_add_co_u32 %0, s[10:11], %1, %0")
K32(cmp_cndmask,  "v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %0, vcc")
K32(cmp_only,     "v_cmp_lt_u32 vcc, %1, %0")
K32(cmp_sgpr,     "v_cmp_lt_u32 s[10:11], %1, %0")
K32(add_sgpr_src, "v_add_u32 %0, s4, %0")
K32(and_or_b32,   "v_and_or_b32 %0, %0, %1, %1")
K32(add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 1")
K32(perm_b32,     "v_perm_b32 %0, %0, %1, %1")
K32(mul_lo_sgpr,  "v_mul_lo_u32 %0, s4, %0")
K32(and_sgpr,     "v_and_b32 %0, s4, %0")
K32(and_literal,  "v_and_b32 %0, 0x0fffffff, %0")
K32(lshr_sgpr,    "v_lshrrev_b32 %0, s4, %0")
K32(lshr_28,      "v_lshrrev_b32 %0, 28, %0")
K32(sub_sgpr,     "v_sub_u32 %0, s4, %0")
K32(mov_sgpr,     "v_mov_b32 %0, s4")
K32(xor_inline,   "v_xor_b32 %0, 8, %0")
K32(sub_co_subb,  "v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %0, vcc, %0, %1, vcc")
K32(pk_fma_f32_half, "v_fma_f32 %0, %0, %1, %1")
K32(mad_u32_u24_v3,  "v_mad_u32_u24 %0, %0, %1, %1")
// every SIMD occupied (32 waves per CU) but nothing for the vector ALU to do: what the package draws with the shader clock up
K32(snop,            "s_nop 15")

// ---- 64-bit forms
#define K64(NAME, ASMSTR)                                                        \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint64_t* cyc, uint32_t seed) { \
  asm volatile("s_mov_b32 s4, 0x12345\n\ts_mov_b64 s[10:11], 0" ::: "s4", "s10", "s11");           \
  uint64_t r[8];                                                                 \
  for (int i = 0; i < 8; ++i) r[i] = (uint64_t)(seed + threadIdx.x * 8 + i) * 0x9E3779B97F4A7C15ull; \
  uint32_t a = seed | 0x10001u;                                                  \
  uint64_t a64 = ((uint64_t)a << 32) | a;                                        \
  uint64_t t0 = __builtin_amdgcn_s_memtime();                                    \
  for (int it = 0; it < ITER; ++it) {                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(a64) : "vcc", "s10", "s11");                      \
  }                                                                              \
  uint64_t t1 = __builtin_amdgcn_s_memtime();                                    \
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));        \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;\
}

K64(mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %1, %0")
K64(lshlrev_b64,  "v_lshlrev_b64 %0, 3, %0")
K64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %2")
K64(fma_f64,      "v_fma_f64 %0, %2, %0, %0")
K64(mul_f64,      "v_mul_f64 %0, %2, %0")
K64(add_f64,      "v_add_f64 %0, %2, %0")
K64(pk_fma_f32,   "v_pk_fma_f32 %0, %2, %0, %0")
K64(pk_add_f32,   "v_pk_add_f32 %0, %2, %0")
K64(mad_u64_sgpr, "v_mad_u64_u32 %0, vcc, s4, %1, %0")
K64(mad_u64_c0,   "v_mad_u64_u32 %0, vcc, %1, %1, 0")
K64(lshrrev_b64,  "v_lshrrev_b64 %0, 3, %0")

__global__ void k_clock(uint64_t* o) {
  uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = threadIdx.x;
  for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
  uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = c1 - c0; o[1] = r1 - r0; o[2] = x; }
}

// LDS traffic: each lane writes and reads back 8 bytes (a wave moves 512 B per instruction), 8 independent slots
__global__ void __launch_bounds__(256) k_lds_rw64(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  __shared__ uint64_t buf[256 * 8];
  uint64_t r[8];
  for (int i = 0; i < 8; ++i) r[i] = (uint64_t)(seed + threadIdx.x * 8 + i) * 0x9E3779B97F4A7C15ull;
  buf[threadIdx.x] = r[0];
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(&buf[0]) + threadIdx.x * 8u;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    _Pragma("unroll") for (int i = 0; i < 8; ++i)
      asm volatile("ds_write_b64 %1, %0 offset:%2\n\tds_read_b64 %0, %1 offset:%2" : "+v"(r[i]) : "v"(base), "n"(i * 2048) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= (uint32_t)r[i] ^ (uint32_t)(r[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

// v_mad_u64_u32 with BOTH multiplicands changing every instruction (the low and high halves of the neighbouring
// accumulators, as in a butterfly's multiply-add chain): the multiplier array toggles like it does on real data
// (in K64(mad_u64_u32) above one multiplicand pair is constant, which reads low)
__global__ void __launch_bounds__(256) k_mad_u64_varying(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint64_t r[8];
  for (int i = 0; i < 8; ++i) r[i] = (uint64_t)(seed + threadIdx.x * 8 + i) * 0x9E3779B97F4A7C15ull;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    _Pragma("unroll") for (int i = 0; i < 8; ++i)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"((uint32_t)r[(i + 1) & 7]), "v"((uint32_t)(r[(i + 3) & 7] >> 32)) : "vcc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

// Does the operand ORDER of the multiply matter for power (one port of a multiplier array is usually the recoded one)?
// varying x constant with the constant (2^29, like SplitK::mulp, or a dense 31-bit one, like a twiddle half) as src0 or src1.
#define KORD(NAME, CONSTV, ASMSTR)                                                \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint64_t* cyc, uint32_t seed) { \
  uint64_t r[8];                                                                 \
  for (int i = 0; i < 8; ++i) r[i] = (uint64_t)(seed + threadIdx.x * 8 + i) * 0x9E3779B97F4A7C15ull; \
  uint32_t c = CONSTV;                                                           \
  asm volatile("" : "+s"(c));                                                    \
  uint64_t t0 = __builtin_amdgcn_s_memtime();                                    \
  for (int it = 0; it < ITER; ++it) {                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"((uint32_t)r[(i + 1) & 7]), "s"(c) : "vcc"); \
  }                                                                              \
  uint64_t t1 = __builtin_amdgcn_s_memtime();                                    \
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));        \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;\
}
KORD(mad_var_x_sparse, 0x20000000u, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KORD(mad_sparse_x_var, 0x20000000u, "v_mad_u64_u32 %0, vcc, %2, %1, %0")
KORD(mad_var_x_dense,  0x5A3C96E7u, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KORD(mad_dense_x_var,  0x5A3C96E7u, "v_mad_u64_u32 %0, vcc, %2, %1, %0")

typedef void (*kern_t)(uint32_t*, uint64_t*, uint32_t);
struct Entry { const char* name; kern_t k; int instr_per_slot; };

// usage: valu_rates                      -> the issue-rate table
//        valu_rates spin "<name>" secs flag -> run that one instruction back to back for `secs` seconds (creates `flag`
//                                             while running) so rocm-smi can sample clock and power: energy per instruction
int main(int argc, char** argv) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device=%s cus=%d clock_khz=%d\n", prop.name, cus, prop.clockRate);
  std::vector<Entry> es = {
    {"v_fma_f32", k_fma_f32, 1}, {"v_add_u32", k_add_u32, 1}, {"v_and_b32", k_and_b32, 1},
    {"v_add3_u32", k_add3_u32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1},
    {"v_alignbit_b32", k_alignbit, 1}, {"v_bfe_u32", k_bfe_u32, 1}, {"v_cndmask_b32", k_cndmask, 1},
    {"v_add_co+v_addc (pair)", k_addc_pair, 2},
    {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u24, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1},
    {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"v_sub_u32", k_sub_u32, 1}, {"v_or_b32", k_or_b32, 1}, {"v_xor_b32", k_xor_b32, 1},
    {"v_lshlrev_b32", k_lshlrev_b32, 1}, {"v_lshrrev_b32", k_lshrrev_b32, 1}, {"v_mov_b32", k_mov_b32, 1},
    {"v_min_u32", k_min_u32, 1}, {"v_add_co_u32 (vcc)", k_add_co_only, 1}, {"v_addc_co_u32 (vcc)", k_addc_only, 1},
    {"v_add_co_u32 (sgpr pair)", k_add_co_sgpr, 1}, {"v_cmp+v_cndmask (pair)", k_cmp_cndmask, 2},
    {"v_cmp_lt_u32 vcc", k_cmp_only, 1}, {"v_cmp_lt_u32 sgpr", k_cmp_sgpr, 1},
    {"v_add_u32 sgpr src", k_add_sgpr_src, 1}, {"v_and_or_b32", k_and_or_b32, 1},
    {"v_add_lshl_u32", k_add_lshl_u32, 1}, {"v_perm_b32", k_perm_b32, 1}, {"v_mul_lo_u32 sgpr", k_mul_lo_sgpr, 1},
    {"v_sub_co+v_subb (pair)", k_sub_co_subb, 2}, {"v_fma_f32 3src", k_pk_fma_f32_half, 1},
    {"v_mad_u32_u24 3src", k_mad_u32_u24_v3, 1},
    {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1},
    {"v_mad_u64_u32 sgpr", k_mad_u64_sgpr, 1}, {"v_mad_u64_u32 +0", k_mad_u64_c0, 1},
    {"v_mov_b32_dpp", k_dpp_mov, 1}, {"v_add_u32_dpp row_ror", k_dpp_add, 1},
    {"v_and_b32 sgpr src", k_and_sgpr, 1}, {"v_and_b32 literal", k_and_literal, 1},
    {"v_lshrrev_b32 sgpr shift", k_lshr_sgpr, 1}, {"v_lshrrev_b32 inline 28", k_lshr_28, 1},
    {"v_sub_u32 sgpr src", k_sub_sgpr, 1}, {"v_mov_b32 sgpr src", k_mov_sgpr, 1}, {"v_xor_b32 inline 8", k_xor_inline, 1},
    {"s_nop 15 (no ALU work)", k_snop, 1},
    {"v_mad_u64_u32 varying", k_mad_u64_varying, 1},
    {"mad var x 2^29", k_mad_var_x_sparse, 1}, {"mad 2^29 x var", k_mad_sparse_x_var, 1},
    {"mad var x dense", k_mad_var_x_dense, 1}, {"mad dense x var", k_mad_dense_x_var, 1},
    {"ds_write_b64+ds_read_b64", k_lds_rw64, 2},
    {"ds_bpermute_b32+wait", k_bpermute, 1}, {"ds_swizzle_b32+wait", k_swizzle, 1},
  };
  const int max_blocks = cus * 8 * 4;
  uint32_t* out; uint64_t* cyc;
  CHECK(hipMalloc(&out, sizeof(uint32_t) * max_blocks * 256));
  CHECK(hipMalloc(&cyc, sizeof(uint64_t) * max_blocks * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  if (argc >= 5 && !strcmp(argv[1], "spin")) {
    const Entry* sel = nullptr;
    for (auto& e : es) if (!strcmp(e.name, argv[2])) sel = &e;
    if (!sel) { fprintf(stderr, "unknown instruction %s\n", argv[2]); return 2; }
    const double secs = atof(argv[3]);
    sel->k<<<max_blocks, 256>>>(out, cyc, 12345u); CHECK(hipDeviceSynchronize());
    FILE* f = fopen(argv[4], "w"); if (f) { fputs("go\n", f); fclose(f); }
    double total_ms = 0; long launches = 0;
    while (total_ms < secs * 1e3) {
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 20; ++i) sel->k<<<max_blocks, 256>>>(out, cyc, 12345u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      total_ms += ms; launches += 20;
    }
    remove(argv[4]);
    const double rate = (double)launches * max_blocks * 256 * ITER * 8 * sel->instr_per_slot / (total_ms * 1e-3);
    printf("spin %-28s %.2f Tlane-op/s  %.3f G wave-instr/s\n", sel->name, rate / 1e12, rate / 64e9);
    return 0;
  }
  // effective clock under an all-CU VALU load: d(s_memtime)/d(s_memrealtime) * 100 MHz
  k_clock<<<cus * 8, 256>>>(cyc); CHECK(hipDeviceSynchronize());
  k_clock<<<cus * 8, 256>>>(cyc); CHECK(hipDeviceSynchronize());
  uint64_t hc[3]; CHECK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
  double ghz = (double)hc[0] / (double)hc[1] * 0.1;
  printf("in-kernel clock under load: %.3f GHz (memtime %llu / memrealtime %llu)\n", ghz,
         (unsigned long long)hc[0], (unsigned long long)hc[1]);
  printf("%-28s %12s %12s %12s   (wall-derived: Tlane-op/s, lanes/clk/SIMD at measured clock, cycles per wave-instr)\n",
         "instruction", "Tlane-op/s", "lanes/clk", "cyc/winstr");
  for (auto& e : es) {
    int blocks = max_blocks;               // 4 full waves of 8 blocks/CU (32 waves/CU resident)
    e.k<<<blocks, 256>>>(out, cyc, 12345u);  // warm
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      e.k<<<blocks, 256>>>(out, cyc, 12345u);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    double rate = (double)blocks * 256 * ITER * 8 * e.instr_per_slot / (best * 1e-3);
    double lpc = rate / (cus * 4.0 * ghz * 1e9);
    printf("%-28s %12.2f %12.2f %12.2f\n", e.name, rate / 1e12, lpc, 64.0 / lpc);
  }
  return 0;
}
