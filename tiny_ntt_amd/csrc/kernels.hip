// kernels.hip — gfx950 kernels of libtinyntt and their launchers.
//
//  polymul_fused_kernel : K6, the throughput path.  Persistent workgroups, one
//      polynomial pair at a time; coefficients live in VGPRs (2^LPT per thread), the
//      log2(n) radix-2 stages run as ceil(log2 n / LPT) register phases with LDS
//      transposes between them (padded, bank-conflict-free images addressed as
//      base + immediate offset); psi-twist and n^-1 folded into the twiddles
//      (fused_core.h); the same kernel on the x^n - 1 twiddle tables is the cyclic product.
//      HBM traffic per product = read a, read b, write c.
//  ntt_fused_kernel     : the standalone transforms (twist+forward, cg_ntt, cg_intt)
//      on the same machinery, natural order in and out.
//  cg_kernel            : K1-K5/K7, the reference's own constant-geometry dataflow (cg_ntt.py:49-64): cg_kernel_impl.h,
//      instantiated in slices by cg_part.hip; only the dispatcher launch_cg() lives here.
//  fill_lcg_kernel / checksum_kernel : the reference benchmark's input
//      generator and digest (benchmark_ntt_60bit.cpp:79-87,182-188), on device.
#include <hip/hip_runtime.h>
#include "plan.h"
#include "dev_addr.h"

#ifndef TN_ABL_NO_BARRIER
#define TN_ABL_NO_BARRIER 0      // timing ablation: cross-wave transposes without workgroup barriers (wrong results)
#endif
#ifndef TN_ABL_ROWMASK
#define TN_ABL_ROWMASK 0         // timing ablation: rows & mask -> operands and results stay in L2 (no HBM traffic; wrong results)
#endif
#ifndef TN_ABL_NO_GLOBAL
#define TN_ABL_NO_GLOBAL 0       // timing ablation: operands synthesised in registers instead of loaded from HBM
#endif
#ifndef TN_STORE_AT_TOP
#define TN_STORE_AT_TOP 1        // 1: row k's result is stored at the top of iteration k+1 (see the kernel's comment)
#endif
#ifndef TN_NT_STREAM
#define TN_NT_STREAM 1           // 1: non-temporal loads/stores for the streamed operands a, b, c
#endif
#ifndef TN_FUSED_MIN_WAVES
#define TN_FUSED_MIN_WAVES 4     // waves per SIMD the register allocator must leave room for (4 -> <= 128 VGPRs)
#endif
#ifndef TN_POLYMUL60_WAVES
#define TN_POLYMUL60_WAVES 6     // the same for the benchmark-shape product kernel (n = 4096, 64-bit lanes, lazy): 6 -> <= 80 VGPRs, three
                                 //    512-thread workgroups per CU (3 x 51,204 B of LDS).  Measured +3 % over 4 (profiles/r2_h_*): the extra
                                 //    waves fill the issue slots the barriers and LDS round trips leave (10 % fewer cycles per launch) and the
                                 //    power cap gives two thirds of that back as clock (2.06 -> 1.92 GHz at 1.39 kW)
#endif

#ifndef TN_NTTF_PIN_LOGN
#define TN_NTTF_PIN_LOGN 99      // standalone transforms: from this log2 n on, loop invariants are pinned inside the row loop (ntt_fused_kernel).  Off:
                                 // pinning removes the 60-68 B/lane of scratch of the n = 8192 kernels and is SLOWER (0.408 vs 0.428, profiles/r3_n8192_ab.txt)
#endif
#ifndef TN_VEC_PREFETCH
#define TN_VEC_PREFETCH 1         // 1: twiddles of a vector-loaded phase (n = 8192) are requested one transpose ahead (28 more registers across it)
#endif
#ifndef TN_FUSED_CIN
#define TN_FUSED_CIN 0           // 1: also build the n = 4096 / 64-bit product kernel whose bound schedule assumes canonical inputs (see launch_fused_t)
#endif
#ifndef TN_DYNAMIC_ROWS
#define TN_DYNAMIC_ROWS 1        // 1: persistent workgroups take their next row from a device counter (atomicAdd) instead of a fixed
                                 //    stride: workgroups do not all run at the same speed, and with a fixed share the slowest sets the time
#endif
#ifndef TN_SCHED_CHUNK_BYTES
#define TN_SCHED_CHUNK_BYTES 32768   // dynamic scheduler: bytes of one operand handed out per atomicAdd (>= one row).  The launch's tail is up
                                     // to one chunk long: 32 / 64 / 128 KiB measured 2.138 / 2.147 / 2.160 ms at n = 4096 / 64-bit
#endif
#ifndef TN_SADDR
#define TN_SADDR 1               // 1: operand rows are addressed as scalar base (+ register offset, scalar unit) + 32-bit thread offset
#endif
#ifndef TN_KARG_ARITH
#define TN_KARG_ARITH 1          // 1: the product kernel reads its arithmetic constants and scalar twiddles per phase (kernarg_arith)
#endif
#ifndef TN_RESIDENT_TW
#define TN_RESIDENT_TW 1         // 1: the thread-private twiddles of the LAST forward stage (4 of the 7 records of the last phase; they do
                                 //    not depend on the row) stay in registers across rows of the persistent loop: 64 fewer bytes per
                                 //    thread and row from L2.  Only where the register budget is 128 (polymul_waves() <= 4): worth 0.8 %,
                                 //    against the 3 % of the third workgroup per CU that those 16 registers would cost
#endif
#ifndef TN_SHARE_MID_TW
#define TN_SHARE_MID_TW 1        // 1: ... and the phase before it (twiddles staged in LDS) likewise: a: ph 0-1, b: ph 0-1, a: ph 2, b: ph 2, b: ph 3, a: ph 3
#endif
#ifndef TN_SHARE_LAST_TW
#define TN_SHARE_LAST_TW 1       // 1: a and b run their last forward phase back to back on ONE fetch of its thread-private twiddles
#endif
#ifdef TN_MARKS
#define TN_MARK(n) asm volatile("; TNMARK " n)
#else
#define TN_MARK(n)
#endif
namespace tn {

// Row hand-out of the persistent fused kernels.  Dynamic (one atomicAdd on a device counter per chunk of rows) when the
// launch is long enough for every resident workgroup to take at least four chunks of TN_SCHED_CHUNK_BYTES worth of rows
// (1 row at n = 4096 / 64-bit, 8 at n = 1024 / 32-bit): a chunk that large keeps the one counter address from becoming
// the bottleneck (one row per atomic at n = 256 ran 13x slower than a fixed stride; at n = 1024 / 24-bit, batch 16,384, 10x).
// Otherwise a fixed stride of single rows.
struct RowPlan { u32 chunk; bool dynamic; };
static inline RowPlan plan_rows(size_t row_bytes, size_t batch, size_t resident) {
  size_t want = (size_t)TN_SCHED_CHUNK_BYTES / row_bytes;
  if (want < 1) want = 1;
  if (TN_DYNAMIC_ROWS && batch >= 4 * resident * want) return {(u32)want, true};
  return {1u, false};
}

// ============================================================================
// Fused kernel
// ============================================================================
// One LDS transpose between register layouts.  Data that crosses waves needs workgroup
// barriers on all three sides (the buffer is shared with the wave-private transposes before
// and after); a wave-local transpose only needs the compiler not to reorder it (the LDS
// operations of one wave execute in issue order and touch that wave's private region).
#ifndef TN_SHUFFLE_LAST
#define TN_SHUFFLE_LAST 0        // developer A/B (profiles/r3_shuffle_ab.txt): 1 = the LAST transpose of a transform (an 8 x 8 transpose between the
                                 // register index and the low three lane bits) through DPP lane permutes instead of LDS
#endif
// 8 x 8 transpose between registers and groups of 8 neighbouring lanes, without LDS: three butterfly stages; in stage b a lane
// and its partner (lane ^ 2^b) exchange, for every register pair (r, r | 2^b), the register the other one needs.  Per pair and
// 32-bit half: one select of what to send, the permute (quad_perm for distances 1 and 2; row_shl:4 / row_shr:4 under
// complementary bank masks for distance 4), two selects of what to keep.
template <int DIST> __device__ __forceinline__ u32 lane_xor(u32 v) {
  if constexpr (DIST == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);         // quad_perm [1,0,3,2]
  else if constexpr (DIST == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  else {
    int r = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);                                     // row_shl:4 -> banks 0, 2 (lane i <- lane i + 4)
    return (u32)__builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xf, 0xA, false);                                 // row_shr:4 -> banks 1, 3 (lane i <- lane i - 4)
  }
}
template <typename E>
__device__ __forceinline__ void transpose8_lanes(E (&x)[8], u32 lane) {
  static_for<0, 3>([&](auto b_) {
    constexpr int b = decltype(b_)::value, m = 1 << b;
    const bool hi = (lane >> b) & 1u;
    static_for<0, 8>([&](auto r_) {
      constexpr int r0 = decltype(r_)::value;
      if constexpr (!(r0 & m)) {
        constexpr int r1 = r0 | m;
        const E send = hi ? x[r0] : x[r1];
        E recv;
        if constexpr (sizeof(E) == 8) recv = ((u64)lane_xor<m>((u32)(send >> 32)) << 32) | lane_xor<m>((u32)send);
        else recv = lane_xor<m>((u32)send);
        x[r0] = hi ? recv : x[r0];
        x[r1] = hi ? x[r1] : recv;
      }
    });
  });
}

template <typename E, typename Cfg, int EX, int FROM, int TO>
__device__ __forceinline__ void exchange(E (&x)[Cfg::R], u32 tau, E* lds) {
  if constexpr (TN_SHUFFLE_LAST && Cfg::LPT == 3 && EX == Cfg::PHASES - 2 && Cfg::pos(EX + 1) == 0 && Cfg::pos(EX) == Cfg::LPT) {
    // register index <-> lane bits [0, 3): jidx(EX) = (tau >> 3) << 6 | r << 3 | (tau & 7),  jidx(EX + 1) = tau << 3 | r
    transpose8_lanes<E>(x, tau);
  } else if constexpr (Cfg::ex_wave_local(EX)) {
    __builtin_amdgcn_wave_barrier();
    ex_store<E, Cfg, EX, FROM>(x, tau, lds);
    __builtin_amdgcn_wave_barrier();     // lanes read what OTHER lanes of the wave wrote: loads may not move above the stores
    ex_load<E, Cfg, EX, TO>(x, tau, lds);
    __builtin_amdgcn_wave_barrier();
  } else {
#if TN_ABL_NO_BARRIER
    __builtin_amdgcn_wave_barrier();
    ex_store<E, Cfg, EX, FROM>(x, tau, lds);
    __builtin_amdgcn_wave_barrier();
    ex_load<E, Cfg, EX, TO>(x, tau, lds);
    __builtin_amdgcn_wave_barrier();
#else
    __syncthreads();
    ex_store<E, Cfg, EX, FROM>(x, tau, lds);
    __syncthreads();
    ex_load<E, Cfg, EX, TO>(x, tau, lds);
    __syncthreads();
#endif
  }
}

// The kernel's Arith argument as it lies in the kernel-argument segment (first argument of both fused kernels), addressed
// through an opaque zero: fields are then read by scalar loads AFTER the point where `zero` was defined, i.e. per phase,
// instead of living in SGPRs (or, spilled, in VGPR lanes) across the whole persistent row loop.
template <typename E>
__device__ __forceinline__ const Arith<E>& kernarg_arith(u32 zero) {
  typedef const __attribute__((address_space(4))) char* KP;
  return *(const Arith<E>*)((KP)__builtin_amdgcn_kernarg_segment_ptr() + zero);
}

// Forward transform, phases [P0, P1).  The thread-private twiddles of the last phase live in tw.pre[]; with `fetch_pre` they
// are requested from L2 just before the transpose that precedes that phase, so their latency hides behind it.
// KARG: take the arithmetic constants and the scalar twiddles of each phase through a fresh opaque zero (see kernarg_arith).
template <typename E, typename Cfg, typename Pol, int P0, int P1, bool KARG = false, int PRE_END = Cfg::LOGN>
__device__ __forceinline__ void forward_range(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw_in, const Arith<E>& ar_in, E* lds, bool fetch_pre,
                                              u32 tau_g) {      // tau_g: the thread index again, for global addressing (opaque_copy)
  typename TwOf<E>::type vec[Cfg::R];        // twiddles of a vector-loaded phase (n = 8192), requested one transpose ahead
  static_for<P0, P1>([&](auto p_) {
    constexpr int p = decltype(p_)::value;
    constexpr bool FULL = Cfg::stage_end(p) - Cfg::stage_begin(p) == Cfg::LPT;
    TN_MARK("fwd_phase");
    TwRefs<E> tw = tw_in;
    if constexpr (KARG) tw.zero = opaque_zero();
    if constexpr (TN_VEC_PREFETCH && Cfg::tw_src(p) == Cfg::TW_VEC && FULL && p > P0) tw.mid = vec;
    const Arith<E>& ar = KARG ? kernarg_arith<E>(tw.zero) : ar_in;
    fwd_phase<E, Cfg, Pol, p>(x, tau, tw, ar);
    TN_MARK("fwd_other");
    if constexpr (p == Cfg::PHASES - 2) {
      if (fetch_pre) {
        sched_fence();                 // request the last phase's private twiddles; they fly during the transpose
        tw_prefetch_stages<E, Cfg, Cfg::stage_begin(Cfg::PHASES - 1), PRE_END>(tw.pre, tau_g, tw.glob);   // (stages >= PRE_END: resident)
        sched_fence();
      }
    }
    if constexpr (p + 1 < P1) {
      constexpr int pn = p + 1 < Cfg::PHASES ? p + 1 : p;
      if constexpr (TN_VEC_PREFETCH && Cfg::tw_src(pn) == Cfg::TW_VEC && Cfg::stage_end(pn) - Cfg::stage_begin(pn) == Cfg::LPT) {
        sched_fence();
        tw_fetch_vec<E, Cfg, pn>(vec, tau_g, tw.glob);
        sched_fence();
      }
    }
    if constexpr (p + 1 < Cfg::PHASES) exchange<E, Cfg, p, p, p + 1>(x, tau, lds);
  });
}
template <typename E, typename Cfg, typename Pol>
__device__ __forceinline__ void forward_all(E (&x)[Cfg::R], u32 tau, const typename TwOf<E>::type* __restrict__ glob,
                                            const typename TwOf<E>::type* lds_tw, const Arith<E>& ar, E* lds, u32 zero = 0) {
  typename TwOf<E>::type pre[Cfg::NPRE];
  const TwRefs<E> tw = {glob, lds_tw, pre, nullptr, zero};
  forward_range<E, Cfg, Pol, 0, Cfg::PHASES>(x, tau, tw, ar, lds, true, tau);
}

// Inverse transform.  `after_first` runs once the first phase (the one whose thread-private
// twiddles come from L2 through vector loads) has been computed: vector-memory operations
// complete in order, so the long-latency HBM prefetch of the next row must be issued AFTER
// those twiddle loads have been consumed, or every wave would wait for HBM at the top of the inverse.
template <typename E, typename Cfg, typename Pol, bool KARG = false, typename F>
__device__ __forceinline__ void inverse_all(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw_in, const Arith<E>& ar_in, E* lds,
                                            F&& after_first) {
  typename TwOf<E>::type vec[Cfg::R];        // twiddles of a vector-loaded phase (n = 8192), requested one transpose ahead
  static_for<0, Cfg::PHASES>([&](auto i_) {
    constexpr int p = Cfg::PHASES - 1 - decltype(i_)::value;
    constexpr bool FULL = Cfg::stage_end(p) - Cfg::stage_begin(p) == Cfg::LPT;
    TN_MARK("inv_phase");
    TwRefs<E> tw = tw_in;
    if constexpr (KARG) tw.zero = opaque_zero();
    if constexpr (TN_VEC_PREFETCH && Cfg::tw_src(p) == Cfg::TW_VEC && FULL && p < Cfg::PHASES - 1) tw.mid = vec;
    const Arith<E>& ar = KARG ? kernarg_arith<E>(tw.zero) : ar_in;
    inv_phase<E, Cfg, Pol, p>(x, tau, tw, ar);
    TN_MARK("inv_other");
    // (the next phase's vector-loaded twiddles are requested BEFORE the next row's HBM prefetch: vector-memory operations return
    //  in order, and behind the prefetch the phase would wait for HBM: n = 8192 cg_intt, profiles/r3_n8192_ab.txt)
    if constexpr (p > 0) {
      constexpr int pn = p > 0 ? p - 1 : 0;
      if constexpr (TN_VEC_PREFETCH && Cfg::tw_src(pn) == Cfg::TW_VEC && Cfg::stage_end(pn) - Cfg::stage_begin(pn) == Cfg::LPT) {
        sched_fence();
        tw_fetch_vec<E, Cfg, pn>(vec, tau, tw.glob);
        sched_fence();
      }
    }
    if constexpr (p == Cfg::PHASES - 1) { sched_fence(); after_first(); sched_fence(); }
    if constexpr (p > 0) exchange<E, Cfg, p - 1, p, p - 1>(x, tau, lds);
  });
}

template <typename E> struct alignas(2 * sizeof(E)) PairOf { E lo, hi; };

#ifdef TN_FUSED_STAMPS
// DIAGNOSTIC BUILD ONLY (tools/gpu_fused_clock.py; MI355X_MICROARCH.md, DVFS give-back item 6): every workgroup of the product
// kernel stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) once before and once after its row loop into an array
// of its own in the code object; nothing in the kernel reads it and no output depends on it.  The shipped library has no stamp.
__device__ unsigned long long tn_fused_stamps[4 * 4096];
#endif

template <typename E, typename Cfg>
__device__ __forceinline__ E ld_operand(const E* __restrict__ p, u32 row, u32 tau, int r) {
#if TN_ABL_NO_GLOBAL
  return (E)(tau * 2654435761u + 7 * r + row);
#else
#if TN_ABL_ROWMASK
  row &= TN_ABL_ROWMASK;
#endif
#if TN_NT_STREAM
  // address = (row base + the register's offset: wave-uniform, scalar unit) + the thread's offset (ONE vector register for all
  // R accesses); written out so the compiler does not keep one vector offset per 8 KiB of row (loop invariants it then spills)
  const TN_GLOBAL_AS E* rp = uniform_ptr(p + ((size_t)row << Cfg::LOGN) + Cfg::jidx(0, 0, r));
  return __builtin_nontemporal_load(rp + (Cfg::jidx(0, tau, 0) & (u32)(Cfg::N - 1)));   // streamed once: keep L2 for the twiddle tables
#else
  return p[((size_t)row << Cfg::LOGN) + Cfg::jidx(0, tau, r)];
#endif
#endif
}

template <typename E, typename Cfg>
__device__ __forceinline__ void st_result(E* __restrict__ c, u32 row, u32 tau, const E (&x)[Cfg::R]) {
#if TN_ABL_ROWMASK
  row &= TN_ABL_ROWMASK;
#endif
  const size_t off = (size_t)row << Cfg::LOGN;
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r) {
#if TN_NT_STREAM
    __builtin_nontemporal_store(x[r], uniform_ptr(c + off + Cfg::jidx(0, 0, r)) + (Cfg::jidx(0, tau, 0) & (u32)(Cfg::N - 1)));
#else
    c[off + Cfg::jidx(0, tau, r)] = x[r];
#endif
  }
}

// waves per SIMD the product kernel's register allocation leaves room for (16 coeff/thread shapes need > 128 VGPRs)
template <typename E, int LOGN, int LPT, bool LAZY>
constexpr int polymul_waves() {
  return LPT >= 4 ? 2 : (sizeof(E) == 8 && LOGN == 12 && LAZY) ? TN_POLYMUL60_WAVES : TN_FUSED_MIN_WAVES;
}

template <typename E, int LOGN, int LPT, bool LAZY, bool CIN = false>
__global__ void __launch_bounds__((1 << (LOGN - LPT)), (polymul_waves<E, LOGN, LPT, LAZY>()))
polymul_fused_kernel(const Arith<E> ar, const typename TwOf<E>::type* __restrict__ tab_fwd,
                     const typename TwOf<E>::type* __restrict__ tab_inv, const E* __restrict__ a, const E* __restrict__ b,
                     E* __restrict__ c, u32 batch, u32* sched, u32 chunk) {
  // The twiddle tables are separate __restrict__ kernel arguments (not fields of a struct) so the
  // compiler can prove the stores to c never alias them: wave-uniform twiddle loads then become
  // scalar loads (s_load_dwordx4) instead of vector loads that every wave would wait on.
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  typedef Policy<E, LAZY, CIN> Pol;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  E* lds = reinterpret_cast<E*>(tn_smem);
  typedef typename TwOf<E>::type Tw;
  const u32 tau = threadIdx.x;
  // twiddles of the lane-dependent middle phases: staged once per (persistent) workgroup in LDS
  Tw* lds_fwd = reinterpret_cast<Tw*>(lds + Cfg::lds_elems());
  Tw* lds_inv = lds_fwd + Cfg::lds_tw_count();
  u32* lds_next = reinterpret_cast<u32*>(lds_inv + Cfg::lds_tw_count());      // row index this workgroup takes next
  for (u32 i = tau; i < (u32)Cfg::lds_tw_count(); i += Cfg::THREADS) {
    lds_fwd[i] = tab_fwd[Cfg::lds_tw_lo() + i];
    lds_inv[i] = tab_inv[Cfg::lds_tw_lo() + i];
  }
  __syncthreads();
  // Persistent workgroup.  Rows come in chunks of `chunk` consecutive rows: chunk blockIdx.x first, then whatever the
  // device-wide counter sched[0] hands out (atomicAdd): a workgroup on a slower CU / XCD simply takes fewer chunks, so
  // the launch ends when the work does, not when the slowest fixed share does.  (chunk > 1 for short rows keeps the
  // rate of atomics on that one address low.)  sched == nullptr: fixed stride.  The next row's first
  // operand is fetched from HBM into the registers that held b (dead after the pointwise
  // product) while the inverse transform of the current row runs; b itself is requested at the
  // top of the row and not needed until a's forward transform is done.
  E xa[Cfg::R], xb[Cfg::R];
#ifdef TN_FUSED_STAMPS
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  u32 row = blockIdx.x * chunk;
  u32 taken = 1;                            // rows taken from the current chunk           (both workgroup-uniform: scalar registers)
  u32 chunk_id = blockIdx.x;                // fixed-stride mode: the chunk being processed
  if (row < batch) {
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) xb[r] = ld_operand<E, Cfg>(a, row, tau, r);
  }
  // The stores of row k are issued at the TOP of iteration k+1 (software-pipelined): vector-memory
  // operations retire in order and the compiler drains them all at the loop back-edge, so stores
  // issued at the bottom would expose their full latency there on every row.  Issued at the top,
  // the only operations in flight at the back-edge are the prefetch loads, which are needed anyway.
  // (first iteration: nothing to store yet -> the zero-initialised registers are written to this
  //  row's own slot, which the same thread overwrites with the real result one iteration later;
  //  keeping the store unconditional keeps the loop top branch-free so the ordering below holds)
  // last forward phase's thread-private twiddles, shared by a and b; those of the last stage are loaded once per workgroup
  constexpr int PRE_END = (TN_RESIDENT_TW && polymul_waves<E, LOGN, LPT, LAZY>() <= 4 && Cfg::stage_end(Cfg::PHASES - 1) - Cfg::stage_begin(Cfg::PHASES - 1) >= 2) ? Cfg::LOGN - 1 : Cfg::LOGN;
  Tw prf[Cfg::NPRE];
  tw_prefetch_stages<E, Cfg, PRE_END, Cfg::LOGN>(prf, tau, tab_fwd);
  u32 prev = row;
  bool have_c = false;
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r) xa[r] = 0;
  while (row < batch) {
    // The workgroup-uniform twiddles of the first phase (13 records = 52 SGPRs) are reloaded by the scalar unit in every
    // row instead of being hoisted out of the loop: hoisted they do not fit the SGPR file and come back through
    // v_readlane (a vector-ALU slot each), while a scalar load that hits the scalar cache is free.
    const u32 zero = opaque_zero();
    const u32 tl = opaque_copy(tau);         // thread index for global addressing within this row (see opaque_copy)
    // one thread determines the next row now; everyone reads the answer after a's transform (barriers in between)
    const bool in_chunk = taken != chunk;                  // workgroup-uniform bookkeeping, kept on the scalar unit (counting up:
    taken = in_chunk ? taken + 1 : 1;                      // a down-counter's "subtract and test the borrow" is selected as a vector op)
    chunk_id = in_chunk ? chunk_id : chunk_id + gridDim.x;
    if (tau == 0) *lds_next = in_chunk ? row + 1 : (sched ? gridDim.x + atomicAdd(&sched[0], 1u) : chunk_id) * chunk;
    // consume this row's a (prefetched during the previous inverse) FIRST: at this point only those
    // loads are in flight, so the wait is exact; only then issue the stores of the previous row and b's loads
    TN_MARK("loop_top");
    E xn[Cfg::R];
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) xn[r] = xb[r];
    load_reduce<E, Cfg, Pol>(xn, ar);
    sched_fence();
#if TN_STORE_AT_TOP
    st_result<E, Cfg>(c, prev, tl, xa);
#endif
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) xb[r] = ld_operand<E, Cfg>(b, row, tl, r);
    sched_fence();
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) xa[r] = xn[r];
    constexpr bool KARG = TN_KARG_ARITH != 0;
    constexpr bool SHARE = TN_SHARE_LAST_TW && Cfg::PHASES >= 2;
    // SHARE2: the phase before the last is a full LDS-sourced phase: its 2^LPT - 1 twiddles are read into registers once
    constexpr int PM = Cfg::PHASES >= 3 ? Cfg::PHASES - 2 : 0;
    constexpr bool SHARE2 = SHARE && TN_SHARE_MID_TW && Cfg::PHASES >= 3 && Cfg::tw_src(PM) == Cfg::TW_LDS &&
                            Cfg::stage_end(PM) - Cfg::stage_begin(PM) == Cfg::LPT;
    const TwRefs<E> twf = {tab_fwd, lds_fwd, prf, nullptr, zero};
    // A^ stays in registers while b is transformed
    if constexpr (SHARE2) forward_range<E, Cfg, Pol, 0, PM, KARG>(xa, tau, twf, ar, lds, false, tl);
    else if constexpr (SHARE) forward_range<E, Cfg, Pol, 0, Cfg::PHASES - 1, KARG>(xa, tau, twf, ar, lds, false, tl);
    else forward_all<E, Cfg, Pol>(xa, tau, tab_fwd, lds_fwd, ar, lds, zero);
    __syncthreads();
    const u32 next = wave_uniform(*lds_next);
    load_reduce<E, Cfg, Pol>(xb, ar);
    if constexpr (SHARE2) {
      forward_range<E, Cfg, Pol, 0, PM, KARG>(xb, tau, twf, ar, lds, false, tl);
      Tw mid[Cfg::R];
      tw_fetch_mid<E, Cfg, PM>(mid, tau, lds_fwd);
      const TwRefs<E> twm = {tab_fwd, lds_fwd, prf, mid, zero};
      forward_range<E, Cfg, Pol, PM, PM + 1, KARG>(xa, tau, twm, ar, lds, false, tl);
      forward_range<E, Cfg, Pol, PM, PM + 1, KARG, PRE_END>(xb, tau, twm, ar, lds, true, tl);
      forward_range<E, Cfg, Pol, PM + 1, Cfg::PHASES, KARG>(xb, tau, twf, ar, lds, false, tl);
      forward_range<E, Cfg, Pol, PM + 1, Cfg::PHASES, KARG>(xa, tau, twf, ar, lds, false, tl);
    } else if constexpr (SHARE) {
      forward_range<E, Cfg, Pol, 0, Cfg::PHASES - 1, KARG, PRE_END>(xb, tau, twf, ar, lds, true, tl);
      forward_range<E, Cfg, Pol, Cfg::PHASES - 1, Cfg::PHASES, KARG>(xb, tau, twf, ar, lds, false, tl);
      forward_range<E, Cfg, Pol, Cfg::PHASES - 1, Cfg::PHASES, KARG>(xa, tau, twf, ar, lds, false, tl);
    } else {
      forward_all<E, Cfg, Pol>(xb, tau, tab_fwd, lds_fwd, ar, lds, zero);
    }
    // the inverse starts with the thread-private phase: request its twiddles before the product
    Tw pre[Cfg::NPRE];
    tw_prefetch<E, Cfg>(pre, tl, tab_inv);
    TN_MARK("pointwise");
    pointwise<E, Cfg, Pol>(xa, xb, ar);
    TN_MARK("after_pointwise");
    const TwRefs<E> twi = {tab_inv, lds_inv, pre, nullptr, zero};
    inverse_all<E, Cfg, Pol, KARG>(xa, tau, twi, ar, lds, [&]() {
      // next row's first operand -> the registers that held b.  Unconditional (after the last row this row's a is read
      // again and dropped): a branch here costs a register copy of all R values on the path that skips it.
      const u32 nrow = next < batch ? next : row;
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) xb[r] = ld_operand<E, Cfg>(a, nrow, tl, r);
    });
#if TN_STORE_AT_TOP
    prev = row;
    have_c = true;
#else
    st_result<E, Cfg>(c, row, tau, xa);
#endif
    row = next;
  }
  if (have_c) st_result<E, Cfg>(c, prev, tau, xa);
#ifdef TN_FUSED_STAMPS
  if (tau == 0 && blockIdx.x < 4096) {
    tn_fused_stamps[4 * blockIdx.x + 0] = st_t0; tn_fused_stamps[4 * blockIdx.x + 1] = st_r0;
    tn_fused_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime(); tn_fused_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // the last workgroup to run out of rows re-arms the counters for the next launch that uses this slot
  if (sched && tau == 0 && atomicAdd(&sched[1], 1u) == gridDim.x - 1) { sched[0] = 0; sched[1] = 0; }
}

// Standalone transforms on the register-tiled machinery, natural order in and out:
//   FNTT_TWIST_FWD   X[k] = sum_i x[i] psi^(i(2k+1))   = twist + cg_ntt = forward_ntt_bench (benchmark_ntt_60bit.cpp:161-165)
//   FNTT_CYCLIC_FWD  cg_ntt(x, omega=psi^2)            (cg_ntt.py:29-65)
//   FNTT_CYCLIC_INV  cg_intt(X, omega=psi^2)           (cg_ntt.py:68-75)
// All three are the same merged Cooley-Tukey / Gentleman-Sande butterflies; only the twiddle table differs: the
// negacyclic one (psi_brv: factorisation tree of x^n + 1) or the cyclic one (cyc_brv / cyc_inv_brv: tree of x^n - 1,
// see HostTables).  The merged transform produces / consumes bit-reversed order in the last phase's register layout;
// one extra LDS transpose through a natural-order image turns that into unit-stride HBM accesses.
template <typename E, int LOGN, int LPT, bool LAZY, int MODE>
__global__ void __launch_bounds__((1 << (LOGN - LPT)), (LPT >= 4 ? 2 : TN_FUSED_MIN_WAVES))
ntt_fused_kernel(const Arith<E> ar, const typename TwOf<E>::type* __restrict__ tab, const E* __restrict__ in, E* __restrict__ out, u32 batch,
                 u32* sched, u32 chunk) {
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  typedef Policy<E, LAZY> Pol;
  typedef typename TwOf<E>::type Tw;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  E* lds = reinterpret_cast<E*>(tn_smem);
  const u32 tau = threadIdx.x;
  // LDS: [transpose image][natural-order image][staged twiddles][2 next-row slots].  The natural-order image has its own
  // region so that the only workgroup barrier a row adds to those of the cross-wave transpose is the one between writing
  // and reading that image; the next-row slot is double-buffered for the same reason.
  E* nat = lds + Cfg::lds_elems();
  Tw* lds_tab = reinterpret_cast<Tw*>(nat + Cfg::N);
  u32* lds_next = reinterpret_cast<u32*>(lds_tab + Cfg::lds_tw_count());
  for (u32 i = tau; i < (u32)Cfg::lds_tw_count(); i += Cfg::THREADS) lds_tab[i] = tab[Cfg::lds_tw_lo() + i];
  // rows: chunk blockIdx.x, then whatever the device-wide counter hands out (see polymul_fused_kernel); the prefetch needs
  // the next row at the top of an iteration, so the counter is asked one iteration ahead
  u32 left = chunk - 1, chunk_id = blockIdx.x;            // thread 0's copies are the ones used
  auto take_next = [&](u32 cur, u32 slot) {               // thread 0 only
    if (left) { --left; lds_next[slot] = cur + 1; }
    else {
      chunk_id = sched ? gridDim.x + atomicAdd(&sched[0], 1u) : chunk_id + gridDim.x;
      left = chunk - 1;
      lds_next[slot] = chunk_id * chunk;
    }
  };
  if (tau == 0) take_next(blockIdx.x * chunk, 1u);
  __syncthreads();
  u32 next = wave_uniform(lds_next[1]);
  constexpr int LAST = Cfg::PHASES - 1;
  // The next row's input is requested as soon as the current one has been consumed, so the HBM latency of row k+1
  // hides behind the arithmetic of row k (persistent workgroup, like the product kernel).
  E xn[Cfg::R];
  u32 row = blockIdx.x * chunk;
  if (row < batch) {
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) xn[r] = ld_operand<E, Cfg>(in, row, tau, r);
  }
  for (u32 it = 0; row < batch; ++it) {
    if (tau == 0) take_next(next, it & 1u);              // read back after this row's barrier(s)
    // (as in the product kernel: an opaque zero / thread index per row keep the uniform twiddle loads, the arithmetic constants
    //  and the operand addresses inside the row loop instead of in registers across it — at n = 8192 they spilled: 60-68 B/lane)
    constexpr bool PIN = LOGN >= TN_NTTF_PIN_LOGN;
    const u32 zero = PIN ? opaque_zero() : 0u;
    const u32 tl = PIN ? opaque_copy(tau) : tau;
    constexpr bool KARG = PIN && TN_KARG_ARITH != 0;
    E x[Cfg::R];
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) x[r] = xn[r];
    sched_fence();
    if (MODE != FNTT_CYCLIC_INV && next < batch) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) xn[r] = ld_operand<E, Cfg>(in, next, tl, r);
    }
    sched_fence();
    if (MODE == FNTT_CYCLIC_INV) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = Pol::load(x[r], ar);
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) nat[Cfg::nat_addr(Cfg::jidx(0, tau, r))] = x[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = nat[Cfg::nat_addr(bitrev(Cfg::jidx(LAST, tau, r), LOGN))];
      Tw pre[Cfg::NPRE];
      tw_prefetch<E, Cfg>(pre, tl, tab);
      const TwRefs<E> tw = {tab, lds_tab, pre, nullptr, zero};
      // the inverse starts with the thread-private phase (28 registers of twiddles): the prefetch goes after it,
      // as in the product kernel
      inverse_all<E, Cfg, Pol, KARG>(x, tau, tw, ar, lds, [&]() {
        if (next < batch) {
#pragma unroll
          for (int r = 0; r < Cfg::R; ++r) xn[r] = ld_operand<E, Cfg>(in, next, tl, r);
        }
      });
      st_result<E, Cfg>(out, row, tl, x);
    } else {
#pragma unroll
      for (int r = 0; r < Cfg::R / 2; ++r) x[r] = Pol::load(x[r], ar);                 // the other half is multiplied first
      if (!LAZY) {
#pragma unroll
        for (int r = Cfg::R / 2; r < Cfg::R; ++r) x[r] = Pol::load(x[r], ar);
      }
      {
        Tw pre[Cfg::NPRE];
        const TwRefs<E> tw = {tab, lds_tab, pre, nullptr, zero};
        forward_range<E, Cfg, Pol, 0, Cfg::PHASES, KARG>(x, tau, tw, ar, lds, true, tl);
      }
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r)
        nat[Cfg::nat_addr(bitrev(Cfg::jidx(LAST, tau, r), LOGN))] = LAZY ? Pol::canon(x[r], ar) : x[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = nat[Cfg::nat_addr(Cfg::jidx(0, tau, r))];
      st_result<E, Cfg>(out, row, tl, x);
    }
    if (Cfg::THREADS <= 64) __syncthreads();             // single-wave workgroups have no barrier inside the transposes
    row = next;
    next = wave_uniform(lds_next[it & 1u]);
  }
  if (sched && tau == 0 && atomicAdd(&sched[1], 1u) == gridDim.x - 1) { sched[0] = 0; sched[1] = 0; }
}

template <typename E, int LOGN, int LPT, bool LAZY>
static hipError_t launch_nttf_t(const tn_plan* p, int mode, const void* in, void* out, size_t batch, hipStream_t s) {
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  typedef typename TwOf<E>::type Tw;
  const size_t lds_bytes = (size_t)(Cfg::lds_elems() + Cfg::N) * sizeof(E) + (size_t)Cfg::lds_tw_count() * sizeof(Tw) + 16;   // + natural image, next-row slots
  const PlanView<E> pv = make_view<E>(p);
  const void* kern = nullptr;
  const Tw* tab = nullptr;
  Arith<E> ar = pv.ar;
  if (mode == FNTT_TWIST_FWD) { kern = (const void*)ntt_fused_kernel<E, LOGN, LPT, LAZY, FNTT_TWIST_FWD>; tab = pv.psi_brv; }
  else if (mode == FNTT_CYCLIC_FWD) { kern = (const void*)ntt_fused_kernel<E, LOGN, LPT, LAZY, FNTT_CYCLIC_FWD>; tab = pv.cyc_brv; }
  else { kern = (const void*)ntt_fused_kernel<E, LOGN, LPT, LAZY, FNTT_CYCLIC_INV>; tab = pv.cyc_inv_brv; ar.fninv_w1 = ar.fninv; }   // cyc_inv_brv[1] = 1
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  int per_cu = 0;
  hipError_t qe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, Cfg::THREADS, lds_bytes);
  if (qe != hipSuccess || per_cu < 1) per_cu = 1;
  const size_t resident = (size_t)per_cu * (size_t)p->num_cus;
  const RowPlan rp = plan_rows(Cfg::N * sizeof(E), batch, resident);
  u32 chunk = rp.chunk;
  const size_t chunks = (batch + chunk - 1) / chunk;
  const u32 grid = (u32)(chunks < resident ? chunks : resident);
  const E* in_ = (const E*)in; E* out_ = (E*)out; u32 b32 = (u32)batch;
  SchedSlot slot;
  if (rp.dynamic) slot = sched_acquire(p, s);
  u32* sched = slot.ptr;
  void* args[] = {&ar, &tab, &in_, &out_, &b32, &sched, &chunk};
  const hipError_t le = hipLaunchKernel(kern, dim3(grid), dim3(Cfg::THREADS), args, lds_bytes, s);
  sched_release(p, slot, s, le == hipSuccess);
  return le;
}

template <typename E, bool LAZY>
static hipError_t launch_nttf_e(const tn_plan* p, int mode, const void* in, void* out, size_t batch, hipStream_t s) {
#ifdef TN_ONLY_MAIN
  return hipErrorInvalidValue;
#else
  switch (p->logn) {
    case 8: return launch_nttf_t<E, 8, fused_lpt(8), LAZY>(p, mode, in, out, batch, s);
    case 9: return launch_nttf_t<E, 9, fused_lpt(9), LAZY>(p, mode, in, out, batch, s);
    case 10: return launch_nttf_t<E, 10, fused_lpt(10), LAZY>(p, mode, in, out, batch, s);
    case 11: return launch_nttf_t<E, 11, fused_lpt(11), LAZY>(p, mode, in, out, batch, s);
    case 12: return launch_nttf_t<E, 12, fused_lpt(12), LAZY>(p, mode, in, out, batch, s);
    case 13: return launch_nttf_t<E, 13, fused_lpt(13), LAZY>(p, mode, in, out, batch, s);
    default: return hipErrorInvalidValue;
  }
#endif
}

hipError_t launch_ntt_fused(const tn_plan* p, int mode, const void* in, void* out, size_t batch, hipStream_t s) {
  if (batch == 0) return hipSuccess;
  if (p->elem_bytes == 8)
    return p->lazy ? launch_nttf_e<u64, true>(p, mode, in, out, batch, s) : launch_nttf_e<u64, false>(p, mode, in, out, batch, s);
  return p->lazy ? launch_nttf_e<u32, true>(p, mode, in, out, batch, s) : launch_nttf_e<u32, false>(p, mode, in, out, batch, s);
}

bool fused_supported(u32 logn, int) { return fused_lpt((int)logn) != 0; }

template <typename E, int LOGN, int LPT, bool LAZY>
static hipError_t launch_fused_t(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s, bool cyclic) {
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  const size_t lds_bytes = (size_t)Cfg::lds_elems() * sizeof(E) +
                           (size_t)2 * Cfg::lds_tw_count() * sizeof(typename TwOf<E>::type) + 16;      // + the next-row slot
  // promised-canonical inputs (TN_PLAN_CANONICAL_INPUTS): the n = 4096 / 64-bit lazy kernel has a second instantiation whose bound
  // schedule starts from q (no load folds); every other shape ignores the promise
  // Built only with -DTN_FUSED_CIN=1 (developer A/B, tools/gpu_ledger.py): measured 2.7 % SLOWER than the any-word kernel on the same
  // box (same 1,963 vector instructions per wave and row, 36 instead of 8 bytes of scratch: profiles/r3_headline_ledger.txt), so the
  // shipped library accepts the promise and runs the any-word kernel.
  constexpr bool HAS_CIN = TN_FUSED_CIN && sizeof(E) == 8 && LOGN == 12 && LAZY;
  auto kern = polymul_fused_kernel<E, LOGN, LPT, LAZY>;
  if constexpr (HAS_CIN) { if (p->canonical_inputs && !cyclic) kern = polymul_fused_kernel<E, LOGN, LPT, LAZY, true>; }
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  // persistent grid: as many workgroups as can be resident (occupancy query), each looping over rows
  int per_cu = 0;
  hipError_t qe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, Cfg::THREADS, lds_bytes);
  if (qe != hipSuccess || per_cu < 1) per_cu = 1;
  const size_t resident = (size_t)per_cu * (size_t)p->num_cus;
  const RowPlan rp = plan_rows(Cfg::N * sizeof(E), batch, resident);
  const u32 chunk = rp.chunk;
  const size_t chunks = (batch + chunk - 1) / chunk;
  const u32 grid = (u32)(chunks < resident ? chunks : resident);
  const PlanView<E> pv = make_view<E>(p);
  // cyclic = product in Z_q[x]/(x^n - 1) (python_poly_mult, test_ntt_poly_mult.py:38-43): same kernel, twiddle
  // tables of the x^n - 1 factorisation tree (HostTables::cyc_brv), whose inverse table has entry 1 equal to 1
  Arith<E> ar = pv.ar;
  if (cyclic) ar.fninv_w1 = ar.fninv;
  // one counter pair per launch in flight (ring; each pair is re-armed by the kernel that used it)
  SchedSlot slot;
  if (rp.dynamic) slot = sched_acquire(p, s);
  u32* sched = slot.ptr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), lds_bytes, s, ar, cyclic ? pv.cyc_brv : pv.psi_brv,
                     cyclic ? pv.cyc_inv_brv : pv.psi_inv_brv, (const E*)a, (const E*)b, (E*)c, (u32)batch, sched, chunk);
  const hipError_t le = hipGetLastError();
  sched_release(p, slot, s, le == hipSuccess);
  return le;
}

template <typename E, bool LAZY>
static hipError_t launch_fused_e(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s, bool cyclic) {
#ifdef TN_ONLY_MAIN      // developer builds (tools/build_variant.sh -DTN_ONLY_MAIN): only the n = 4096 / 64-bit lazy product kernel
  if constexpr (!(sizeof(E) == 8 && LAZY)) return hipErrorInvalidValue;
  else return p->logn == 12 ? launch_fused_t<E, 12, fused_lpt(12), LAZY>(p, a, b, c, batch, s, cyclic) : hipErrorInvalidValue;
#else
  switch (p->logn) {
    case 8: return launch_fused_t<E, 8, fused_lpt(8), LAZY>(p, a, b, c, batch, s, cyclic);
    case 9: return launch_fused_t<E, 9, fused_lpt(9), LAZY>(p, a, b, c, batch, s, cyclic);
    case 10: return launch_fused_t<E, 10, fused_lpt(10), LAZY>(p, a, b, c, batch, s, cyclic);
    case 11: return launch_fused_t<E, 11, fused_lpt(11), LAZY>(p, a, b, c, batch, s, cyclic);
    case 12: return launch_fused_t<E, 12, fused_lpt(12), LAZY>(p, a, b, c, batch, s, cyclic);
    case 13: return launch_fused_t<E, 13, fused_lpt(13), LAZY>(p, a, b, c, batch, s, cyclic);
    default: return hipErrorInvalidValue;
  }
#endif
}

hipError_t launch_polymul_fused(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s, bool cyclic) {
  if (batch == 0) return hipSuccess;
  if (p->elem_bytes == 8)
    return p->lazy ? launch_fused_e<u64, true>(p, a, b, c, batch, s, cyclic) : launch_fused_e<u64, false>(p, a, b, c, batch, s, cyclic);
  return p->lazy ? launch_fused_e<u32, true>(p, a, b, c, batch, s, cyclic) : launch_fused_e<u32, false>(p, a, b, c, batch, s, cyclic);
}

const char* fused_kernel_name(const tn_plan* p) {
  (void)p;
  return "polymul_fused_kernel";
}

// ============================================================================
// Constant-geometry kernels: dispatch to the slices of cg_part.hip (cg_kernel_impl.h)
// ============================================================================
#define TN_CG_PART_DECL(k) hipError_t launch_cg_part##k(const tn_plan*, int, int, int, bool, const void*, const void*, void*, void*, size_t, hipStream_t)
TN_CG_PART_DECL(0); TN_CG_PART_DECL(1); TN_CG_PART_DECL(2); TN_CG_PART_DECL(3); TN_CG_PART_DECL(4); TN_CG_PART_DECL(5); TN_CG_PART_DECL(6);
#undef TN_CG_PART_DECL

// group: butterflies per lane-step (1, 2, 4, 8); layout: CgLayout.  The LDS layouts differ only where the sweep of BASELINE
// config 5 is defined (lazy 64-bit plans at n = 4096); every other plan runs the linear image whatever the variant says.
hipError_t launch_cg(const tn_plan* p, int mode, int group, int layout, const void* a, const void* b, void* out, void* trace,
                     size_t batch, hipStream_t s) {
  if (batch == 0) return hipSuccess;
#ifdef TN_ONLY_MAIN
  return hipErrorInvalidValue;
#else
  if (group != 1 && group != 2 && group != 4 && group != 8) return hipErrorInvalidValue;
  while ((2u * (u32)group) > p->n) group /= 2;                      // tiny n: a lane-step cannot hold more than the polynomial
  const bool big = p->n / (2u * (u32)group) > (group == 1 ? 2048u : 1024u);   // n = 8192 at GROUP 1, 2 (CgShape)
  if (p->elem_bytes == 4) return launch_cg_part0(p, mode, group, layout, big, a, b, out, trace, batch, s);
  if (!(p->lazy)) return launch_cg_part1(p, mode, group, layout, big, a, b, out, trace, batch, s);
  // lazy 64-bit plan: every table holds split constants.  Per-stage traces need canonical values at every stage.
  if (trace || !p->cg_lazy) return launch_cg_part2(p, mode, group, layout, big, a, b, out, trace, batch, s);
  if (p->logn != 12 || !p->cg_sched) return launch_cg_part3(p, mode, group, layout, big, a, b, out, trace, batch, s);
  if (group <= 2) return launch_cg_part4(p, mode, group, layout, big, a, b, out, trace, batch, s);
  if (group == 4) return launch_cg_part5(p, mode, group, layout, big, a, b, out, trace, batch, s);
  return launch_cg_part6(p, mode, group, layout, big, a, b, out, trace, batch, s);
#endif
}

const char* cg_kernel_name(const tn_plan*, int, int) { return "cg_kernel"; }

// ============================================================================
// Elementwise product and the O(n^2) direct product (on-device checker)
// ============================================================================
// pointwise_mul (benchmark_ntt_60bit.cpp:142-146; cg_ntt.py:88): c[i] = a[i] * b[i] mod q
template <typename E>
__global__ void pointwise_kernel(PlanView<E> pv, const E* __restrict__ a, const E* __restrict__ b, E* __restrict__ c, size_t total) {
  const Arith<E> ar = pv.ar;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    c[i] = mulmod_barrett(mul_tw(a[i], ar.one, ar.q), mul_tw(b[i], ar.one, ar.q), ar.q, ar.mu, ar.k);
}

// negacyclic_mul_reference (benchmark_ntt_60bit.cpp:167-180; test_cg_ntt.py:11-21; the
// benchmark_simple family): c[k] = sum_{i<=k} a[i] b[k-i] - sum_{i>k} a[i] b[n+k-i]  (mod q).
// One workgroup per (row, 256 output coefficients); operands staged in LDS.
template <typename E>
__global__ void __launch_bounds__(256)
schoolbook_kernel(PlanView<E> pv, const E* __restrict__ a, const E* __restrict__ b, E* __restrict__ c, u32 batch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  const u32 n = pv.n;
  const Arith<E> ar = pv.ar;
  E* sa = reinterpret_cast<E*>(tn_smem);
  E* sb = sa + n;
  const u32 chunks = (n + 255) / 256;
  for (u32 w = blockIdx.x; w < batch * chunks; w += gridDim.x) {
    const u32 row = w / chunks, k = (w % chunks) * 256 + threadIdx.x;
    const size_t off = (size_t)row * n;
    __syncthreads();
    for (u32 i = threadIdx.x; i < n; i += 256) {
      sa[i] = mul_tw(a[off + i], ar.one, ar.q);
      sb[i] = mul_tw(b[off + i], ar.one, ar.q);
    }
    __syncthreads();
    if (k < n) {
      E acc = 0;
      for (u32 i = 0; i < n; ++i) {
        const E t = mulmod_barrett(sa[i], sb[(k - i) & (n - 1)], ar.q, ar.mu, ar.k);
        if (i <= k) acc = csub((E)(acc + t), ar.q);
        else acc = acc >= t ? (E)(acc - t) : (E)(acc + (ar.q - t));
      }
      c[off + k] = acc;
    }
  }
}

template <typename E>
static hipError_t launch_aux_e(const tn_plan* p, int what, const void* a, const void* b, void* c, size_t batch, hipStream_t s) {
  if (what == 0) {
    const size_t total = batch * p->n;
    const u32 blocks = (u32)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pointwise_kernel<E>, dim3(blocks), dim3(256), 0, s, make_view<E>(p), (const E*)a, (const E*)b, (E*)c, total);
  } else {
    const size_t lds_bytes = (size_t)2 * p->n * sizeof(E);
    auto kern = schoolbook_kernel<E>;
    if (lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
    const size_t work = batch * ((p->n + 255) / 256);
    const u32 blocks = (u32)(work < 65536 ? work : 65536);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, s, make_view<E>(p), (const E*)a, (const E*)b, (E*)c, (u32)batch);
  }
  return hipGetLastError();
}

hipError_t launch_pointwise(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s) {
  if (batch == 0) return hipSuccess;
  return p->elem_bytes == 8 ? launch_aux_e<u64>(p, 0, a, b, c, batch, s) : launch_aux_e<u32>(p, 0, a, b, c, batch, s);
}
hipError_t launch_schoolbook(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s) {
  if (batch == 0) return hipSuccess;
  return p->elem_bytes == 8 ? launch_aux_e<u64>(p, 1, a, b, c, batch, s) : launch_aux_e<u32>(p, 1, a, b, c, batch, s);
}

// ============================================================================
// Synthetic inputs + digest (reference benchmark conventions)
// ============================================================================
static constexpr u64 LCG_A = 6364136223846793005ULL, LCG_C = 1442695040888963407ULL;

template <typename E>
__global__ void fill_lcg_kernel(E* __restrict__ dst, u32 n, u64 q, int narrow, u64 seed0, u64 stride, u32 batch) {
  const u32 CH = n < 16 ? n : 16;
  const u32 chunks = n / CH;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)batch * chunks) return;
  const u32 row = (u32)(gid / chunks), i0 = (u32)(gid % chunks) * CH;
  // x_i0 = f^i0(seed), f(x) = A x + C mod 2^64: square-and-multiply on affine maps
  u64 ra = 1, rc = 0, ba = LCG_A, bc = LCG_C;
  for (u32 e = i0; e; e >>= 1) {
    if (e & 1) { rc = ba * rc + bc; ra = ba * ra; }
    bc = ba * bc + bc; ba = ba * ba;
  }
  u64 x = ra * (seed0 + (u64)row * stride) + rc;
  for (u32 i = 0; i < CH; ++i) {
    x = LCG_A * x + LCG_C;                                         // benchmark_ntt_60bit.cpp:83
    dst[(size_t)row * n + i0 + i] = (E)(narrow ? (x >> 17) % q : x % q);   // :84 / benchmark_ntt.cpp:87
  }
}

template <typename E>
__global__ void checksum_kernel(const E* __restrict__ src, u64* __restrict__ out, u32 n, int narrow, u32 batch) {
  const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= batch) return;
  const u64 M = 0xffffffffffffffc5ULL;
  u64 acc = 0;
  for (u32 i = 0; i < n; ++i) {
    const u64 v = src[(size_t)row * n + i];
    if (narrow) {
      acc = (acc * 1315423911ULL + v) % M;                          // benchmark_ntt.cpp:228-233 (wraps mod 2^64 first)
    } else {
      // (acc*K + v) mod M with a 128-bit intermediate (benchmark_ntt_60bit.cpp:182-188); M = 2^64 - 59
      u64 lo = acc * 1315423911ULL, hi = mulhi64(acc, 1315423911ULL);
      const u64 lo2 = lo + v;
      hi += (lo2 < lo);
      u64 r = lo2 + hi * 59;                                        // 2^64 == 59 (mod M); hi < 2^31
      if (r < lo2) r += 59;
      acc = r >= M ? r - M : r;
    }
  }
  out[row] = acc;
}

hipError_t launch_fill_lcg(const tn_plan* p, void* dst, size_t batch, u64 seed0, u64 stride, hipStream_t s) {
  if (batch == 0) return hipSuccess;
  const u32 CH = p->n < 16 ? p->n : 16;
  const size_t total = batch * (p->n / CH);
  const u32 blocks = (u32)((total + 255) / 256);
  const int narrow = p->q < ((u64)1 << 32);
  if (p->elem_bytes == 8)
    hipLaunchKernelGGL(fill_lcg_kernel<u64>, dim3(blocks), dim3(256), 0, s, (u64*)dst, p->n, p->q, narrow, seed0, stride, (u32)batch);
  else
    hipLaunchKernelGGL(fill_lcg_kernel<u32>, dim3(blocks), dim3(256), 0, s, (u32*)dst, p->n, p->q, narrow, seed0, stride, (u32)batch);
  return hipGetLastError();
}

hipError_t launch_checksum(const tn_plan* p, const void* src, u64* out, size_t batch, hipStream_t s) {
  if (batch == 0) return hipSuccess;
  const u32 blocks = (u32)((batch + 63) / 64);
  const int narrow = p->q < ((u64)1 << 32);
  if (p->elem_bytes == 8)
    hipLaunchKernelGGL(checksum_kernel<u64>, dim3(blocks), dim3(64), 0, s, (const u64*)src, out, p->n, narrow, (u32)batch);
  else
    hipLaunchKernelGGL(checksum_kernel<u32>, dim3(blocks), dim3(64), 0, s, (const u32*)src, out, p->n, narrow, (u32)batch);
  return hipGetLastError();
}

}  // namespace tn

#ifdef TN_FUSED_STAMPS
extern "C" size_t tn_debug_fused_stamps(void* host, size_t max_bytes) {
  (void)hipDeviceSynchronize();
  const size_t nb = sizeof(tn::tn_fused_stamps) < max_bytes ? sizeof(tn::tn_fused_stamps) : max_bytes;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(tn::tn_fused_stamps), nb, 0, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return nb;
}
#endif
