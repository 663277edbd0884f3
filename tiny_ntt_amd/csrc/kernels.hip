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
//  cg_kernel            : K1-K5/K7, the reference's own constant-geometry
//      dataflow (cg_ntt.py:49-64) held in two LDS ping-pong images; canonical
//      arithmetic at every step when a trace is taken, so each stage's output
//      equals the reference's list `A`; GROUP = butterflies issued per lane-step
//      (8 = cg_ntt_8butterfly.py), LAYOUT = linear / padded / XOR-swizzled image
//      (the LDS-bank-conflict sweep of BASELINE config 5).
//  fill_lcg_kernel / checksum_kernel : the reference benchmark's input
//      generator and digest (benchmark_ntt_60bit.cpp:79-87,182-188), on device.
#include <hip/hip_runtime.h>
#include "plan.h"

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
template <typename E, typename Cfg, int EX, int FROM, int TO>
__device__ __forceinline__ void exchange(E (&x)[Cfg::R], u32 tau, E* lds) {
  if constexpr (Cfg::ex_wave_local(EX)) {
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
    if constexpr (Cfg::tw_src(p) == Cfg::TW_VEC && FULL && p > P0) tw.mid = vec;
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
      if constexpr (Cfg::tw_src(pn) == Cfg::TW_VEC && Cfg::stage_end(pn) - Cfg::stage_begin(pn) == Cfg::LPT) {
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
    if constexpr (Cfg::tw_src(p) == Cfg::TW_VEC && FULL && p < Cfg::PHASES - 1) tw.mid = vec;
    const Arith<E>& ar = KARG ? kernarg_arith<E>(tw.zero) : ar_in;
    inv_phase<E, Cfg, Pol, p>(x, tau, tw, ar);
    TN_MARK("inv_other");
    if constexpr (p == Cfg::PHASES - 1) { sched_fence(); after_first(); sched_fence(); }
    if constexpr (p > 0) {
      constexpr int pn = p > 0 ? p - 1 : 0;
      if constexpr (Cfg::tw_src(pn) == Cfg::TW_VEC && Cfg::stage_end(pn) - Cfg::stage_begin(pn) == Cfg::LPT) {
        sched_fence();
        tw_fetch_vec<E, Cfg, pn>(vec, tau, tw.glob);
        sched_fence();
      }
      exchange<E, Cfg, p - 1, p, p - 1>(x, tau, lds);
    }
  });
}

template <typename E> struct alignas(2 * sizeof(E)) PairOf { E lo, hi; };

// A global-memory pointer the compiler must keep in scalar registers (both halves through wave_uniform): the access it
// bases is then "scalar base + 32-bit thread offset" (global_load ... v_off, s[base]) and the base arithmetic stays on
// the scalar unit.  (The explicit address space keeps the access a global_* instruction after the integer round trip.)
#define TN_GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ TN_GLOBAL_AS T* uniform_ptr(T* p) {
#if TN_SADDR
  const unsigned long long v = (unsigned long long)p;
  return (TN_GLOBAL_AS T*)(((unsigned long long)wave_uniform((u32)(v >> 32)) << 32) | wave_uniform((u32)v));
#else
  return (TN_GLOBAL_AS T*)p;
#endif
}

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

template <typename E, int LOGN, int LPT, bool LAZY>
__global__ void __launch_bounds__((1 << (LOGN - LPT)), (polymul_waves<E, LOGN, LPT, LAZY>()))
polymul_fused_kernel(const Arith<E> ar, const typename TwOf<E>::type* __restrict__ tab_fwd,
                     const typename TwOf<E>::type* __restrict__ tab_inv, const E* __restrict__ a, const E* __restrict__ b,
                     E* __restrict__ c, u32 batch, u32* sched, u32 chunk) {
  // The twiddle tables are separate __restrict__ kernel arguments (not fields of a struct) so the
  // compiler can prove the stores to c never alias them: wave-uniform twiddle loads then become
  // scalar loads (s_load_dwordx4) instead of vector loads that every wave would wait on.
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  typedef Policy<E, LAZY> Pol;
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
    E x[Cfg::R];
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) x[r] = xn[r];
    sched_fence();
    if (MODE != FNTT_CYCLIC_INV && next < batch) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) xn[r] = ld_operand<E, Cfg>(in, next, tau, r);
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
      tw_prefetch<E, Cfg>(pre, tau, tab);
      const TwRefs<E> tw = {tab, lds_tab, pre};
      // the inverse starts with the thread-private phase (28 registers of twiddles): the prefetch goes after it,
      // as in the product kernel
      inverse_all<E, Cfg, Pol>(x, tau, tw, ar, lds, [&]() {
        if (next < batch) {
#pragma unroll
          for (int r = 0; r < Cfg::R; ++r) xn[r] = ld_operand<E, Cfg>(in, next, tau, r);
        }
      });
      st_result<E, Cfg>(out, row, tau, x);
    } else {
#pragma unroll
      for (int r = 0; r < Cfg::R / 2; ++r) x[r] = Pol::load(x[r], ar);                 // the other half is multiplied first
      if (!LAZY) {
#pragma unroll
        for (int r = Cfg::R / 2; r < Cfg::R; ++r) x[r] = Pol::load(x[r], ar);
      }
      forward_all<E, Cfg, Pol>(x, tau, tab, lds_tab, ar, lds);
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r)
        nat[Cfg::nat_addr(bitrev(Cfg::jidx(LAST, tau, r), LOGN))] = LAZY ? Pol::canon(x[r], ar) : x[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = nat[Cfg::nat_addr(Cfg::jidx(0, tau, r))];
      st_result<E, Cfg>(out, row, tau, x);
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
  if (rp.dynamic) slot = sched_acquire(p);
  u32* sched = slot.ptr;
  void* args[] = {&ar, &tab, &in_, &out_, &b32, &sched, &chunk};
  const hipError_t le = hipLaunchKernel(kern, dim3(grid), dim3(Cfg::THREADS), args, lds_bytes, s);
  sched_release(p, slot, s);
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
  auto kern = polymul_fused_kernel<E, LOGN, LPT, LAZY>;
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
  if (rp.dynamic) slot = sched_acquire(p);
  u32* sched = slot.ptr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), lds_bytes, s, ar, cyclic ? pv.cyc_brv : pv.psi_brv,
                     cyclic ? pv.cyc_inv_brv : pv.psi_inv_brv, (const E*)a, (const E*)b, (E*)c, (u32)batch, sched, chunk);
  const hipError_t le = hipGetLastError();
  sched_release(p, slot, s);
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
// Constant-geometry kernel (reference dataflow, canonical arithmetic)
// ============================================================================
// LDS image of one polynomial.  LAYOUT:
//   CG_LINEAR   element x at x.
//   CG_PADDED   16 bytes of padding after every 2*GROUP elements (one lane-step's contiguous read).
//   CG_SWIZZLED x ^ (((x >> 4) & 15) << 1) ^ (((x >> 9) & 7) << 1): pairs (2i, 2i+1) stay adjacent and 16-byte aligned (bit 0
//               untouched), and every access pattern of the sweep is bank-conflict free in the gfx950 banking model of
//               tests/test_lds_banks.py (found by exhaustive search, tools/cg_layout_search.py): the 128-bit pair reads, the two
//               output streams i and i + n/2 (64-bit stores at GROUP = 1, 128-bit stores of two neighbours at GROUP >= 2),
//               the bit-reversed 128-bit scatter of the loads, and the 128-bit linear read-out; only GROUP = 2 keeps a 2-way
//               conflict on its pair reads.
enum CgLayout { CG_LINEAR = 0, CG_PADDED = 1, CG_SWIZZLED = 2 };
template <typename E, int GROUP, int LAYOUT> struct CgMap {
  static constexpr u32 CH = 2 * GROUP, PADE = 16 / sizeof(E);
  __device__ __forceinline__ static u32 at(u32 x) {
    if (LAYOUT == CG_PADDED) return x + (x / CH) * PADE;
    if (LAYOUT == CG_SWIZZLED) return x ^ (((x >> 4) & 15u) << 1) ^ (((x >> 9) & 7u) << 1);
    return x;
  }
  __host__ __device__ static constexpr u32 span(u32 n) { return LAYOUT == CG_PADDED ? n + (n / CH) * PADE : n; }
  // at(x + d) from at(x) for x a multiple of a power of two > d, d < 16 (x, x + d inside one padding chunk): the swizzle
  // only XORs bits 1..4 with functions of bits >= 4, so it commutes with setting low bits
  __device__ __forceinline__ static u32 step(u32 ax, u32 d) { return LAYOUT == CG_SWIZZLED ? (ax ^ d) : (ax + d); }
};

// two neighbouring coefficients (2i, 2i+1): one 16-byte (8-byte for 32-bit lanes) LDS access in every layout
template <typename E> struct alignas(2 * sizeof(E)) CgPair { E lo, hi; };

// The reference butterfly with CANONICAL inputs and outputs (cg_ntt.py:57-59): t = omega * right, (left + t) % q, (left - t) % q.
// SPLIT (64-bit lanes, q = 2^k - c: the plan's tables hold split constants, modarith.h): the product rides the multiply-add
// chain as left + t' with t' < 5q, the difference is left + 5q - t'; one fold (-> below 2q) and one conditional
// subtraction make each canonical.  Otherwise: Shoup product and compare-select (any odd q).
template <typename E, bool SPLIT>
__device__ __forceinline__ void cg_butterfly(E left, E right, typename TwOf<E>::type w, const Arith<E>& ar, E& sum, E& dif) {
  if constexpr (SPLIT) {
    const u64 x = mul_sp_acc(left, right, w, ar.sk);                // left < q, right < q: t' < 2^(k+1) + q/8 + 2^(k+1) + eps < 5q
    const u64 y = ((left << 1) + ar.qmul[5]) - x;                  // left + 5q - t'
    sum = csub(fold(x, ar.k, ar.fold_c), ar.q);
    dif = csub(fold(y, ar.k, ar.fold_c), ar.q);
  } else {
    const E t = mul_tw(right, w, ar.q);                             // :57
    sum = csub((E)(left + t), ar.q);                                // :58
    dif = left >= t ? left - t : left + (ar.q - t);                 // :59
  }
}
// a * w mod q, canonical, for ANY word a (twist :82-83, untwist / n^-1 :74-75,:92); w = nullptr: a mod q
template <typename E, bool SPLIT>
__device__ __forceinline__ E cg_mul(E a, const typename TwOf<E>::type* w, const Arith<E>& ar) {
  typedef Policy<E, SPLIT> P;
  if (w) return P::mul_tw_canon(a, *w, ar);
  if constexpr (SPLIT) return P::canon(a, ar);
  else return mul_tw(a, ar.one, ar.q);
}

// Lazy form of the same butterfly for the SPLIT policy when no per-stage trace is asked for (outputs only congruent
// mod q): left is folded (< 2^k + eps), the product rides the chain, the difference is 2 left + 6q - x.  With every
// input below 7.01 * 2^k (true for canonical inputs and preserved by the butterfly: t' < 4 * 2^k + 7.01 * 2^k / 8 + eps
// < 4.9 * 2^k <= 6q) both outputs stay below 7.01 * 2^k; h_cg_lazy_ok() replays these bounds exactly for the plan's (k, c).
__device__ __forceinline__ void cg_butterfly_lazy(u64 left, u64 right, Tw64 w, const Arith<u64>& ar, u64& sum, u64& dif) {
  const u64 u = fold(left, ar.k, ar.fold_c);
  const u64 x = mul_sp_acc(u, right, w, ar.sk);
  dif = ((u << 1) + ar.qmul[6]) - x;
  sum = x;
}

// Lane-steps one thread runs per stage: workgroups have n/2/GROUP threads up to 1024 (launch_cg_t); beyond that a thread
// takes several (n = 4096 / GROUP = 1: two)
// BIG: the n = 8192 instantiation for 64-bit lanes (twice the lane-steps and pairs per thread; own kernels so that the
// n <= 4096 ones keep their register budget)
template <typename E, int GROUP, bool BIG = false> struct CgShape {
  static constexpr int MAXN = (sizeof(E) == 8 && !BIG) ? 4096 : 8192;
  static constexpr int ITERS = (MAXN / 2 / GROUP) > 1024 ? (MAXN / 2 / GROUP) / 1024 : 1;
  static constexpr int KEEP = GROUP * ITERS;           // pairs of A^ one thread holds for the pointwise product
  static constexpr int THREADS_MAX = (MAXN / 2 / GROUP) > 1024 ? 1024 : (MAXN / 2 / GROUP);
  // waves per SIMD the register allocator leaves room for = what two workgroups per CU (the LDS limit) amount to:
  // 8 for 1024-thread workgroups (<= 64 VGPRs), 4 for 512, 2 for 256 (GROUP = 8 at 64-bit: 8 butterflies and 8 pairs of A^ per thread)
  // (BIG and the 32-bit kernels, whose per-thread arrays are sized for n = 8192: one workgroup's worth, no spills)
  static constexpr int MIN_WAVES = (BIG || sizeof(E) == 4) ? (THREADS_MAX / 256 < 1 ? 1 : THREADS_MAX / 256)
                                                           : (2 * THREADS_MAX / 256 < 1 ? 1 : 2 * THREADS_MAX / 256);
};

// One CG transform in LDS: src holds the bit-reversed input; log2(n) stages
// ping-pong between src and dst; returns the buffer holding the natural-order
// result.  (cg_ntt.py:49-64.)  If trace != nullptr every stage's output is
// also written there ([logn][n]).  The geometry is constant: a thread's read and write addresses are the same in
// every stage and are computed once.  LAZYB: lazy butterflies (cg_butterfly_lazy), else canonical ones.
template <typename E, int GROUP, int LAYOUT, bool SPLIT, bool LAZYB, bool BIG>
__device__ E* cg_stages(E* src, E* dst, const typename TwOf<E>::type* __restrict__ omega_tab, u32 n, u32 logn,
                        const Arith<E>& ar, E* trace) {
  typedef CgMap<E, GROUP, LAYOUT> M;
  typedef CgPair<E> Pair;
  constexpr int ITERS = CgShape<E, GROUP, BIG>::ITERS;
  const u32 pairs = n >> 1;
  u32 rd[ITERS], wlo[ITERS], whi[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const u32 i0 = (threadIdx.x + (u32)it * blockDim.x) * GROUP;
    rd[it] = M::at(2 * i0); wlo[it] = M::at(i0); whi[it] = M::at(i0 + pairs);
  }
  for (u32 stage = 1; stage <= logn; ++stage) {
    const u32 k = n >> stage;                                      // cg_ntt.py:50
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const u32 i0 = (threadIdx.x + (u32)it * blockDim.x) * GROUP;
      if (i0 >= pairs) break;
      E left[GROUP], right[GROUP], sum[GROUP], dif[GROUP];
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {                            // :55-56 (8 at a time: cg_ntt_8butterfly.py:70-77)
        const Pair v = *reinterpret_cast<const Pair*>(src + M::step(rd[it], 2 * g));
        left[g] = v.lo; right[g] = v.hi;
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const typename TwOf<E>::type w = omega_tab[(i0 + g) & ~(k - 1)];  // omega_s^(i//k) = omega^(k*(i//k))  (:51,:54)
        if constexpr (LAZYB) cg_butterfly_lazy(left[g], right[g], w, ar, sum[g], dif[g]);
        else cg_butterfly<E, SPLIT>(left[g], right[g], w, ar, sum[g], dif[g]);             // :57-59
      }
      if (GROUP == 1) {
        dst[wlo[it]] = sum[0];
        dst[whi[it]] = dif[0];
      } else {
#pragma unroll
        for (int g = 0; g < GROUP; g += 2) {                       // neighbours share one 16-byte store
          Pair a; a.lo = sum[g]; a.hi = sum[g + 1];
          Pair b; b.lo = dif[g]; b.hi = dif[g + 1];
          *reinterpret_cast<Pair*>(dst + M::step(wlo[it], g)) = a;
          *reinterpret_cast<Pair*>(dst + M::step(whi[it], g)) = b;
        }
      }
    }
    __syncthreads();
    if (trace) {
      for (u32 i = threadIdx.x; i < n; i += blockDim.x) trace[(size_t)(stage - 1) * n + i] = dst[M::at(i)];
    }
    E* t = src; src = dst; dst = t;                                // :63-64
  }
  return src;
}

// bit_reverse_list on load (cg_ntt.py:21-26,:39): reordered[rev(idx)] = f(values[idx]).  Thread t takes idx = t and
// idx = t + n/2, whose images rev(t) = 2 rev'(t) and 2 rev'(t) + 1 (rev' over log2(n) - 1 bits) are neighbours:
// coalesced global reads, ONE 16-byte scattered LDS write per pair.  tw == nullptr: plain reduction mod q.
// lazy (SPLIT policy, lazy stages follow): only congruent mod q and below the lazy stages' input bound (h_cg_lazy_ok):
// the bare split-constant product of any word, or one fold of it.
template <typename E, bool SPLIT>
__device__ __forceinline__ E cg_mul_lazy(E a, const typename TwOf<E>::type* w, const Arith<E>& ar) {
  if constexpr (SPLIT) return w ? mul_sp(a, *w, ar.sk) : fold(a, ar.k, ar.fold_c);
  else return cg_mul<E, SPLIT>(a, w, ar);
}
template <typename E, int GROUP, int LAYOUT, bool SPLIT>
__device__ void cg_load_brv(E* buf, const E* __restrict__ in, const typename TwOf<E>::type* __restrict__ tw, u32 n, u32 logn,
                            const Arith<E>& ar, bool lazy = false) {
  typedef CgMap<E, GROUP, LAYOUT> M;
  const u32 half = n >> 1;
  for (u32 t = threadIdx.x; t < half; t += blockDim.x) {
    CgPair<E> v;                                                   // twist :82-83 / implicit % of :55-58
    if (SPLIT && lazy) {
      v.lo = cg_mul_lazy<E, SPLIT>(in[t], tw ? tw + t : nullptr, ar);
      v.hi = cg_mul_lazy<E, SPLIT>(in[t + half], tw ? tw + t + half : nullptr, ar);
    } else {
    v.lo = cg_mul<E, SPLIT>(in[t], tw ? tw + t : nullptr, ar);
    v.hi = cg_mul<E, SPLIT>(in[t + half], tw ? tw + t + half : nullptr, ar);
    }
    *reinterpret_cast<CgPair<E>*>(buf + M::at(2 * (__brev(t) >> (33 - logn)))) = v;
  }
  __syncthreads();
}

// natural-order read-out, two coefficients per thread and step (16-byte LDS reads, coalesced 16-byte global stores);
// out[i] = r[i] * scale[i] (per-coefficient table), else r[i] * uni (one constant), else r[i] (made canonical if `canon`:
// lazy stages leave values only congruent mod q)
template <typename E, int GROUP, int LAYOUT, bool SPLIT>
__device__ void cg_store_out(E* __restrict__ out, const E* r, u32 n, const typename TwOf<E>::type* __restrict__ scale,
                             const typename TwOf<E>::type* uni, const Arith<E>& ar, bool canon = false) {
  typedef CgMap<E, GROUP, LAYOUT> M;
  for (u32 i = 2 * threadIdx.x; i < n; i += 2 * blockDim.x) {
    CgPair<E> v = *reinterpret_cast<const CgPair<E>*>(r + M::at(i));
    if (scale) { v.lo = cg_mul<E, SPLIT>(v.lo, scale + i, ar); v.hi = cg_mul<E, SPLIT>(v.hi, scale + i + 1, ar); }
    else if (uni) { v.lo = cg_mul<E, SPLIT>(v.lo, uni, ar); v.hi = cg_mul<E, SPLIT>(v.hi, uni, ar); }
    else if (canon) { v.lo = cg_mul<E, SPLIT>(v.lo, nullptr, ar); v.hi = cg_mul<E, SPLIT>(v.hi, nullptr, ar); }
    *reinterpret_cast<CgPair<E>*>(out + i) = v;
  }
}

// A^ of the product modes waits in registers while b is transformed (two LDS images per workgroup instead of three:
// two workgroups per CU at n = 4096 / 64-bit): thread t keeps coefficients t + k blockDim and t + k blockDim + n/2,
// k < CgShape::KEEP.
constexpr int CG_MODE_LAZY = 0x100;   // or-ed into the kernel's mode: lazy butterflies allowed where no trace is taken (h_cg_lazy_ok)
template <typename E, int GROUP, int LAYOUT, bool SPLIT, bool BIG>
__global__ void __launch_bounds__((CgShape<E, GROUP, BIG>::THREADS_MAX), (CgShape<E, GROUP, BIG>::MIN_WAVES))
cg_kernel(PlanView<E> pv, int mode_flags, const E* __restrict__ a, const E* __restrict__ b, E* __restrict__ out, E* trace, u32 batch) {
  typedef CgMap<E, GROUP, LAYOUT> M;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  const u32 n = pv.n, logn = pv.logn, half = n >> 1;
  const u32 span = (M::span(n) + 3u) & ~3u;
  E* p0 = reinterpret_cast<E*>(tn_smem);
  E* p1 = p0 + span;
  const Arith<E> ar = pv.ar;
  const int mode = mode_flags & 0xff;
  constexpr bool LZ = SPLIT;                                       // lazy stages exist only for the split policy
  const bool lazy = LZ && (mode_flags & CG_MODE_LAZY) && !trace;
  // run one transform: lazy stages where allowed, canonical ones otherwise (always when tracing)
  auto stages = [&](E* src, E* dst, const typename TwOf<E>::type* tab, E* tr) -> E* {
    if constexpr (LZ) { if (lazy) return cg_stages<E, GROUP, LAYOUT, SPLIT, LZ, BIG>(src, dst, tab, n, logn, ar, nullptr); }
    return cg_stages<E, GROUP, LAYOUT, SPLIT, false, BIG>(src, dst, tab, n, logn, ar, tr);
  };
  for (u32 row = blockIdx.x; row < batch; row += gridDim.x) {
    const size_t off = (size_t)row * n;
    E* tr = trace ? trace + (size_t)row * logn * n : nullptr;
    if (mode == CG_NTT_FWD || mode == CG_TWIST_FWD) {
      cg_load_brv<E, GROUP, LAYOUT, SPLIT>(p0, a + off, mode == CG_TWIST_FWD ? pv.psi_pow : nullptr, n, logn, ar, lazy);
      E* r = stages(p0, p1, pv.omega_pow, tr);
      cg_store_out<E, GROUP, LAYOUT, SPLIT>(out + off, r, n, nullptr, nullptr, ar, lazy);
    } else if (mode == CG_NTT_INV) {                               // cg_intt: cg_ntt.py:68-75
      cg_load_brv<E, GROUP, LAYOUT, SPLIT>(p0, a + off, nullptr, n, logn, ar, lazy);
      E* r = stages(p0, p1, pv.omega_inv_pow, nullptr);
      cg_store_out<E, GROUP, LAYOUT, SPLIT>(out + off, r, n, nullptr, &ar.fninv, ar);              // :74-75  (fninv: n^-1 in the plan's table format)
    } else {                                                       // nwc_poly_mult: cg_ntt.py:78-92
      // CG_CYCLIC_POLYMUL: the same chain without twist/untwist = python_poly_mult
      // (test/cocotb_tests/test_ntt_poly_mult.py:38-43), what the RTL / RoCC accelerator computes
      const typename TwOf<E>::type* twist = (mode == CG_CYCLIC_POLYMUL) ? nullptr : pv.psi_pow;
      cg_load_brv<E, GROUP, LAYOUT, SPLIT>(p0, a + off, twist, n, logn, ar, lazy);                // :82
      E* ra = stages(p0, p1, pv.omega_pow, nullptr);                                              // :86
      // keep A^ in registers: the pairs this thread will need for the pointwise product (t, t + n/2)
      constexpr int KEEP = CgShape<E, GROUP, BIG>::KEEP;                // >= (n/2) / blockDim for every launch shape (launch_cg_t)
      E ka_lo[KEEP], ka_hi[KEEP];
#pragma unroll
      for (int k = 0; k < KEEP; ++k) {
        const u32 t = threadIdx.x + (u32)k * blockDim.x;
        if (t < half) { ka_lo[k] = ra[M::at(t)]; ka_hi[k] = ra[M::at(t + half)]; }
      }
      __syncthreads();
      cg_load_brv<E, GROUP, LAYOUT, SPLIT>(p0, b + off, twist, n, logn, ar, lazy);                // :83
      E* rb = stages(p0, p1, pv.omega_pow, nullptr);                                              // :87
      E* f2 = (rb == p0) ? p1 : p0;
#pragma unroll
      for (int k = 0; k < KEEP; ++k) {                                                     // :88, stored bit-reversed for :73
        const u32 t = threadIdx.x + (u32)k * blockDim.x;
        if (t < half) {
          const E a0 = ka_lo[k], a1 = ka_hi[k], b0 = rb[M::at(t)], b1 = rb[M::at(t + half)];
          CgPair<E> v;
          if (LZ && lazy) {                                        // lazy stages on both sides: the fused kernels' lazy product (< 2q)
            v.lo = pointwise_lazy(a0, b0, ar);
            v.hi = pointwise_lazy(a1, b1, ar);
          } else {
            v.lo = mulmod_barrett(a0, b0, ar.q, ar.mu, ar.k);
            v.hi = mulmod_barrett(a1, b1, ar.q, ar.mu, ar.k);
          }
          *reinterpret_cast<CgPair<E>*>(f2 + M::at(2 * (__brev(t) >> (33 - logn)))) = v;
        }
      }
      __syncthreads();
      E* rc = stages(f2, rb, pv.omega_inv_pow, nullptr);                                          // :90 (:72-73)
      cg_store_out<E, GROUP, LAYOUT, SPLIT>(out + off, rc, n, twist ? pv.psi_inv_ninv : nullptr, &ar.fninv, ar);   // :74-75 and :91-92 in one exact product
    }
    __syncthreads();
  }
}

template <typename E, int GROUP, int LAYOUT, bool SPLIT, bool BIG = false>
static hipError_t launch_cg_t(const tn_plan* p, int mode, const void* a, const void* b, void* out, void* trace, size_t batch,
                              hipStream_t s) {
  typedef CgMap<E, GROUP, LAYOUT> M;
  const u32 span = (M::span(p->n) + 3u) & ~3u;
  const size_t lds_bytes = (size_t)2 * span * sizeof(E);
  auto kern = cg_kernel<E, GROUP, LAYOUT, SPLIT, BIG>;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  // one lane-step per thread and stage: n/2/GROUP threads, capped (then a thread takes CgShape::ITERS lane-steps)
  u32 threads = p->n / 2 / GROUP;
  threads = threads < 64 ? 64 : (threads > (u32)CgShape<E, GROUP, BIG>::THREADS_MAX ? (u32)CgShape<E, GROUP, BIG>::THREADS_MAX : threads);
  if ((p->n / 2 / GROUP + threads - 1) / threads > (u32)CgShape<E, GROUP, BIG>::ITERS ||
      (p->n / 2 + threads - 1) / threads > (u32)CgShape<E, GROUP, BIG>::KEEP) return hipErrorInvalidValue;
  if (SPLIT && p->cg_lazy) mode |= CG_MODE_LAZY;
  const u32 grid = (u32)(batch < (size_t)1 << 20 ? batch : (size_t)1 << 20);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, s, make_view<E>(p), mode, (const E*)a, (const E*)b, (E*)out,
                     (E*)trace, (u32)batch);
  return hipGetLastError();
}

template <typename E, int LAYOUT, bool SPLIT>
static hipError_t launch_cg_l(const tn_plan* p, int mode, int group, const void* a, const void* b, void* out,
                              void* trace, size_t batch, hipStream_t s) {
  if constexpr (sizeof(E) == 8 && LAYOUT == CG_LINEAR) {
    if (p->n > 4096) {                                            // n = 8192 at 64-bit lanes: the BIG kernels (linear layout only)
      switch (group) {
        case 1: return launch_cg_t<E, 1, LAYOUT, SPLIT, true>(p, mode, a, b, out, trace, batch, s);
        case 2: return launch_cg_t<E, 2, LAYOUT, SPLIT, true>(p, mode, a, b, out, trace, batch, s);
        case 4: return launch_cg_t<E, 4, LAYOUT, SPLIT, true>(p, mode, a, b, out, trace, batch, s);
        case 8: return launch_cg_t<E, 8, LAYOUT, SPLIT, true>(p, mode, a, b, out, trace, batch, s);
        default: return hipErrorInvalidValue;
      }
    }
  }
  switch (group) {
    case 1: return launch_cg_t<E, 1, LAYOUT, SPLIT>(p, mode, a, b, out, trace, batch, s);
    case 2: return launch_cg_t<E, 2, LAYOUT, SPLIT>(p, mode, a, b, out, trace, batch, s);
    case 4: return launch_cg_t<E, 4, LAYOUT, SPLIT>(p, mode, a, b, out, trace, batch, s);
    case 8: return launch_cg_t<E, 8, LAYOUT, SPLIT>(p, mode, a, b, out, trace, batch, s);
    default: return hipErrorInvalidValue;
  }
}

// SPLIT: the plan's tables hold split constants (lazy plans with 64-bit lanes; every table of such a plan does)
template <typename E, bool SPLIT>
static hipError_t launch_cg_e(const tn_plan* p, int mode, int group, int layout, const void* a, const void* b, void* out,
                              void* trace, size_t batch, hipStream_t s) {
  if (p->n < 64 || (group == 1 && layout == CG_PADDED)) layout = layout == CG_SWIZZLED && p->n >= 64 ? layout : CG_LINEAR;
  if (sizeof(E) == 8 && p->n > 4096) layout = CG_LINEAR;              // the layout only matters to the conflict sweep (n = 4096): same bits
  if (p->n < (u32)(4 * group)) group = 1;                               // tiny n: the grouped store pairs assume n >= 4 GROUP
  switch (layout) {
    case CG_LINEAR: return launch_cg_l<E, CG_LINEAR, SPLIT>(p, mode, group, a, b, out, trace, batch, s);
    case CG_PADDED: return launch_cg_l<E, CG_PADDED, SPLIT>(p, mode, group, a, b, out, trace, batch, s);
    case CG_SWIZZLED: return launch_cg_l<E, CG_SWIZZLED, SPLIT>(p, mode, group, a, b, out, trace, batch, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_cg(const tn_plan* p, int mode, int group, int layout, const void* a, const void* b, void* out, void* trace,
                     size_t batch, hipStream_t s) {
  if (batch == 0) return hipSuccess;
#ifdef TN_ONLY_MAIN
  return hipErrorInvalidValue;
#else
  if (p->elem_bytes == 8)
    return p->lazy ? launch_cg_e<u64, true>(p, mode, group, layout, a, b, out, trace, batch, s)
                   : launch_cg_e<u64, false>(p, mode, group, layout, a, b, out, trace, batch, s);
  return launch_cg_e<u32, false>(p, mode, group, layout, a, b, out, trace, batch, s);
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
