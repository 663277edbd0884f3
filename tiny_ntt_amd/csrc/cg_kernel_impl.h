// cg_kernel_impl.h — the constant-geometry kernel (K1-K5/K7 of SURVEY.md §2; BASELINE config 5) and its launcher template.
// Included by the cg_part*.hip translation units, each of which instantiates one slice of
// (lane width, arithmetic, lane grouping, LDS layout); launch_cg() in kernels.hip picks the slice for a plan.
//
// The reference's dataflow (cg_ntt.py:49-64, cg_ntt_8butterfly.py:61-97) run as TRIPS of log2(2 GROUP) stages in registers
// (cg_core.h): GROUP = 8 -> four stages per LDS round trip, three trips at n = 4096.  Per workgroup: ONE image of the
// polynomial and the twiddle table omega^j, j <= n/2, staged in LDS once per persistent workgroup.  Twiddles that do not
// depend on the lane (first trip; later trips while k >= 64 GROUP) are scalar loads from the L2-resident table.
// The inverse transform of a product reads the same LDS table backwards (omega^-j = -omega^(n/2-j): the butterfly's two
// outputs change places), so one table serves both directions.
// Per product: read a, read b, write c in HBM; 6 LDS transposes (two per transform), 12 workgroup barriers at GROUP = 8.
// Persistent workgroups; rows handed out by a device-wide counter (plan.h: sched_acquire) or at a fixed stride.
#pragma once
#include <hip/hip_runtime.h>
#include "plan.h"
#include "cg_core.h"
#include "dev_addr.h"

#ifndef TN_CG_DYNAMIC_ROWS
#define TN_CG_DYNAMIC_ROWS 1     // 1: persistent workgroups take their rows from a device counter (fixed stride below TN_CG_DYNAMIC_MIN rows each)
#endif
#ifndef TN_CG_DYNAMIC_MIN
#define TN_CG_DYNAMIC_MIN 8
#endif
#ifndef TN_CG_CHUNK_BYTES
#define TN_CG_CHUNK_BYTES 32768  // bytes of one operand handed out per atomicAdd (as TN_SCHED_CHUNK_BYTES of the fused kernels)
#endif
#ifndef TN_CG_NT_STREAM
#define TN_CG_NT_STREAM 1        // 1: non-temporal loads/stores for the streamed operands (keeps L2 for the tables)
#endif

// every lambda of the kernel body must be inlined: a call would pass the captured register arrays through scratch memory
#define TN_INL __attribute__((always_inline))

#ifndef TN_CG_PINGPONG_MAXG
#define TN_CG_PINGPONG_MAXG 0
#endif
#ifndef TN_CG_CT_THREADS
#define TN_CG_CT_THREADS 0       // > 0: threads per polynomial of the n = 4096 kernels whatever the GROUP (see CgShape)
#endif

namespace tn {

constexpr int CG_FLAG_RESTAGE = 0x100;   // or-ed into the kernel's mode: omega^(n/2) != -1 (any-psi plans): the inverse transform of a
                                         // product reads a re-staged inverse table instead of the forward one backwards

// Lane-steps one thread runs per trip.  Workgroups have n / (2 GROUP) threads up to 1024; beyond that a thread takes
// several lane-steps (GROUP = 1: two at n = 4096).  BIG: the n = 8192 instantiations of GROUP 1 and 2 (twice the
// lane-steps per thread; own kernels so that the n <= 4096 ones keep their register budget).
// CTLOGN = 12 (n = 4096 compiled in: the sweep of BASELINE config 5): TN_CG_CT_THREADS threads per polynomial whatever the
// GROUP, i.e. 4096 / (2 GROUP) / threads lane-steps per thread.
template <typename E, int GROUP, bool BIG, int CTLOGN = 0> struct CgShape {
  static constexpr int R = 2 * GROUP;
  static constexpr bool FIXED = CTLOGN == 12 && TN_CG_CT_THREADS > 0 && (4096 / R) > TN_CG_CT_THREADS;
  static constexpr int MAXN = BIG ? 8192 : (GROUP >= 4 ? 8192 : 4096);
  static constexpr int ITERS = FIXED ? (4096 / R) / (TN_CG_CT_THREADS > 0 ? TN_CG_CT_THREADS : 1) : ((MAXN / R) > 1024 ? (MAXN / R) / 1024 : 1);
  static constexpr int THREADS_MAX = FIXED ? TN_CG_CT_THREADS : ((MAXN / R) > 1024 ? 1024 : (MAXN / R));
  // waves per SIMD the register allocator leaves room for = what two workgroups per CU (the LDS limit at n = 4096 / 64-bit)
  // amount to: GROUP 8: 2 x 256 threads -> 2 (<= 256 VGPRs: 16 coefficients, 16 of A^ and 16 prefetched per thread);
  // GROUP 4: 2 x 512 -> 4; GROUP 1, 2: 2 x 1024 -> 8; BIG: one workgroup per CU -> 4
  static constexpr int MIN_WAVES = FIXED ? (2 * TN_CG_CT_THREADS / 256 < 1 ? 1 : 2 * TN_CG_CT_THREADS / 256)
                                         : BIG ? 4 : (GROUP >= 8 ? 2 : (GROUP == 4 ? 4 : (GROUP <= TN_CG_PINGPONG_MAXG ? 4 : 8)));
  // GROUP <= TN_CG_PINGPONG_MAXG: two images used alternately, ONE barrier per transpose (a stage per trip at GROUP 1 leaves no
  // arithmetic to hide a second barrier behind); 96 KiB per workgroup at n = 4096 / 64-bit -> one workgroup per CU
  static constexpr bool PINGPONG = !BIG && GROUP <= TN_CG_PINGPONG_MAXG;
};

template <typename E, int GROUP, int LAYOUT, int AM, bool BIG, int CTLOGN>
__global__ void __launch_bounds__((CgShape<E, GROUP, BIG, CTLOGN>::THREADS_MAX), (CgShape<E, GROUP, BIG, CTLOGN>::MIN_WAVES))
cg_kernel(const Arith<E> ar, u32 logn_rt, int mode_flags, const typename TwOf<E>::type* __restrict__ om_fwd,
          const typename TwOf<E>::type* __restrict__ om_inv, const typename TwOf<E>::type* __restrict__ psi_pow,
          const typename TwOf<E>::type* __restrict__ psi_inv_ninv, const typename TwOf<E>::type* __restrict__ psi_inv_pow,
          const E* __restrict__ a, const E* __restrict__ b, E* __restrict__ out, E* __restrict__ trace, u32 batch, u32* sched, u32 chunk) {
  // (the tables are separate __restrict__ arguments so that wave-uniform twiddle loads become scalar loads: see polymul_fused_kernel)
  typedef CgGeom<GROUP> Ge;
  typedef CgMap<E, GROUP, LAYOUT> M;
  typedef CgArith<E, AM> A;
  typedef typename TwOf<E>::type Tw;
  typedef CgPair<E> Pair;
  typedef typename TwRawOf<E>::type TwRaw;
  constexpr int R = Ge::R, L = Ge::L, ITERS = CgShape<E, GROUP, BIG, CTLOGN>::ITERS;
  const u32 logn = CTLOGN ? (u32)CTLOGN : logn_rt;
  const u32 n = 1u << logn, TP = n >> L;                          // TP: lane-steps per polynomial
  const u32 cs = logn - L;                                        // log2 TP: column e of a lane-step starts at e << cs
  const u32 ntrips = Ge::ntrips(logn), r1 = Ge::first_stages(logn);
  const bool big = n >= 512 && TP >= 256;                         // the table swizzle permutes inside aligned 256-record blocks: it maps record n/2 to
                                                                  // itself and commutes with the per-stage block offsets h (n >> (j+1)) >= TP from there on
  const int mode = mode_flags & 0xff;
  const bool restage = (mode_flags & CG_FLAG_RESTAGE) != 0;

  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  constexpr bool PINGPONG = CgShape<E, GROUP, BIG, CTLOGN>::PINGPONG;
  const u32 img_elems = (M::span(n) + 3u) & ~3u;
  E* img = reinterpret_cast<E*>(tn_smem);
  Tw* ltab = reinterpret_cast<Tw*>(img + (PINGPONG ? 2u : 1u) * img_elems);
  u32 pp = 0;                                                     // ping-pong: element offset of the image the next transpose writes
  u32* lds_next = reinterpret_cast<u32*>(ltab + (n >> 1) + 1);     // two slots: the row this workgroup takes after the current one (double-buffered)
  auto stage_table = [&](const Tw* __restrict__ src) TN_INL {            // omega^j (or omega^-j), j <= n/2
    for (u32 j = threadIdx.x; j <= (n >> 1); j += blockDim.x) ltab[cg_twmap<GROUP, LAYOUT>(j, big)] = src[j];
  };

  // lane-step it of this thread: threadIdx.x + it * blockDim.x.  With n compiled in the launcher starts exactly TP / ITERS
  // threads, so every lane-step is live.
  auto lane_step = [&](int it) TN_INL -> u32 { return threadIdx.x + (u32)it * (CTLOGN ? (TP / ITERS) : blockDim.x); };
  auto is_live = [&](int it) TN_INL -> bool { return CTLOGN ? true : lane_step(it) < TP; };

#ifdef TN_CG_STAMPS
  // DIAGNOSTIC BUILD ONLY (tools/gpu_cg_stamps.py): per-wave cycle counts (s_memtime) of the phases of a product row and of
  // the time spent waiting at workgroup barriers, accumulated in scalar registers and written once, at the end, to a
  // buffer of their own (the trace pointer, unused by products); no output value depends on them.
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_bar = 0, st_last = __builtin_amdgcn_s_memtime();
  const unsigned long long st_begin = st_last, st_rbegin = __builtin_amdgcn_s_memrealtime();
#define TN_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_last; st_last = t_; } while (0)
#define TN_BARRIER() do { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); st_bar += __builtin_amdgcn_s_memtime() - t0_; } while (0)
#else
#define TN_STAMP(k) do { } while (0)
#define TN_BARRIER() __syncthreads()
#endif

  // x: one lane-step's registers in bit-reversed order (x[brvL(e')] = element ls + e' TP of the input list, cg_ntt.py:39)
  // -> natural order (x[e] = element ls + e TP of the transform).  glob: the table in global memory for the trips whose
  // twiddles are wave-uniform; the LDS table is read directly, or backwards with the outputs swapped (rev).
  // The first trip of a transform (registers only).  merge_: stage 1 is the merged twist + butterfly (lazy arithmetic, twisted
  // transforms): w0[it][g] = the twist record of butterfly g's right input, which arrives untwisted (CgArith::bf_first).
  auto first_trip = [&](E (&x)[ITERS][R], const Tw* __restrict__ glob, E* tr, auto merge_, const TwRaw (&w0)[ITERS][GROUP]) TN_INL {
    constexpr bool MERGE = decltype(merge_)::value;
    // opaque zero / thread index: keep the (loop-invariant) uniform twiddle loads and per-column LDS addresses of a transform
    // inside the persistent row loop instead of in registers across it (see polymul_fused_kernel)
    const u32 zero = opaque_zero();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      const u32 lsi = opaque_copy(lane_step(it));
      const u32 T = logn > (u32)L ? __brev(lsi) >> (32 - (logn - L)) : 0u;      // first trip: thread t plays lane-step brv(t) (cg_core.h)
      auto first = [&](auto nst_) TN_INL {
        constexpr int NST = decltype(nst_)::value;
        cg_trip<E, GROUP, AM, NST, false, 0, MERGE>(x[it], ar,
          [&](auto j_, auto h_) { return glob[(u32)decltype(h_)::value * (n >> (decltype(j_)::value + 1)) + zero]; },
          [&](auto j_) {
            if constexpr (!A::LAZY) {
              if (tr) {
#pragma unroll
                for (int e = 0; e < R; ++e) tr[(size_t)decltype(j_)::value * n + Ge::pos(logn, decltype(j_)::value + 1, T, e)] = x[it][e];
              }
            }
          },
          [&](auto g_) { return tw_pack(w0[it][decltype(g_)::value]); });
      };
      if (r1 == (u32)L) first(std::integral_constant<int, L>());
      else if constexpr (L >= 2) {
        if (r1 == 1) first(std::integral_constant<int, 1>());
        else if constexpr (L >= 3) {
          if (r1 == 2) first(std::integral_constant<int, 2>());
          else if constexpr (L >= 4) first(std::integral_constant<int, 3>());
        }
      }
    }
  };
  // ... and the rest of it: the transposes and the later trips.
  auto rest = [&](E (&x)[ITERS][R], const Tw* __restrict__ glob, bool rev, E* tr) TN_INL {
    const u32 zero = opaque_zero();
    // Two workgroup barriers per LDS transpose, both around the WRITE: one before it (every wave has read what the image
    // held: the previous trip's columns, or the previous transform's) and one after it (the columns are in the image).
    // A wave's reads are followed by its arithmetic, not by a barrier, so no wave waits at a barrier for LDS latency.
    E* wimg = img + pp;                                             // the image this transpose goes through
    if (ntrips > 1) {
      if constexpr (!PINGPONG) TN_BARRIER();
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (!is_live(it)) continue;
        const u32 lsi = opaque_copy(lane_step(it));
        const u32 T = __brev(lsi) >> (32 - (logn - L));
        if (CTLOGN && r1 == (u32)L) {                              // full first trip: columns T + e TP
          const u32 aT = M::at(T);
#pragma unroll
          for (int e = 0; e < R; ++e) wimg[M::col(aT, e, cs)] = x[it][e];
        } else {
#pragma unroll
          for (int e = 0; e < R; ++e) wimg[M::at(Ge::pos(logn, (int)r1, T, e))] = x[it][e];
        }
      }
    }
    u32 s0 = r1;
    for (u32 trip = 1; trip < ntrips; ++trip, s0 += L) {
      TN_BARRIER();                                             // the columns are in the image
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (!is_live(it)) continue;
        const u32 base = M::at((u32)R * opaque_copy(lane_step(it)));
#pragma unroll
        for (int e = 0; e < R; e += 2) {
          const Pair v = *reinterpret_cast<const Pair*>(wimg + M::step(base, e));
          x[it][e] = v.lo; x[it][e + 1] = v.hi;
        }
      }
      const bool uniform = (int)logn - (int)s0 - L >= 6;           // the trip's twiddles depend on (lane-step >> 6) only
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (!is_live(it)) continue;
        const u32 T = opaque_copy(lane_step(it)), base0 = cg_tw_base0<GROUP>(logn, s0, T);
        auto after = [&](auto j_) TN_INL {
          if constexpr (!A::LAZY) {
            if (tr) {
#pragma unroll
              for (int e = 0; e < R; ++e) tr[(size_t)(s0 + decltype(j_)::value) * n + Ge::pos(logn, decltype(j_)::value + 1, T, e)] = x[it][e];
            }
          }
        };
        // (par: parity of s0, for the scheduled arithmetic's fold-on-even-stages rule; a compile-time 0 otherwise)
        auto run = [&](auto par_) TN_INL {
          constexpr int PAR = decltype(par_)::value;
          if (uniform) {
            const u32 ub = wave_uniform(base0) + zero;
            cg_trip<E, GROUP, AM, L, false, PAR>(x[it], ar,
              [&](auto j_, auto h_) { return glob[(u32)decltype(h_)::value * (n >> (decltype(j_)::value + 1)) + (ub >> decltype(j_)::value)]; }, after);
          } else if (rev) {
            cg_trip<E, GROUP, AM, L, true, PAR>(x[it], ar,
              [&](auto j_, auto h_) {
                // record n/2 - (h BIG + lo), BIG = n >> (j+1), lo = base0 >> j < BIG  =  (n/2 - (h+1) BIG) + (BIG - lo): the table map
                // permutes inside aligned 256-record blocks and BIG is a multiple of 256 (or the map is the identity), so it
                // applies to the lane-dependent part alone and the rest is an immediate offset
                constexpr int j = decltype(j_)::value;
                const u32 blk = n >> (j + 1);
                return ltab[((n >> 1) - ((u32)decltype(h_)::value + 1u) * blk) + cg_twmap<GROUP, LAYOUT>(blk - (base0 >> j), big)];
              }, after);
          } else {
            cg_trip<E, GROUP, AM, L, false, PAR>(x[it], ar,
              [&](auto j_, auto h_) {
                constexpr int j = decltype(j_)::value;
                return ltab[(u32)decltype(h_)::value * (n >> (j + 1)) + cg_twmap<GROUP, LAYOUT>(base0 >> j, big)];
              }, after);
          }
        };
        if constexpr (AM == CGA_SPLIT_SCHED) {
          if (s0 & 1u) run(std::integral_constant<int, 1>()); else run(std::integral_constant<int, 0>());
        } else run(std::integral_constant<int, 0>());
      }
      if (trip + 1 < ntrips) {
        if constexpr (PINGPONG) { pp ^= img_elems; wimg = img + pp; }      // the other image: slower waves may still be reading this one
        else TN_BARRIER();                                      // every wave has read this trip's input
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          if (!is_live(it)) continue;
          const u32 T = opaque_copy(lane_step(it)), aT = M::at(T);
#pragma unroll
          for (int e = 0; e < R; ++e) wimg[CTLOGN ? M::col(aT, e, cs) : M::at(T + ((u32)e << cs))] = x[it][e];
        }
      }
    }
    if constexpr (PINGPONG) pp ^= img_elems;                         // the next transform starts in the image this one did not use last
  };
  // x: one lane-step's registers in bit-reversed order (x[brvL(e')] = element ls + e' TP of the input list, cg_ntt.py:39)
  // -> natural order (x[e] = element ls + e TP of the transform).  glob: the table in global memory for the trips whose
  // twiddles are wave-uniform; the LDS table is read directly, or backwards with the outputs swapped (rev).
  const TwRaw no_w0[ITERS][GROUP] = {};
  auto transform = [&](E (&x)[ITERS][R], const Tw* __restrict__ glob, bool rev, E* tr) TN_INL {
    first_trip(x, glob, tr, std::integral_constant<bool, false>(), no_w0);
    rest(x, glob, rev, tr);
  };

  // Operand and table accesses are "uniform base (row, column e: scalar unit) + the lane-step as a 32-bit offset"; the
  // lane-step is taken through opaque_copy per use so that base + offset is not a loop invariant of the row loop
  // (hoisted, every column's address is a 64-bit VGPR pair that lives across the whole loop: polymul_fused_kernel).
  // raw words of one row: x[it][e] = in[row n + ls + e TP] (unit stride across lanes)
  auto load_row = [&](E (&x)[ITERS][R], const E* __restrict__ in, u32 row, u32 zero) TN_INL {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      const u32 tl = opaque_copy(lane_step(it));
#pragma unroll
      for (int e = 0; e < R; ++e) {
        const TN_GLOBAL_AS E* cp = uniform_ptr(in + ((size_t)row << logn) + ((u32)e << cs) + zero);
#if TN_CG_NT_STREAM
        x[it][e] = __builtin_nontemporal_load(cp + tl);
#else
        x[it][e] = cp[tl];
#endif
      }
    }
  };
  // the lane-step's records of a per-coefficient table: rec[e] = tab[ls + e TP]  (requested; consumed by enter / store_row)
  auto fetch_records = [&](TwRaw (&rec)[ITERS][R], const Tw* tab) TN_INL {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      const u32 tl = opaque_copy(lane_step(it));
      static_for<0, R>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
#if defined(TN_CG_ABL_NOREC)          // timing ablation (wrong results): no per-coefficient table loads
        rec[it][e] = TwRaw{(decltype(TwRaw().x))(ar.fninv.w + e), (decltype(TwRaw().x))ar.fninv.wp};
#else
        rec[it][e] = ld_global(reinterpret_cast<const TN_GLOBAL_AS TwRaw*>(uniform_ptr(tab + ((u32)e << cs)) + tl));
#endif
      });
    }
  };
  // y[brvL(e)] = x[e] * psi^(ls + e TP)  (cg_ntt.py:82-83), or x[e] mod q — in the first trip's register order.
  // MERGED (lazy arithmetic, twisted): only the LEFT inputs of stage 1 (columns e < GROUP) are twisted here; the right inputs
  // (columns e >= GROUP = the other half of the polynomial) stay raw and their records go to w0 for CgArith::bf_first.
  constexpr bool CAN_MERGE = TN_CG_MERGE_TWIST && A::LAZY;
  auto enter = [&](E (&y)[ITERS][R], const E (&x)[ITERS][R], auto twisted_, const Tw* tab, TwRaw (&w0)[ITERS][GROUP]) TN_INL {
    constexpr bool TWISTED = decltype(twisted_)::value;
    TwRaw rec[ITERS][R];
    if constexpr (TWISTED) fetch_records(rec, tab);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      static_for<0, R>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr (TWISTED && CAN_MERGE && e >= GROUP) { y[it][Ge::brvL(e)] = x[it][e]; w0[it][Ge::brvL(e) >> 1] = rec[it][e]; }
        else if constexpr (TWISTED) y[it][Ge::brvL(e)] = A::in_mul(x[it][e], tw_pack(rec[it][e]), ar);
        else y[it][Ge::brvL(e)] = A::in_red(x[it][e], ar);
      });
    }
  };
  // the same for both operands of a product on ONE fetch of the twist records (both rows already in registers)
  auto enter2 = [&](E (&ya)[ITERS][R], const E (&xa_)[ITERS][R], E (&yb)[ITERS][R], const E (&xb_)[ITERS][R], auto twisted_, const Tw* tab,
                    TwRaw (&w0)[ITERS][GROUP]) TN_INL {
    constexpr bool TWISTED = decltype(twisted_)::value;
    TwRaw rec[ITERS][R];
    if constexpr (TWISTED) fetch_records(rec, tab);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      static_for<0, R>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr (TWISTED && CAN_MERGE && e >= GROUP) {
          ya[it][Ge::brvL(e)] = xa_[it][e]; yb[it][Ge::brvL(e)] = xb_[it][e]; w0[it][Ge::brvL(e) >> 1] = rec[it][e];
        } else if constexpr (TWISTED) {
          const Tw w = tw_pack(rec[it][e]);
          ya[it][Ge::brvL(e)] = A::in_mul(xa_[it][e], w, ar);
          yb[it][Ge::brvL(e)] = A::in_mul(xb_[it][e], w, ar);
        } else {
          ya[it][Ge::brvL(e)] = A::in_red(xa_[it][e], ar);
          yb[it][Ge::brvL(e)] = A::in_red(xb_[it][e], ar);
        }
      });
    }
  };
  // out[row n + ls + e TP] = x[e] * (psi^-(ls + e TP) n^-1)   (kind 2; cg_ntt.py:74-75 and :91-92 in one exact product),
  //                          x[e] * n^-1 (kind 1; :74-75),  canonical x[e] (kind 0)
  const Tw ninv = AM == CGA_SHOUP ? ar.ninv : ar.fninv;            // n^-1 in the record format of the plan's tables
  // Untwist in two exact factors (TN_CG_UNTWIST2): psi^-(ls + e TP) n^-1 = psi^-(e TP) [the same for every lane-step: scalar
  // loads] x psi^-ls n^-1 [one record per lane-step, loaded once per persistent workgroup].  One more multiplication per
  // coefficient (exact, so the same residue) instead of R record loads per row whose L2 latency nothing covers at the end
  // of a row (measured: gpurun_out stamps, 4.7k of 40k cycles per row at GROUP 8).
#ifndef TN_CG_UNTWIST2
#define TN_CG_UNTWIST2 1
#endif
  TwRaw out_rec[ITERS];
  if (TN_CG_UNTWIST2 && mode == CG_POLYMUL) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      if (is_live(it)) out_rec[it] = ld_global(reinterpret_cast<const TN_GLOBAL_AS TwRaw*>(uniform_ptr(psi_inv_ninv)) + lane_step(it));
  }
  // v[e] = the canonical output coefficient of column e
  auto finish_row = [&](E (&v)[ITERS][R], const E (&x)[ITERS][R], auto kind_, u32 zero, const Tw* tab) TN_INL {
    constexpr int KIND = decltype(kind_)::value;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      const u32 tl = opaque_copy(lane_step(it));
      (void)tl;
      if constexpr (KIND == 2) {
        static_for<0, R>([&](auto e_) {
          constexpr int e = decltype(e_)::value;
#if defined(TN_CG_ABL_NOREC)
          v[it][e] = A::out_mul(x[it][e], ninv, ar);
#elif TN_CG_UNTWIST2
          const Tw* up = psi_inv_pow + zero;                        // (zero: keeps the R uniform records out of registers across the row loop)
          v[it][e] = A::out_mul(e ? A::in_mul(x[it][e], up[(u32)e << cs], ar) : x[it][e], tw_pack(out_rec[it]), ar);
#else
          v[it][e] = A::out_mul(x[it][e], tw_pack(ld_global(reinterpret_cast<const TN_GLOBAL_AS TwRaw*>(uniform_ptr(tab + ((u32)e << cs)) + tl))), ar);
#endif
        });
      } else if constexpr (KIND == 1) {
#pragma unroll
        for (int e = 0; e < R; ++e) v[it][e] = A::out_mul(x[it][e], ninv, ar);
      } else {
#pragma unroll
        for (int e = 0; e < R; ++e) v[it][e] = A::out_canon(x[it][e], ar);
      }
    }
  };
  auto emit_row = [&](const E (&v)[ITERS][R], u32 row, u32 zero) TN_INL {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (!is_live(it)) continue;
      const u32 tl = opaque_copy(lane_step(it));
#pragma unroll
      for (int e = 0; e < R; ++e) {
        TN_GLOBAL_AS E* cp = uniform_ptr(out + ((size_t)row << logn) + ((u32)e << cs) + zero);
#if TN_CG_NT_STREAM
        __builtin_nontemporal_store(v[it][e], cp + tl);
#else
        cp[tl] = v[it][e];
#endif
      }
    }
  };
  auto store_row = [&](const E (&x)[ITERS][R], u32 row, auto kind_, u32 zero, const Tw* tab) TN_INL {
    E v[ITERS][R];
    finish_row(v, x, kind_, zero, tab);
    emit_row(v, row, zero);
  };

  // Row hand-out.  sched != nullptr: rows come from a device-wide counter (one atomicAdd per chunk of rows, by thread 0), as in the fused
  // kernels: the two workgroups of a CU do not run at the same speed (the SIMD's issue arbitration favours the older wave: with
  // equal fixed shares one workgroup finished after 1.83 ms and its neighbour ran the last 0.9 ms alone, profiles/r3_cg_stamps.txt),
  // so the faster one simply takes more rows and the launch ends when the work does.  sched == nullptr: fixed stride gridDim.x.
  // The index travels through LDS one row ahead of its use: while row k runs, every thread reads the index of row k+1 from one
  // slot (written during row k-1) and thread 0 requests the index of row k+2 and writes it to the other slot; the workgroup
  // barriers inside every row order each slot's write before its read and its read before the next write.
  // Rows are handed out in chunks of `chunk` consecutive rows (>= 32 KiB of one operand: one atomic per short row would make the one
  // counter address the bottleneck, as measured on the fused kernels); chunk blockIdx.x first.  The publisher's cursor: how many rows
  // of the current chunk are out and - fixed stride - which chunk that is (workgroup-uniform; the last published row is `next`).
  u32 pub_taken = 1, chunk_id = blockIdx.x, got = 0;
  bool need = pub_taken == chunk;                                  // the next row to publish opens a new chunk
  if (need && sched && threadIdx.x == 0) got = atomicAdd(&sched[0], 1u);
  auto publish = [&](u32 slot_, u32 last) TN_INL {                 // last: the row published before this one (workgroup-uniform)
    u32 v = last + 1u;
    if (need) { chunk_id += gridDim.x; v = (sched ? gridDim.x + got : chunk_id) * chunk; pub_taken = 1; }
    else pub_taken += 1;
    if (threadIdx.x == 0) lds_next[slot_] = v;                     // (only thread 0 holds the atomic's answer)
  };
  publish(0u, blockIdx.x * chunk);
  stage_table(mode == CG_NTT_INV ? om_inv : om_fwd);
  __syncthreads();

  typedef std::integral_constant<bool, true> True;
  typedef std::integral_constant<bool, false> False;
  typedef std::integral_constant<int, 0> K0;
  typedef std::integral_constant<int, 1> K1;
  typedef std::integral_constant<int, 2> K2;
  // Persistent workgroup over the rows the hand-out below gives it; the next row's first operand is requested from HBM
  // before the last transform of the current row and consumed at the top of the next iteration.  One body per mode,
  // each with its switches compiled in (a run-time switch inside a body keeps both sides' registers alive).
  E xa[ITERS][R], xn[ITERS][R];
  // AHEAD: products request BOTH operands of the next row before the inverse transform (one more row of registers while it
  // runs); a and b are then twisted together on one fetch of the twist records and nothing at the top of a row waits for
  // HBM.  Only at GROUP 8 (256-register budget): the smaller groups' budgets have no room for the third row (measured with
  // it: GROUP 2 3.37 -> 3.71 ms from spills, GROUP 4 would not fit at all; GROUP 8 2.87 -> 2.80 ms).
#ifndef TN_CG_AHEAD
#define TN_CG_AHEAD 1
#endif
  constexpr bool AHEAD = TN_CG_AHEAD && GROUP == 8;
  E xm[AHEAD ? ITERS : 1][R];
  u32 row = blockIdx.x * chunk;
  if (row < batch) {
    load_row(xn, a, row, 0u);
    if constexpr (AHEAD) { if (mode == CG_POLYMUL || mode == CG_CYCLIC_POLYMUL) load_row(xm, b, row, 0u); }
  }
  // nwc_poly_mult (cg_ntt.py:78-92); untwisted: the same chain without twist / untwist = python_poly_mult
  // (test/cocotb_tests/test_ntt_poly_mult.py:38-43), what the RTL / RoCC accelerator computes
  // The result of row k is stored AFTER the twist of row k+1 (software-pipelined): vector-memory operations retire in order and
  // the compiler drains them all at the loop back-edge, so stores issued at the bottom of a row expose their latency there on
  // every row (measured: ~950 of 40k cycles); issued here they drain behind the transforms.  (First row: the zero-initialised
  // registers go to this row's own slot, which the same thread overwrites one iteration later; no branch at the loop top.)
#ifndef TN_CG_DEFER
#define TN_CG_DEFER 0            // off: the extra row of registers across the back-edge spills at every GROUP (A/B: profiles/README.md)
#endif
  constexpr bool DEFER = TN_CG_DEFER != 0;
  E vprev[DEFER ? ITERS : 1][R];
  u32 prev_row = blockIdx.x;
  bool have_prev = false;
  if constexpr (DEFER) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
      for (int e = 0; e < R; ++e) vprev[it][e] = 0;
  }
  u32 slot = 0, next = 0;                                          // slot: which of the two LDS slots holds the next row's index
  // called once per row, after the row's first table fetch has been consumed (the atomic's latency hides behind it): thread 0
  // publishes the index of the row after `next`
  auto publish_next = [&]() TN_INL { publish(slot ^ 1u, next); };
  auto product_row = [&](auto twisted_, u32 nrow, u32 zero, const Tw* tw_in, const Tw* tw_out) TN_INL {
    constexpr bool TWISTED = decltype(twisted_)::value;
    E xb[ITERS][R];
    if constexpr (AHEAD) {
      TN_STAMP(7);
      TwRaw w0[ITERS][GROUP];
      enter2(xa, xn, xb, xm, twisted_, tw_in, w0);                 // :82-83
      publish_next();
      if constexpr (DEFER) { sched_fence(); emit_row(vprev, prev_row, zero); sched_fence(); }
      TN_STAMP(0);
      constexpr bool MERGED = TWISTED && CAN_MERGE;
      first_trip(xa, om_fwd, nullptr, std::integral_constant<bool, MERGED>(), w0);     // both first trips on one fetch of the records
      first_trip(xb, om_fwd, nullptr, std::integral_constant<bool, MERGED>(), w0);
      rest(xa, om_fwd, false, nullptr);                            // :86  A^ stays in registers
      TN_STAMP(1);
      rest(xb, om_fwd, false, nullptr);                            // :87
      TN_STAMP(2);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (!is_live(it)) continue;
        E c[R];
        static_for<0, R>([&](auto e_) {                            // :88
          constexpr int e = decltype(e_)::value;
          c[e] = A::pointwise(xa[it][e], xb[it][e], ar);
          if constexpr ((e & 1) == 1) sched_fence();               // two products in flight at a time
        });
        static_for<0, R>([&](auto e_) { constexpr int e = decltype(e_)::value; xb[it][Ge::brvL(e)] = c[e]; });   // the inverse's first-trip order (bit_reverse_list of :73)
      }
      load_row(xn, a, nrow, zero);
      load_row(xm, b, nrow, zero);
      TN_STAMP(3);
    } else {
      // vector-memory operations retire in order: b's row (HBM) is requested AFTER the twist of a has consumed its records
      // (L2), so that nothing of a's path waits for HBM; b stays in flight while a is transformed
      constexpr bool MERGED = TWISTED && CAN_MERGE;
      TwRaw w0[ITERS][GROUP];
      enter(xa, xn, twisted_, tw_in, w0);                          // :82
      publish_next();
      sched_fence();
      if constexpr (DEFER) emit_row(vprev, prev_row, zero);
      load_row(xb, b, row, zero);
      sched_fence();
      first_trip(xa, om_fwd, nullptr, std::integral_constant<bool, MERGED>(), w0);
      rest(xa, om_fwd, false, nullptr);                            // :86  A^ stays in registers
      sched_fence();                                               // (or the scheduler requests b's twist records a whole transform early)
      enter(xn, xb, twisted_, tw_in, w0);                          // :83
      first_trip(xn, om_fwd, nullptr, std::integral_constant<bool, MERGED>(), w0);
      rest(xn, om_fwd, false, nullptr);                            // :87
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (!is_live(it)) continue;
        static_for<0, R>([&](auto e_) {                            // :88, left in the inverse's first-trip order (bit_reverse_list of :73)
          constexpr int e = decltype(e_)::value;
          xb[it][Ge::brvL(e)] = A::pointwise(xa[it][e], xn[it][e], ar);
          if constexpr ((e & 1) == 1) sched_fence();               // two products in flight at a time
        });
      }
      load_row(xn, a, nrow, zero);
    }
    if (restage) { __syncthreads(); stage_table(om_inv); }         // (the transform's first barrier orders the staging before its first read)
    transform(xb, om_inv, !restage, nullptr);                      // :90 (:72-73)
    TN_STAMP(4);
    sched_fence();
    if constexpr (DEFER) {
      if constexpr (TWISTED) finish_row(vprev, xb, K2(), zero, tw_out);   // :74-75 and :91-92 in one exact product
      else finish_row(vprev, xb, K1(), zero, tw_out);
      prev_row = row; have_prev = true;
    } else {
      if constexpr (TWISTED) store_row(xb, row, K2(), zero, tw_out);
      else store_row(xb, row, K1(), zero, tw_out);
    }
    TN_STAMP(5);
    if (restage) { __syncthreads(); stage_table(om_fwd); }
  };
  while (row < batch) {
    next = wave_uniform(lds_next[slot]);                           // published at least one workgroup barrier ago
    need = pub_taken == chunk;                                     // the row after that opens a new chunk: requested now, published by publish_next()
    if (need && sched && threadIdx.x == 0) got = atomicAdd(&sched[0], 1u);
    const u32 zero = opaque_zero();                                // pins the column bases (scalar adds) inside the row loop
    const Tw* tw_in = opaque_sptr(psi_pow);
    const Tw* tw_out = opaque_sptr(psi_inv_ninv);
    const u32 nrow = next < batch ? next : row;                    // (after the last row this row is read again and dropped: no branch around the prefetch)
    if (mode == CG_POLYMUL) product_row(True(), nrow, zero, tw_in, tw_out);
    else if (mode == CG_CYCLIC_POLYMUL) product_row(False(), nrow, zero, tw_in, tw_out);
    else {
      E* tr = trace ? trace + (size_t)row * logn * n : nullptr;
      TwRaw w0[ITERS][GROUP];
      if (mode == CG_TWIST_FWD) {                                  // forward_ntt_bench: twist + cg_ntt
        enter(xa, xn, True(), tw_in, w0);
        publish_next();
        sched_fence();
        load_row(xn, a, nrow, zero);
        first_trip(xa, om_fwd, tr, std::integral_constant<bool, CAN_MERGE>(), w0);
        rest(xa, om_fwd, false, tr);
        store_row(xa, row, K0(), zero, tw_out);
      } else {
        enter(xa, xn, False(), tw_in, w0);
        publish_next();
        sched_fence();
        load_row(xn, a, nrow, zero);
        if (mode == CG_NTT_INV) { transform(xa, om_inv, false, nullptr); store_row(xa, row, K1(), zero, tw_out); }       // cg_intt: cg_ntt.py:68-75
        else { transform(xa, om_fwd, false, tr); store_row(xa, row, K0(), zero, tw_out); }
      }
    }
    if (ntrips == 1) __syncthreads();                              // (a single trip has no barrier of its own between the slot's write and its read)
    row = next; slot ^= 1u;
  }
  if constexpr (DEFER) { if (have_prev) emit_row(vprev, prev_row, 0u); }
  // the last workgroup to run out of rows re-arms the counters for the next launch that uses this slot
  if (sched && threadIdx.x == 0 && atomicAdd(&sched[1], 1u) == gridDim.x - 1) { sched[0] = 0; sched[1] = 0; }
#ifdef TN_CG_STAMPS
  if (trace && (mode == CG_POLYMUL || mode == CG_CYCLIC_POLYMUL) && (threadIdx.x & 63) == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(trace) + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12;
    for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
    o[8] = st_bar; o[9] = __builtin_amdgcn_s_memtime() - st_begin; o[10] = __builtin_amdgcn_s_memrealtime() - st_rbegin;      // [10]: the same interval on the constant 100 MHz clock
    o[11] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 /* XCC_ID */) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 /* HW_ID */) | (0 << 6) | (31 << 11));   // where the wave ran
  }
#endif
}

#ifdef TN_CG_STAMPS
// diagnostic builds: one device buffer for the stamps of the last product launch, read back by tn_debug_cg_stamps()
inline void*& tn_cg_stamp_ptr() { static void* p = nullptr; return p; }
inline size_t& tn_cg_stamp_size() { static size_t n = 0; return n; }
inline void* tn_cg_stamp_buffer(size_t bytes) {
  if (bytes > tn_cg_stamp_size()) { if (tn_cg_stamp_ptr()) (void)hipFree(tn_cg_stamp_ptr()); (void)hipMalloc(&tn_cg_stamp_ptr(), bytes); tn_cg_stamp_size() = bytes; }
  (void)hipMemset(tn_cg_stamp_ptr(), 0, bytes);
  return tn_cg_stamp_ptr();
}
#endif

// Launch one slice.  Returns hipErrorInvalidValue for shapes the instantiation cannot run (the dispatcher, launch_cg() in
// kernels.hip, only asks for valid ones).
template <typename E, int GROUP, int LAYOUT, int AM, bool BIG, int CTLOGN>
static hipError_t launch_cg_t(const tn_plan* p, int mode, const void* a, const void* b, void* out, void* trace, size_t batch,
                              hipStream_t s) {
  typedef CgMap<E, GROUP, LAYOUT> M;
  typedef CgShape<E, GROUP, BIG, CTLOGN> Sh;
  typedef typename TwOf<E>::type Tw;
  const u32 n = p->n, logn = p->logn;
  if ((int)logn < CgGeom<GROUP>::L || (CTLOGN && (int)logn != CTLOGN)) return hipErrorInvalidValue;
  const u32 tp = n / Sh::R;
  u32 threads = tp < 64 ? 64 : (tp > (u32)Sh::THREADS_MAX ? (u32)Sh::THREADS_MAX : tp);
  if ((tp + threads - 1) / threads > (u32)Sh::ITERS) return hipErrorInvalidValue;
  const size_t lds_bytes = (size_t)(Sh::PINGPONG ? 2 : 1) * ((M::span(n) + 3u) & ~3u) * sizeof(E) + (size_t)(n / 2 + 1) * sizeof(Tw) + 16;   // + the next-row slots
  auto kern = cg_kernel<E, GROUP, LAYOUT, AM, BIG, CTLOGN>;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  int per_cu = 0;
  hipError_t qe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (int)threads, lds_bytes);
  if (qe != hipSuccess || per_cu < 1) per_cu = 1;
  const size_t resident = (size_t)per_cu * (size_t)p->num_cus;
  // dynamic hand-out in chunks of >= TN_SCHED_CHUNK_BYTES of one operand when every resident workgroup gets at least TN_CG_DYNAMIC_MIN
  // chunks; otherwise single rows at a fixed stride
  size_t chunk = 1;
  if (TN_CG_DYNAMIC_ROWS) { chunk = (size_t)TN_CG_CHUNK_BYTES / ((size_t)n * sizeof(E)); if (chunk < 1) chunk = 1; }
  const bool dynamic = TN_CG_DYNAMIC_ROWS && batch >= (size_t)TN_CG_DYNAMIC_MIN * resident * chunk;
  if (!dynamic) chunk = 1;
  const size_t chunks = (batch + chunk - 1) / chunk;
  const u32 grid = (u32)(chunks < resident ? chunks : resident);
  const PlanView<E> pv = make_view<E>(p);
  if (p->general) mode |= CG_FLAG_RESTAGE;
#ifdef TN_CG_STAMPS
  if (!trace && (mode & 0xff) == CG_POLYMUL) trace = tn_cg_stamp_buffer((size_t)grid * (threads / 64) * 12 * sizeof(unsigned long long));
#endif
  // rows from the device-wide counter when every resident workgroup gets at least TN_CG_DYNAMIC_MIN rows (one counter pair per launch
  // in flight: plan.h sched_acquire; no pair free, or the stream is being captured: fixed stride)
  SchedSlot slot;
  if (dynamic) slot = sched_acquire(p, s);
  if (!slot.ptr) chunk = 1;                                       // (no pair free / stream capture: fixed stride of single rows; the grid stays)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, s, pv.ar, logn, mode, pv.omega_pow, pv.omega_inv_pow, pv.psi_pow,
                     pv.psi_inv_ninv, pv.psi_inv_pow, (const E*)a, (const E*)b, (E*)out, (E*)trace, (u32)batch, slot.ptr, (u32)chunk);
  const hipError_t le = hipGetLastError();
  sched_release(p, slot, s, le == hipSuccess);
  return le;
}

}  // namespace tn
