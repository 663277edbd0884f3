// cg_core.h — the per-lane-step body of the constant-geometry kernels (cg_kernels.hip), written as host+device templates
// so that tests/emu can step the very same index math and arithmetic on the CPU.  Nothing here touches the oracle.
//
// What it computes: the reference's constant-geometry transform (new_reference/cg_ntt.py:49-64) —
//   stage s = 1 .. log2 n, k = n >> s, butterfly i in [0, n/2):  w = omega^(k (i // k)),
//   A[i] = a[2i] + w a[2i+1],  A[i + n/2] = a[2i] - w a[2i+1]
// — with GROUP consecutive butterflies per lane-step (8 = cg_ntt_8butterfly.py:61-89).
//
// How (round 3): the G sums a lane-step writes to A[i0 .. i0+G) are exactly the inputs of butterflies i0/2 .. i0/2+G/2-1
// of the NEXT stage and its G differences those of butterflies i0/2+n/4 ..; so a lane-step that holds R = 2G
// neighbouring coefficients runs L = log2(R) stages in registers ("a trip") before anything has to move: per stage the
// registers go through the same perfect shuffle the list does (z[g] = sum_g, z[G+g] = dif_g), and after j stages
// register e holds position
//     pos(j, T, e) = (R T >> j) + (e mod 2^(L-j)) + (e >> (L-j)) (n >> j)          (T = the lane-step's index)
// of that stage's list.  After a full trip that is T + e n/R: a stride-n/R column, which one LDS transpose (write
// columns, read R neighbours) turns into the next trip's input.  log2 n stages = ceil(log2 n / L) trips; the FIRST
// trip takes the remainder so that the last one is full and its output columns T + e n/R are unit-stride across lanes
// in global memory.  For the same reason the bit-reversed load (cg_ntt.py:39) costs no transpose either: lane-step t
// loads column t + e' n/R, which is the R-neighbourhood of index brv(t) of the bit-reversed list, and first-trip
// twiddles do not depend on the lane-step (omega^(k (i // k)) with i < k), so thread t simply plays lane-step brv(t).
// The product keeps A^ and B^ in those columns, multiplies them in registers, and the inverse transform starts from
// them the same way: no LDS round trip between the forward and the inverse transforms.
#pragma once
#include "fused_core.h"

#ifndef TN_CG_MERGE_TWIST
#define TN_CG_MERGE_TWIST 1      // lazy arithmetic: the twist of the right inputs rides the stage-1 butterflies (CgArith::bf_first)
#endif
#ifndef TN_CG_FENCE
#define TN_CG_FENCE 2            // butterflies between two vector-ALU scheduling fences of a stage (0: none)
#endif

namespace tn {

enum CgLayout { CG_LINEAR = 0, CG_PADDED = 1, CG_SWIZZLED = 2 };
// Arithmetic of the butterflies: Shoup records + compare-select (any modulus); the plan's split-constant records with
// canonical values at every step (per-stage traces of a lazy 64-bit plan); split-constant records, values only congruent
// mod q between stages (h_cg_lazy_ok replays the bounds for the plan's (k, c)).
// CGA_SPLIT_SCHED: the same with a static fold schedule instead of one fold per butterfly (below; h_cg_sched_ok).
enum CgArithMode { CGA_SHOUP = 0, CGA_SPLIT_CANON = 1, CGA_SPLIT_LAZY = 2, CGA_SPLIT_SCHED = 3 };

template <int V> struct ILog2 { static constexpr int value = 1 + ILog2<V / 2>::value; };
template <> struct ILog2<1> { static constexpr int value = 0; };

template <int GROUP> struct CgGeom {
  static constexpr int G = GROUP, R = 2 * GROUP, L = ILog2<R>::value;
  static constexpr u32 brvL(u32 e) {
    u32 r = 0;
    for (int i = 0; i < L; ++i) r |= ((e >> i) & 1u) << (L - 1 - i);
    return r;
  }
  TN_HD static u32 ntrips(u32 logn) { return (logn + L - 1) / L; }
  TN_HD static u32 first_stages(u32 logn) { return logn - (ntrips(logn) - 1) * L; }
  // position (in the list of that stage) held by register e after j stages of a trip run as lane-step T
  TN_HD static u32 pos(u32 logn, int j, u32 T, u32 e) {
    return (((u32)R * T) >> j) + (e & (((u32)R >> j) - 1u)) + ((e >> (L - j)) << (logn - j));
  }
};

// LDS image of one polynomial.  LAYOUT (parameters per GROUP from tools/cg_layout_search.py: exhaustive search over the gfx950
// banking model of tests/test_lds_banks.py for the accesses of the trips — 128-bit reads of a lane-step's R neighbours,
// element-wide column writes T + e n/R for T = lane (later trips) and T = brv(lane) (first trip)):
//   CG_LINEAR   element x at x.
//   CG_PADDED   16 bytes of padding after every CH elements, CH = 16 / 32 / 64 / 64 for GROUP 8 / 4 / 2 / 1 (GROUP 1: and
//               after every 256): additive in the lane-step and in the column, so every access is one address register
//               plus an immediate.
//   CG_SWIZZLED x ^ (((x >> S1) & M1) << 1) ^ (((x >> S2) & M2) << 1) ^ (((x >> S3) & M3) << 1): pairs (2i, 2i+1) stay adjacent
//               and 16-byte aligned (bit 0 untouched, source bits all >= log2 R).
// In both, every access of every GROUP is conflict free in the model except the first trip's bit-reversed column writes
// (2-way: 16 lanes of a 64-bit store land on 8 pairs of banks whatever pair-preserving map is used).
template <int GROUP> struct CgSwz;
template <> struct CgSwz<8> { static constexpr u32 S1 = 4, M1 = 15, S2 = 9, M2 = 7, S3 = 0, M3 = 0, CH = 16, CH2 = 0, TS = 4, TM = 15; };
template <> struct CgSwz<4> { static constexpr u32 S1 = 3, M1 = 1, S2 = 4, M2 = 7, S3 = 8, M3 = 7, CH = 32, CH2 = 0, TS = 4, TM = 7; };
template <> struct CgSwz<2> { static constexpr u32 S1 = 2, M1 = 1, S2 = 3, M2 = 7, S3 = 7, M3 = 7, CH = 64, CH2 = 0, TS = 4, TM = 3; };
template <> struct CgSwz<1> { static constexpr u32 S1 = 2, M1 = 1, S2 = 4, M2 = 7, S3 = 7, M3 = 7, CH = 64, CH2 = 256, TS = 1, TM = 15; };
template <typename E, int GROUP, int LAYOUT> struct CgMap {
  typedef CgSwz<GROUP> Z;
  static constexpr u32 PADE = 16 / sizeof(E);
  TN_HD static u32 at(u32 x) {
    if (LAYOUT == CG_PADDED) return x + (x / Z::CH) * PADE + (Z::CH2 ? (x / (Z::CH2 ? Z::CH2 : 1u)) * PADE : 0u);
    if (LAYOUT == CG_SWIZZLED) return x ^ (((x >> Z::S1) & Z::M1) << 1) ^ (((x >> Z::S2) & Z::M2) << 1) ^ (((x >> Z::S3) & Z::M3) << 1);
    return x;
  }
  TN_HD static constexpr u32 span(u32 n) { return LAYOUT == CG_PADDED ? n + (n / Z::CH) * PADE + (Z::CH2 ? (n / (Z::CH2 ? Z::CH2 : 1u)) * PADE : 0u) : n; }
  // at(x + d) from at(x) for x a multiple of R, d < R: padding chunks are multiples of R; the swizzle only XORs bits >= 1
  // with functions of bits >= log2 R
  TN_HD static u32 step(u32 ax, u32 d) { return LAYOUT == CG_SWIZZLED ? (ax ^ d) : (ax + d); }
  // at(T + (e << cs)) from aT = at(T) for T < 2^cs, 2^cs a multiple of every padding chunk (n >= 4096 here): the padding is
  // additive in the column, and the swizzle is GF(2)-linear (at(T ^ C) = at(T) ^ at(C)) with T + C = T ^ C for a column start C, so a column write
  // is one address register (XORed with a per-column constant for the swizzle) plus an immediate
  TN_HD static u32 col(u32 aT, u32 e, u32 cs) {
    const u32 c0 = e << cs;
    if (LAYOUT == CG_PADDED) return aT + at(c0);
    if (LAYOUT == CG_SWIZZLED) return (aT ^ (at(c0) ^ c0)) + c0;
    return aT + c0;
  }
};
// The twiddle table omega^j, j <= n/2, staged in LDS: record j at twmap(j).  XOR-swizzled for the padded and the swizzled
// image alike: the last trip reads it at strides G, G/2, .., 1 records across lanes, and 16-byte records at a power-of-two
// stride hit the same banks (conflict free forwards; read backwards for the inverse a 2-way conflict remains).
template <int GROUP, int LAYOUT> TN_HD u32 cg_twmap(u32 j, bool big) {
  return (LAYOUT != CG_LINEAR && big) ? (j ^ ((j >> CgSwz<GROUP>::TS) & CgSwz<GROUP>::TM)) : j;
}

// two neighbouring coefficients (2i, 2i+1): one 16-byte (8-byte for 32-bit lanes) LDS access in every layout
template <typename E> struct alignas(2 * sizeof(E)) CgPair { E lo, hi; };

template <typename E, int AM> struct CgArith {
  typedef typename TwOf<E>::type Tw;
  static constexpr bool SPLIT = AM != CGA_SHOUP, LAZY = AM == CGA_SPLIT_LAZY || AM == CGA_SPLIT_SCHED, SCHED = AM == CGA_SPLIT_SCHED;
  typedef Policy<E, SPLIT> P;
  // a * w for ANY word a (twist, cg_ntt.py:82-83) / a mod q (the implicit % of :55-58).  Canonical, or (lazy) only congruent and
  // below the lazy butterflies' input bound: the bare split-constant product, or one fold.
  TN_HD static E in_mul(E a, Tw w, const Arith<E>& ar) {
    if constexpr (LAZY) return mul_sp(a, w, ar.sk);
    else if constexpr (SPLIT) return P::mul_tw_canon(a, w, ar);
    else return mul_tw(a, w, ar.q);
  }
  TN_HD static E in_red(E a, const Arith<E>& ar) {
    if constexpr (LAZY) return fold(a, ar.k, ar.fold_c);
    else if constexpr (SPLIT) return P::canon(a, ar);
    else return mul_tw(a, ar.one, ar.q);
  }
  // The reference butterfly (cg_ntt.py:57-59): t = omega * right, (left + t) % q, (left - t) % q.
  //   Shoup: product and compare-select (any modulus).
  //   split, canonical: the product rides the multiply-add chain as left + t' with t' < 5q, the difference is
  //     left + 5q - t'; one fold (-> below 2q) and one conditional subtraction make each canonical.
  //   split, lazy: left is folded (< 2^k + eps), the difference is 2 left + 6q - x.  With every input below 7.01 * 2^k (true
  //     for canonical inputs and preserved: t' < 4 * 2^k + 7.01 * 2^k / 8 + eps < 4.9 * 2^k <= 6q) both outputs stay below
  //     7.01 * 2^k; h_cg_lazy_ok() replays these bounds exactly for the plan's (k, c).
  //   split, scheduled (n = 4096 compiled in): values of an even stage's inputs are below 12.01 * 2^k, of an odd stage's below
  //     7.01 * 2^k (the twisted / folded / pointwise inputs of stage 1 too); odd stages (STAGE_EVEN false) run on the raw left
  //     input with the difference 2 left + 5q - x (t' < 4.9 * 2^k <= 5q, outputs below 12.01 * 2^k), even stages fold left first
  //     and use 6q (t' < 5.51 * 2^k <= 6q, outputs below 7.01 * 2^k): half the folds.  h_cg_sched_ok() replays this exactly.
  template <bool STAGE_EVEN> TN_HD static void bf(E left, E right, Tw w, const Arith<E>& ar, E& sum, E& dif) {
    if constexpr (SCHED) {
      const u64 u = STAGE_EVEN ? fold(left, ar.k, ar.fold_c) : left;
      const u64 x = mul_sp_acc(u, right, w, ar.sk);
      dif = ((u << 1) + ar.qmul[STAGE_EVEN ? 6 : 5]) - x;
      sum = x;
    } else if constexpr (LAZY) {
      const u64 u = fold(left, ar.k, ar.fold_c);
      const u64 x = mul_sp_acc(u, right, w, ar.sk);
      dif = ((u << 1) + ar.qmul[6]) - x;
      sum = x;
    } else if constexpr (SPLIT) {
      const u64 x = mul_sp_acc(left, right, w, ar.sk);               // left < q, right < q
      const u64 y = ((left << 1) + ar.qmul[5]) - x;                  // left + 5q - t'
      sum = csub(fold(x, ar.k, ar.fold_c), ar.q);
      dif = csub(fold(y, ar.k, ar.fold_c), ar.q);
    } else {
      const E t = mul_tw(right, w, ar.q);                            // :57
      sum = csub((E)(left + t), ar.q);                               // :58
      dif = left >= t ? (E)(left - t) : (E)(left + (ar.q - t));      // :59
    }
  }
  // Stage 1 of a TWISTED transform, lazy modes only (TN_CG_MERGE_TWIST): every twiddle of stage 1 is omega^0 = 1 (cg_ntt.py:51,:54
  // with i < k), so the butterfly's product 1 * a'[2i+1] can BE the twist multiplication a[j + n/2] * psi^(j + n/2) of its
  // right input (:82-83): w = that record, the right input arrives as the raw word (any 64-bit value), the left input twisted as
  // before.  Same residues, half the twist multiplications.  t' < tmax(2^64) (slightly above 6 * 2^k), hence K = 7; the outputs
  // (below 13.01 * 2^k) are admissible inputs of stage 2 in both lazy schedules (h_cg_lazy_ok, h_cg_sched_ok replay this).
  TN_HD static void bf_first(E left, E right_raw, Tw w, const Arith<E>& ar, E& sum, E& dif) {
    static_assert(LAZY, "merged twist: lazy arithmetic only");
    const u64 u = SCHED ? (u64)left : fold(left, ar.k, ar.fold_c);
    const u64 x = mul_sp_acc(u, right_raw, w, ar.sk);
    dif = ((u << 1) + ar.qmul[7]) - x;
    sum = x;
  }
  // A^[i] * B^[i] (cg_ntt.py:88): canonical Barrett product of canonical values, or the fused kernels' lazy product (< 2q)
  TN_HD static E pointwise(E a, E b, const Arith<E>& ar) {
    if constexpr (LAZY) return pointwise_lazy(a, b, ar);
    else return mulmod_barrett(a, b, ar.q, ar.mu, ar.k);
  }
  // canonical x * w (untwist and n^-1, cg_ntt.py:74-75,:92) / canonical x
  TN_HD static E out_mul(E x, Tw w, const Arith<E>& ar) {
    if constexpr (SPLIT) return P::mul_tw_canon(x, w, ar);
    else return mul_tw(x, w, ar.q);
  }
  TN_HD static E out_canon(E x, const Arith<E>& ar) {
    if constexpr (LAZY) return P::canon(x, ar);
    else return x;
  }
};

// One trip: NST <= L stages on the R registers of one lane-step.
//   tw(j, h)  the record of stage j (0-based within the trip) for the butterflies whose top j bits are h
//   SWAP      the record is MINUS the wanted twiddle (inverse transform on the forward table read backwards:
//             omega^-i = -omega^(n/2 - i)), so the two outputs change places
//   after(j)  called after stage j with the registers in their new places (per-stage trace)
//   S0PAR     parity of the number of stages done before this trip (the scheduled arithmetic folds on even stages)
//   MERGE0    stage 0 is the merged twist + butterfly (CgArith::bf_first): w0(g) = the twist record of butterfly g's right input
struct CgNoW0 { template <typename G_> TN_HD int operator()(G_) const { return 0; } };
template <typename E, int GROUP, int AM, int NST, bool SWAP, int S0PAR = 0, bool MERGE0 = false, typename TW, typename AFTER, typename W0 = CgNoW0>
TN_HD void cg_trip(E (&x)[2 * GROUP], const Arith<E>& ar, TW&& tw, AFTER&& after, W0&& w0 = W0()) {
  constexpr int G = GROUP, L = CgGeom<G>::L;
  typedef typename TwOf<E>::type Tw;
  static_for<0, NST>([&](auto j_) {
    constexpr int j = decltype(j_)::value;
    Tw w[1 << j];
    if constexpr (!(MERGE0 && j == 0)) static_for<0, (1 << j)>([&](auto h_) { w[decltype(h_)::value] = tw(j_, h_); });
    E z[2 * G];
    static_for<0, G>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      constexpr bool EVEN = ((S0PAR + j + 1) & 1) == 0;                 // stage number s0 + j + 1 (cg_ntt.py:49)
      if constexpr (MERGE0 && j == 0) CgArith<E, AM>::bf_first(x[2 * g], x[2 * g + 1], w0(g_), ar, z[g], z[G + g]);
      else if constexpr (SWAP) CgArith<E, AM>::template bf<EVEN>(x[2 * g], x[2 * g + 1], w[g >> (L - 1 - j)], ar, z[G + g], z[g]);
      else CgArith<E, AM>::template bf<EVEN>(x[2 * g], x[2 * g + 1], w[g >> (L - 1 - j)], ar, z[g], z[G + g]);
      if constexpr (TN_CG_FENCE > 0 && ((g + 1) % (TN_CG_FENCE > 0 ? TN_CG_FENCE : 1)) == 0 && g + 1 < G) sched_fence_valu();   // bounds the butterflies in flight (live temporaries)
    });
#pragma unroll
    for (int e = 0; e < 2 * G; ++e) x[e] = z[e];
    after(j_);
  });
}

// Index into the table omega^i of the record stage j of a trip needs (cg_ntt.py:51,:54: omega^(k (i // k)), k = n >> s):
//   h (n >> (j+1)) + ((T << (L-1-j)) & ~(k - 1)),   k = n >> (s0 + j + 1)
// for lane-step T of a trip that starts after s0 stages; (T << (L-1)) & ~((n >> (s0+1)) - 1) is computed once per trip
// (base0) and shifted per stage.  First trip: s0 = 0 and T << (L-1) < n/2, so base0 = 0 whatever T is.
template <int GROUP>
TN_HD u32 cg_tw_base0(u32 logn, u32 s0, u32 T) {
  return (T << (CgGeom<GROUP>::L - 1)) & ~((((u32)1 << logn) >> (s0 + 1)) - 1u);
}

}  // namespace tn
