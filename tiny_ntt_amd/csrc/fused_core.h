// fused_core.h — the per-thread body of the fused negacyclic poly-mult kernel
// (K6 of SURVEY.md §2), written as host+device templates so the exact same
// index math and arithmetic can be (a) compiled into the gfx950 kernel and
// (b) stepped thread-by-thread on the CPU by tests/emu (no GPU in the build
// container).  Nothing here touches the oracle.
//
// What it computes: c = a*b in Z_q[x]/(x^n+1), bit-identical to
// nwc_poly_mult (new_reference/cg_ntt.py:78-92).  How: the psi-twist of
// cg_ntt.py:82-83/:92 is folded into the twiddles (a length-n negacyclic
// transform is the radix-2 splitting of x^n+1 with twiddles psi^brv(m+i)), so
// the forward pass is log2(n) Cooley-Tukey stages with NO separate twist pass,
// the inverse is the mirrored Gentleman-Sande pass, and n^-1 is folded into the
// last inverse stage.  All arithmetic is exact mod q, so the result equals the
// reference's twist -> cg_ntt -> pointwise -> cg_intt -> untwist chain.
//
// Work decomposition for one polynomial of n = 2^LOGN coefficients:
//   THREADS = n >> LPT threads, each holding R = 2^LPT coefficients in VGPRs.
//   The LOGN stages run in PHASES = ceil(LOGN/LPT) register-resident phases;
//   between phases the coefficients are transposed through LDS (one padded,
//   bank-conflict-free image; see ex_layout()).
//   Phase p: register index r <-> bits [pos(p), pos(p)+LPT) of the coefficient
//   index j, thread index tau <-> the remaining bits.
#pragma once
#include <type_traits>
#include "modarith.h"

namespace tn {

template <int B, int E_, typename F> struct StaticFor {
  TN_HD static void run(F& f) { f(std::integral_constant<int, B>()); StaticFor<B + 1, E_, F>::run(f); }
};
template <int E_, typename F> struct StaticFor<E_, E_, F> { TN_HD static void run(F&) {} };
template <int B, int E_, typename F> TN_HD void static_for(F&& f) { StaticFor<B, E_, F>::run(f); }

// Per-plan arithmetic constants, passed by value to kernels (lives in SGPRs).
template <typename E> struct Arith {
  E q;
  u64 mu;        // floor(2^(2k)/q)
  int k;         // bitlen(q)
  u32 fold_c;    // 2^k - q (lazy policies only)
  typename TwOf<E>::type one;        // w = 1 (used to canonicalise arbitrary inputs)
  typename TwOf<E>::type ninv;       // n^-1
  typename TwOf<E>::type ninv_w1;    // n^-1 * psi_inv_brv[1]  (last inverse stage, odd half)
};

template <typename E, int LOGN_, int LPT_> struct FusedCfg {
  static constexpr int LOGN = LOGN_, LPT = LPT_;
  static constexpr int N = 1 << LOGN, R = 1 << LPT, THREADS = N >> LPT;
  static constexpr int PHASES = (LOGN + LPT - 1) / LPT;
  static constexpr int pos(int p) { return (LOGN - (p + 1) * LPT) < 0 ? 0 : (LOGN - (p + 1) * LPT); }
  static constexpr int stage_begin(int p) { return p * LPT; }
  static constexpr int stage_end(int p) { return ((p + 1) * LPT) < LOGN ? ((p + 1) * LPT) : LOGN; }
  // LDS image used between phase e and e+1:  addr(j) = j + PAD * (j >> SH)   (in elements)
  static constexpr int ex_pad(int e) { return pos(e + 1) > 0 ? (1 << pos(e + 1)) : (int)(16 / sizeof(E)); }
  static constexpr int ex_sh(int e) { return pos(e + 1) > 0 ? pos(e + 1) + LPT : LPT; }
  static constexpr int lay_span(int e, int cnt) { return cnt + ex_pad(e) * (cnt >> ex_sh(e)); }
  // Waves: thread-id bits >= 6.  WB of them; in every phase after the first they are the top
  // WB bits of the coefficient index, so a wave owns one contiguous slice of 2^WSH coefficients.
  static constexpr int WB = (LOGN - LPT > 6) ? (LOGN - LPT - 6) : 0;
  static constexpr int WSH = LOGN - WB;
  // j-bit position that thread-id bit t maps to in phase p
  static constexpr int tau_bit_to_j(int p, int t) { return t < pos(p) ? t : t + LPT; }
  // An exchange between phases e and e+1 stays inside each 64-lane wave iff in both phases the
  // wave index is made of the top WB coefficient-index bits; then no workgroup barrier is needed
  // (a wave's LDS operations execute in issue order) provided each wave uses a private LDS region.
  static constexpr bool ex_wave_local(int e) {
    for (int t = 6; t < LOGN - LPT; ++t)
      if (tau_bit_to_j(e, t) != t + LPT || tau_bit_to_j(e + 1, t) != t + LPT) return false;
    return true;
  }
  static constexpr int region_elems() {            // per-wave LDS region shared by all wave-local exchanges
    int m = 1 << WSH;
    for (int e = 0; e + 1 < PHASES; ++e)
      if (ex_wave_local(e) && lay_span(e, 1 << WSH) > m) m = lay_span(e, 1 << WSH);
    return m;
  }
  static constexpr int lds_elems() {
    int m = region_elems() << WB;
    for (int e = 0; e + 1 < PHASES; ++e)
      if (!ex_wave_local(e) && lay_span(e, N) > m) m = lay_span(e, N);
    return m;
  }
  TN_HD static u32 jidx(int p, u32 tau, u32 r) {
    const int ps = pos(p);
    return ((tau >> ps) << (ps + LPT)) | (r << ps) | (tau & ((1u << ps) - 1u));
  }
  TN_HD static u32 ex_addr(int e, u32 j) {
    if (WB > 0 && ex_wave_local(e)) {
      const u32 x = j & ((1u << WSH) - 1u);
      return (j >> WSH) * (u32)region_elems() + x + (u32)ex_pad(e) * (x >> ex_sh(e));
    }
    return j + (u32)ex_pad(e) * (j >> ex_sh(e));
  }
  // the part of the twiddle index that comes from the thread id; a compile-time 0 where every
  // thread of the workgroup shares the twiddles (phase 0), so those loads become scalar loads
  TN_HD static u32 thi(int p, u32 tau) { return pos(p) >= LOGN - LPT ? 0u : (tau >> pos(p)); }
};

// ---------------------------------------------------------------------------
// Arithmetic policies.  LIMIT = how many multiples of q a lane word can hold.
//
// Lazy: values are kept only congruent mod q and bounded by a compile-time
// multiple of q; "fold" (one Barrett step with estimate x>>k) is inserted by a
// static schedule when the bound would exceed LIMIT.  Needs q = 2^k - c with c
// small and LIMIT*q <= 2^W; checked at plan creation (plan.cpp).
// Canonical: every value in [0,q) after every operation; any odd q < 2^62 / 2^31.
template <typename E> struct LazyTraits;
template <> struct LazyTraits<u64> { static constexpr int LIMIT = 16, TMUL = 4; };   // mul_tw_lazy < 4q
template <> struct LazyTraits<u32> { static constexpr int LIMIT = 64, TMUL = 2; };   // mul_tw_lazy < 2q

// Lazy Cooley-Tukey butterfly: u' = u + t, v' = u - t + TMUL*q with t = v*w mod q + {0..TMUL-1}q.
// 64-bit lanes: the add of u rides on the mad chain for free, and v' = 2u + 4q - u'
// (computed mod 2^64; the true value u + 4q - t fits by the lazy bound).
TN_HD void ct_lazy(u64& u, u64& v, Tw64 w, u64 q) {
  const u64 x = mul_tw_acc(u, v, w, q);
  v = ((u << 1) + 4 * q) - x;
  u = x;
}
TN_HD void ct_lazy(u32& u, u32& v, Tw32 w, u32 q) {
  const u32 t = mul_tw_lazy(v, w, q);                           // < 2q
  v = u + (2 * q - t);
  u = u + t;
}

template <typename E, bool LAZY> struct Policy {
  typedef typename TwOf<E>::type Tw;
  static constexpr bool lazy = LAZY;
  static constexpr int LIMIT = LazyTraits<E>::LIMIT, TMUL = LazyTraits<E>::TMUL;

  // value bound (in multiples of q) after loading an arbitrary word
  TN_HD static E load(E x, const Arith<E>& ar) {
    if (LAZY) return fold(x, ar.k, ar.fold_c);                 // < 2q
    return mul_tw(x, ar.one, ar.q);                            // canonical
  }
  TN_HD static E canon(E x, const Arith<E>& ar) {              // any bounded lazy value -> [0,q)
    if (LAZY) { x = fold(x, ar.k, ar.fold_c); return csub(x, ar.q); }
    return x;
  }
  // Cooley-Tukey: (u, v) -> (u + w v, u - w v)
  TN_HD static void ct(E& u, E& v, Tw w, const Arith<E>& ar) {
    if (LAZY) {
      ct_lazy(u, v, w, ar.q);
    } else {
      E t = mul_tw(v, w, ar.q);
      E s = u + t;
      v = u >= t ? u - t : u + (ar.q - t);
      u = csub(s, ar.q);
    }
  }
  // Gentleman-Sande: (u, v) -> (u + v, (u - v) w);  BND = compile-time bound of u, v
  template <int BND> TN_HD static void gs(E& u, E& v, Tw w, const Arith<E>& ar) {
    if (LAZY) {
      E d = u + ((E)BND * ar.q - v);
      u = u + v;
      v = mul_tw_lazy(d, w, ar.q);
    } else {
      E d = u >= v ? u - v : u + (ar.q - v);
      u = csub(u + v, ar.q);
      v = mul_tw(d, w, ar.q);
    }
  }
  // last inverse stage, n^-1 folded in, canonical outputs
  template <int BND> TN_HD static void gs_last(E& u, E& v, const Arith<E>& ar) {
    E d = LAZY ? (E)(u + ((E)BND * ar.q - v)) : (u >= v ? (E)(u - v) : (E)(u + (ar.q - v)));
    E s = u + v;                                               // < 2 BND q (lazy) or < 2q: fits the word
    u = mul_tw(s, ar.ninv, ar.q);
    v = mul_tw(d, ar.ninv_w1, ar.q);
  }
};

// Static fold schedule.  fwd: bound grows by TMUL per stage.  inv: bound -> max(2B, TMUL).
template <typename P, int LOGN> struct Sched {
  // bound BEFORE forward stage s (after an optional fold)
  static constexpr int fwd_in(int s) {
    int b = 2;                                   // after load()
    for (int i = 0; i < s; ++i) { if (b + P::TMUL > P::LIMIT) b = 2; b += P::TMUL; }
    return b;
  }
  static constexpr bool fwd_fold(int s) { return P::lazy && fwd_in(s) + P::TMUL > P::LIMIT; }
  static constexpr int fwd_out() {
    int b = 2;
    for (int i = 0; i < LOGN; ++i) { if (b + P::TMUL > P::LIMIT) b = 2; b += P::TMUL; }
    return b;
  }
  // inverse stage index g = 0 .. LOGN-1 in execution order (g = 0 is distance 1)
  static constexpr int inv_in(int g) {
    int b = 1;                                   // pointwise output is canonical
    for (int i = 0; i < g; ++i) { if (2 * b > P::LIMIT) b = 2; b = (2 * b > P::TMUL) ? 2 * b : P::TMUL; }
    return b;
  }
  static constexpr bool inv_fold(int g) { return P::lazy && 2 * inv_in(g) > P::LIMIT; }
  static constexpr int inv_bnd(int g) { return inv_fold(g) ? 2 : inv_in(g); }
};

// ---------------------------------------------------------------------------
// One forward phase on a thread's registers.
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void fwd_phase(E (&x)[Cfg::R], u32 tau, const typename TwOf<E>::type* __restrict__ tw, const Arith<E>& ar) {
  typedef Sched<Pol, Cfg::LOGN> S;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    if (S::fwd_fold(s)) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = fold(x[r], ar.k, ar.fold_c);
    }
    const u32 base = (1u << s) + (thi << (Cfg::LPT - bpos - 1));
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) {
      if (r & (1 << bpos)) continue;
      Pol::ct(x[r], x[r | (1 << bpos)], tw[base + (r >> (bpos + 1))], ar);
    }
  });
}

// The same forward phase on TWO polynomials at once (a and b of one product): every twiddle
// is loaded once and used for both butterflies, and the two chains interleave.
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void fwd_phase_pair(E (&x)[Cfg::R], E (&y)[Cfg::R], u32 tau, const typename TwOf<E>::type* __restrict__ tw,
                          const Arith<E>& ar) {
  typedef Sched<Pol, Cfg::LOGN> S;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    if (S::fwd_fold(s)) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) { x[r] = fold(x[r], ar.k, ar.fold_c); y[r] = fold(y[r], ar.k, ar.fold_c); }
    }
    const u32 base = (1u << s) + (thi << (Cfg::LPT - bpos - 1));
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) {
      if (r & (1 << bpos)) continue;
      const typename TwOf<E>::type w = tw[base + (r >> (bpos + 1))];
      Pol::ct(x[r], x[r | (1 << bpos)], w, ar);
      Pol::ct(y[r], y[r | (1 << bpos)], w, ar);
    }
  });
}

// One inverse phase (stages of phase PH in reverse order).
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void inv_phase(E (&x)[Cfg::R], u32 tau, const typename TwOf<E>::type* __restrict__ tw, const Arith<E>& ar) {
  typedef Sched<Pol, Cfg::LOGN> S;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<0, Cfg::stage_end(PH) - Cfg::stage_begin(PH)>([&](auto i_) {
    constexpr int s = Cfg::stage_end(PH) - 1 - decltype(i_)::value;   // forward stage number being undone
    constexpr int g = Cfg::LOGN - 1 - s;                               // execution order of the inverse
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    if (S::inv_fold(g)) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = fold(x[r], ar.k, ar.fold_c);
    }
    constexpr int BND = S::inv_bnd(g);
    const u32 base = (1u << s) + (thi << (Cfg::LPT - bpos - 1));
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) {
      if (r & (1 << bpos)) continue;
      if (s == 0) Pol::template gs_last<BND>(x[r], x[r | (1 << bpos)], ar);
      else Pol::template gs<BND>(x[r], x[r | (1 << bpos)], tw[base + (r >> (bpos + 1))], ar);
    }
  });
}

// LDS transposes.  e = exchange index (between phase e and e+1); `from` = phase
// whose register layout is being written, `to` = phase whose layout is read.
template <typename E, typename Cfg, int EX, int PH>
TN_HD void ex_store(const E (&x)[Cfg::R], u32 tau, E* lds) {
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r) lds[Cfg::ex_addr(EX, Cfg::jidx(PH, tau, r))] = x[r];
}
template <typename E, typename Cfg, int EX, int PH>
TN_HD void ex_load(E (&x)[Cfg::R], u32 tau, const E* lds) {
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r) x[r] = lds[Cfg::ex_addr(EX, Cfg::jidx(PH, tau, r))];
}

// Pointwise product in the last phase's register layout: canonical result.
template <typename E, typename Cfg, typename Pol>
TN_HD void pointwise(E (&xa)[Cfg::R], const E (&xb)[Cfg::R], const Arith<E>& ar) {
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r)
    xa[r] = mulmod_barrett(Pol::canon(xa[r], ar), Pol::canon(xb[r], ar), ar.q, ar.mu, ar.k);
}

}  // namespace tn
