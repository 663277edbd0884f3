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

// Timing-ablation switches (developer experiments only; results are wrong when set).
#ifndef TN_BFLY_FENCE
#define TN_BFLY_FENCE 0          // >0: scheduling fence after every TN_BFLY_FENCE twiddle groups of a stage
#endif
#ifndef TN_PREFETCH_LAST
#define TN_PREFETCH_LAST 2       // last-phase (thread-private) twiddles: 0 = loaded at use; 1 = one stage early (28 VGPRs live
                                 // through a butterfly stage: spills at 128); 2 = just before the transpose that precedes the phase
#endif
#ifndef TN_TW_AHEAD
#define TN_TW_AHEAD 0            // scalar (wave-uniform) twiddles: 1 = requested one butterfly stage before their use, so the
                                 // scalar-cache / L2 latency hides behind a stage of arithmetic; 0 = loaded at use.
                                 // Measured on MI355X (n=4096, 60-bit): 1 is 2% SLOWER (3.28 vs 3.21 ms) - the kernel runs
                                 // at the package power cap, where hidden stalls buy nothing and the fences cost ILP.
#endif
#ifndef TN_ABL_UNIFORM_TW
#define TN_ABL_UNIFORM_TW 0      // 1: every thread uses the phase-0 (wave-uniform) twiddle indices -> no vector twiddle loads
#endif

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
  // n = 4096, 64-bit, 8 coefficients per thread: XOR-swizzled images instead of padded ones, so the
  // transpose buffer is exactly one polynomial (32 KiB) and two workgroups per CU fit beside the
  // parked operand and the LDS twiddle tables.  With j = (w:3 | g:3 | m:3 | e:3):
  //   exchange 0 (phase 0 <-> 1): both sides touch 64 contiguous coefficients per wave-instruction: identity;
  //   exchange 1 (1 <-> 2): m[1:0] ^= g[1:0];
  //   exchange 2 (2 <-> 3): m[1:0] ^= g[1:0], e[2] ^= m[2] ^ g[1], e[1] ^= m[1]
  // (bank-conflict free for the ds_{read,write}_b64 / _b128 lane groups of gfx950; checked by
  // tests/test_lds_banks.py with a bank simulator).
  static constexpr bool SWZ = (sizeof(E) == 8 && LOGN == 12 && LPT == 3);
  static constexpr int lds_elems() {
    if (SWZ) return N;
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
    if (SWZ) {
      if (e == 0) return j;
      const u32 g = (j >> 6) & 7u, m = (j >> 3) & 7u;
      u32 a = j ^ ((g & 3u) << 3);
      if (e == 2) a ^= ((((m >> 2) ^ (g >> 1)) & 1u) << 2) | (((m >> 1) & 1u) << 1);
      return a;
    }
    if (WB > 0 && ex_wave_local(e)) {
      const u32 x = j & ((1u << WSH) - 1u);
      return (j >> WSH) * (u32)region_elems() + x + (u32)ex_pad(e) * (x >> ex_sh(e));
    }
    return j + (u32)ex_pad(e) * (j >> ex_sh(e));
  }
  // Where the twiddles of phase p come from:
  //   TW_UNIFORM  every thread of the workgroup uses the same ones        -> scalar loads
  //   TW_WAVE     the same within a wave (index depends on the wave id)     -> scalar loads
  //   TW_LDS      lane-dependent, table small: staged once per workgroup in LDS
  //   TW_REGS     last phase: one private set per thread, fetched from the L2-resident table
  //               one phase ahead into registers (pre[])
  enum { TW_UNIFORM = 0, TW_WAVE = 1, TW_LDS = 2, TW_REGS = 3 };
  static constexpr int tw_src(int p) {
    return pos(p) >= LOGN - LPT ? TW_UNIFORM : (pos(p) >= 6 ? TW_WAVE : (pos(p) > 0 ? TW_LDS : (TN_PREFETCH_LAST ? TW_REGS : TW_WAVE)));
  }
  static constexpr int lds_tw_lo() {                 // first table index kept in LDS
    for (int p = 0; p < PHASES; ++p) if (tw_src(p) == TW_LDS) return 1 << stage_begin(p);
    return 0;
  }
  static constexpr int lds_tw_hi() {                 // one past the last
    int h = 0;
    for (int p = 0; p < PHASES; ++p) if (tw_src(p) == TW_LDS) h = 1 << stage_end(p);
    return h;
  }
  static constexpr int lds_tw_count() { return lds_tw_hi() - lds_tw_lo(); }
  // private twiddles of the last phase: stage s contributes R >> (bpos+1) of them
  static constexpr int pre_count(int s) { return R >> ((LOGN - 1 - s) - pos(PHASES - 1) + 1); }
  static constexpr int pre_off(int s) {
    int o = 0;
    for (int i = stage_begin(PHASES - 1); i < s; ++i) o += pre_count(i);
    return o;
  }
  static constexpr int NPRE = (tw_src(PHASES - 1) == TW_REGS) ? pre_off(LOGN) : 1;
  // LDS image in NATURAL coefficient order, used by the standalone transforms to move between HBM order and the
  // last phase's bit-reversed register layout.  Consecutive lanes of that phase hit indices that differ in their
  // HIGH bits (bit-reversed thread id), so bits [5,9) are XORed into the low bits to spread them over the banks.
  TN_HD static u32 nat_addr(u32 k) { return k ^ ((k >> 5) & 15u); }
  // the part of the twiddle index that comes from the thread id; a compile-time 0 where every
  // thread of the workgroup shares the twiddles (phase 0), so those loads become scalar loads
  // ... and wave-uniform (made a scalar) where it only depends on the wave index (pos(p) >= 6).
  TN_HD static u32 thi(int p, u32 tau) {
    if (pos(p) >= LOGN - LPT) return 0u;
    if (pos(p) >= 6) return wave_uniform(tau >> pos(p));
    return tau >> pos(p);
  }
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
// PW = bound of the lazy pointwise product (pointwise() below)
template <> struct LazyTraits<u64> { static constexpr int LIMIT = 16, TMUL = 4, PW = 2; };   // mul_tw_lazy < 4q
template <> struct LazyTraits<u32> { static constexpr int LIMIT = 64, TMUL = 2, PW = 4; };   // mul_tw_lazy < 2q

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
  static constexpr int LIMIT = LazyTraits<E>::LIMIT, TMUL = LazyTraits<E>::TMUL, PW = LazyTraits<E>::PW;

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
    u = mul_tw_canon(s, ar.ninv, ar);
    v = mul_tw_canon(d, ar.ninv_w1, ar);
  }
  // canonical twiddle product.  Lazy 64-bit lanes: the product lies in [0, 4q); one fold puts it below 2q,
  // so ONE conditional subtraction finishes instead of two (24 instead of 32 issue cycles).
  TN_HD static E mul_tw_canon(E a, Tw w, const Arith<E>& ar) {
    if (LAZY && sizeof(E) == 8) return csub(fold(mul_tw_lazy(a, w, ar.q), ar.k, ar.fold_c), ar.q);
    return mul_tw(a, w, ar.q);
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
    int b = P::lazy ? P::PW : 1;                 // pointwise output: lazy < PW q, else canonical
    for (int i = 0; i < g; ++i) { if (2 * b > P::LIMIT) b = 2; b = (2 * b > P::TMUL) ? 2 * b : P::TMUL; }
    return b;
  }
  static constexpr bool inv_fold(int g) { return P::lazy && 2 * inv_in(g) > P::LIMIT; }
  static constexpr int inv_bnd(int g) { return inv_fold(g) ? 2 : inv_in(g); }
};

// The three places a phase can take its twiddles from (see FusedCfg::tw_src).
template <typename E> struct TwRefs {
  typedef typename TwOf<E>::type Tw;
  const Tw* __restrict__ glob;     // full table psi^brv(i) (or inverse) in global memory / L2
  const Tw* lds;                   // entries [lds_tw_lo, lds_tw_hi) of it, staged in LDS
  Tw* pre;                         // the calling thread's last-phase twiddles, in registers
  const Tw* mid = nullptr;         // optional: the calling thread's twiddles of the LDS-sourced phase, already in registers
};

template <typename E, typename Cfg, int PH, int S_>
TN_HD typename TwOf<E>::type tw_get(const TwRefs<E>& t, u32 thi, int g) {
  constexpr int bpos = (Cfg::LOGN - 1 - S_) - Cfg::pos(PH);
  if (Cfg::tw_src(PH) == Cfg::TW_REGS) return t.pre[Cfg::pre_off(S_) + g];
  const u32 idx = (1u << S_) + (TN_ABL_UNIFORM_TW ? 0u : (thi << (Cfg::LPT - bpos - 1))) + (u32)g;
  if (Cfg::tw_src(PH) == Cfg::TW_LDS) {
    if (t.mid) return t.mid[(1 << (S_ - Cfg::stage_begin(PH))) - 1 + g];      // stage i of a full phase has 2^i twiddles, starting at 2^i - 1
    return t.lds[idx - Cfg::lds_tw_lo()];
  }
  return t.glob[idx];
}

// The calling thread's twiddles of an LDS-sourced FULL phase PH (LPT stages: 1 + 2 + ... = R - 1 of them), read into
// registers once so that two transforms can run that phase on one fetch.
template <typename E, typename Cfg, int PH>
TN_HD void tw_fetch_mid(typename TwOf<E>::type (&mid)[Cfg::R], u32 tau, const typename TwOf<E>::type* lds_tw) {
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    constexpr int cnt = Cfg::R >> (bpos + 1), off = (1 << (s - Cfg::stage_begin(PH))) - 1;
#pragma unroll
    for (int g = 0; g < cnt; ++g)
      mid[off + g] = lds_tw[(1u << s) + (thi << (Cfg::LPT - bpos - 1)) + (u32)g - Cfg::lds_tw_lo()];
  });
}

// Phases whose twiddles are wave-uniform (scalar loads): they are requested one stage ahead (TN_TW_AHEAD).
template <typename Cfg, int PH> constexpr bool tw_ahead() {
  return TN_TW_AHEAD && (Cfg::tw_src(PH) == Cfg::TW_UNIFORM || Cfg::tw_src(PH) == Cfg::TW_WAVE);
}
// All twiddles one thread uses in stage S_ of phase PH: R >> (bpos + 1) of them.
template <typename E, typename Cfg, int PH, int S_>
TN_HD void tw_stage(const TwRefs<E>& t, u32 tau, typename TwOf<E>::type (&w)[Cfg::R / 2]) {
  constexpr int bpos = (Cfg::LOGN - 1 - S_) - Cfg::pos(PH);
  const u32 thi = Cfg::thi(PH, tau);
#pragma unroll
  for (int g = 0; g < (Cfg::R >> (bpos + 1)); ++g) w[g] = tw_get<E, Cfg, PH, S_>(t, thi, g);
}

// Fetch the calling thread's last-phase twiddles into registers (issued ahead of their use).
template <typename E, typename Cfg>
TN_HD void tw_prefetch_raw(typename TwOf<E>::type* pre, u32 tau, const typename TwOf<E>::type* __restrict__ glob) {
  constexpr int PH = Cfg::PHASES - 1;
  if (Cfg::tw_src(PH) != Cfg::TW_REGS) return;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
#pragma unroll
    for (int g = 0; g < Cfg::pre_count(s); ++g)
      pre[Cfg::pre_off(s) + g] = glob[(1u << s) + (thi << (Cfg::LPT - bpos - 1)) + (u32)g];
  });
}
template <typename E, typename Cfg>
TN_HD void tw_prefetch(typename TwOf<E>::type (&pre)[Cfg::NPRE], u32 tau, const typename TwOf<E>::type* __restrict__ glob) {
  tw_prefetch_raw<E, Cfg>(pre, tau, glob);
}

// ---------------------------------------------------------------------------
// One forward phase on a thread's registers.
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void fwd_phase(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw, const Arith<E>& ar, typename TwOf<E>::type (&cur)[Cfg::R / 2]) {
  // cur[]: with tw_ahead<PH>, the first stage's twiddles, already requested by the caller (before the transpose
  // that precedes this phase); each stage then requests the next stage's before its own butterflies.
  typedef Sched<Pol, Cfg::LOGN> S;
  typedef typename TwOf<E>::type Tw;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    // the next (last) phase's thread-private twiddles are requested from L2 one stage early
    if (TN_PREFETCH_LAST == 1 && Cfg::PHASES >= 2 && PH == Cfg::PHASES - 2 && s == Cfg::stage_end(PH) - 1 &&
        Cfg::tw_src(Cfg::PHASES - 1) == Cfg::TW_REGS) {
      sched_fence();
      tw_prefetch_raw<E, Cfg>(tw.pre, tau, tw.glob);
      sched_fence();
    }
    // scalar loads return out of order, so a wait for cur[] also waits for everything requested after it:
    // the next stage's twiddles are therefore requested right AFTER this stage's first butterfly has consumed cur[]
    Tw nxt[Cfg::R / 2];
    constexpr bool AHEAD = tw_ahead<Cfg, PH>() && s + 1 < Cfg::stage_end(PH);
    if (S::fwd_fold(s)) {
      // only the "u" side of this stage's butterflies needs its bound back: the "v" side goes through the
      // twiddle multiply, which accepts any word; both outputs then inherit u's bound + TMUL
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) if (!(r & (1 << bpos))) x[r] = fold(x[r], ar.k, ar.fold_c);
    }
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) {
      if (r & (1 << bpos)) continue;
      const Tw w = tw_ahead<Cfg, PH>() ? cur[r >> (bpos + 1)] : tw_get<E, Cfg, PH, s>(tw, thi, r >> (bpos + 1));
      Pol::ct(x[r], x[r | (1 << bpos)], w, ar);
      if constexpr (AHEAD) if (r == 0) {
        sched_fence();
        tw_stage<E, Cfg, PH, (AHEAD ? s + 1 : s)>(tw, tau, nxt);
        sched_fence();
      }
      if (TN_BFLY_FENCE && (r >> (bpos + 1)) % TN_BFLY_FENCE == TN_BFLY_FENCE - 1 && ((r & ((1 << bpos) - 1)) == (1 << bpos) - 1)) sched_fence();
    }
    if constexpr (AHEAD) {
#pragma unroll
      for (int g = 0; g < Cfg::R / 2; ++g) cur[g] = nxt[g];
    }
  });
}
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void fwd_phase(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw, const Arith<E>& ar) {
  typename TwOf<E>::type cur[Cfg::R / 2];
  if constexpr (tw_ahead<Cfg, PH>()) tw_stage<E, Cfg, PH, Cfg::stage_begin(PH)>(tw, tau, cur);
  fwd_phase<E, Cfg, Pol, PH>(x, tau, tw, ar, cur);
}

// One inverse phase (stages of phase PH in reverse order).  cur[]: as in fwd_phase (first executed stage = stage_end - 1).
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void inv_phase(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw, const Arith<E>& ar, typename TwOf<E>::type (&cur)[Cfg::R / 2]) {
  typedef Sched<Pol, Cfg::LOGN> S;
  typedef typename TwOf<E>::type Tw;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<0, Cfg::stage_end(PH) - Cfg::stage_begin(PH)>([&](auto i_) {
    constexpr int s = Cfg::stage_end(PH) - 1 - decltype(i_)::value;   // forward stage number being undone
    constexpr int g = Cfg::LOGN - 1 - s;                               // execution order of the inverse
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    Tw nxt[Cfg::R / 2];
    constexpr bool AHEAD = tw_ahead<Cfg, PH>() && s - 1 >= Cfg::stage_begin(PH) && s - 1 >= 1;   // stage 0 uses ar.ninv*
    if (S::inv_fold(g)) {
#pragma unroll
      for (int r = 0; r < Cfg::R; ++r) x[r] = fold(x[r], ar.k, ar.fold_c);
    }
    constexpr int BND = S::inv_bnd(g);
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) {
      if (r & (1 << bpos)) continue;
      if (s == 0) Pol::template gs_last<BND>(x[r], x[r | (1 << bpos)], ar);
      else {
        const Tw w = tw_ahead<Cfg, PH>() ? cur[r >> (bpos + 1)] : tw_get<E, Cfg, PH, s>(tw, thi, r >> (bpos + 1));
        Pol::template gs<BND>(x[r], x[r | (1 << bpos)], w, ar);
      }
      if constexpr (AHEAD) if (r == 0) {
        sched_fence();
        tw_stage<E, Cfg, PH, (AHEAD ? s - 1 : s)>(tw, tau, nxt);
        sched_fence();
      }
      if (TN_BFLY_FENCE && (r >> (bpos + 1)) % TN_BFLY_FENCE == TN_BFLY_FENCE - 1 && ((r & ((1 << bpos) - 1)) == (1 << bpos) - 1)) sched_fence();
    }
    if constexpr (AHEAD) {
#pragma unroll
      for (int g2 = 0; g2 < Cfg::R / 2; ++g2) cur[g2] = nxt[g2];
    }
  });
}
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void inv_phase(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw, const Arith<E>& ar) {
  typename TwOf<E>::type cur[Cfg::R / 2];
  if constexpr (tw_ahead<Cfg, PH>() && Cfg::stage_end(PH) - 1 >= 1) tw_stage<E, Cfg, PH, Cfg::stage_end(PH) - 1>(tw, tau, cur);
  inv_phase<E, Cfg, Pol, PH>(x, tau, tw, ar, cur);
}

// Reduction of freshly loaded operand words before the first forward stage: only the registers that enter
// stage 0 as "u" (top register bit clear) must be bounded / canonical; the others are multiplied first.
template <typename E, typename Cfg, typename Pol>
TN_HD void load_reduce(E (&x)[Cfg::R], const Arith<E>& ar) {
#pragma unroll
  for (int r = 0; r < Cfg::R / 2; ++r) x[r] = Pol::load(x[r], ar);
}

// LDS transposes.  e = exchange index (between phase e and e+1); `from` = phase
// whose register layout is being written, `to` = phase whose layout is read.
struct alignas(16) Pair64 { u64 lo, hi; };
template <typename E, typename Cfg, int EX, int PH>
TN_HD void ex_store(const E (&x)[Cfg::R], u32 tau, E* lds) {
  if constexpr (Cfg::SWZ && Cfg::pos(PH) == 0) {       // thread owns 8 consecutive coefficients: 16-byte stores
#pragma unroll
    for (int r = 0; r < Cfg::R; r += 2) {
      Pair64 v; v.lo = x[r]; v.hi = x[r + 1];
      *reinterpret_cast<Pair64*>(lds + Cfg::ex_addr(EX, Cfg::jidx(PH, tau, r))) = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) lds[Cfg::ex_addr(EX, Cfg::jidx(PH, tau, r))] = x[r];
  }
}
template <typename E, typename Cfg, int EX, int PH>
TN_HD void ex_load(E (&x)[Cfg::R], u32 tau, const E* lds) {
  if constexpr (Cfg::SWZ && Cfg::pos(PH) == 0) {
#pragma unroll
    for (int r = 0; r < Cfg::R; r += 2) {
      const Pair64 v = *reinterpret_cast<const Pair64*>(lds + Cfg::ex_addr(EX, Cfg::jidx(PH, tau, r)));
      x[r] = (E)v.lo; x[r + 1] = (E)v.hi;
    }
  } else {
#pragma unroll
    for (int r = 0; r < Cfg::R; ++r) x[r] = lds[Cfg::ex_addr(EX, Cfg::jidx(PH, tau, r))];
  }
}

// Pointwise product in the last phase's register layout.  Canonical policy: canonical result.
// Lazy policy: operands are folded below 2^k + eps and the product is left below LazyTraits::PW q
// (Sched::inv_in starts from that bound).  64-bit lanes: split-and-fold product; a plan is only lazy
// if its (k, c) passes h_pw_fast_ok().  Only ONE operand needs folding first: the other may be any value
// below (LIMIT - 2) q, which the forward schedule guarantees (Sched::fwd_out, asserted in pointwise()).
TN_HD u64 pointwise_lazy(u64 a, u64 b, const Arith<u64>& ar) {
  return mulmod_solinas_lazy(fold(a, ar.k, ar.fold_c), b, ar.k, ar.fold_c);       // < 2q
}
TN_HD u32 pointwise_lazy(u32 a, u32 b, const Arith<u32>& ar) {
  return mulmod_barrett_lazy(fold(a, ar.k, ar.fold_c), fold(b, ar.k, ar.fold_c), ar.q, ar.mu, ar.k);   // < 4q
}
template <typename E, typename Cfg, typename Pol>
TN_HD void pointwise(E (&xa)[Cfg::R], const E (&xb)[Cfg::R], const Arith<E>& ar) {
  static_assert(!Pol::lazy || Sched<Pol, Cfg::LOGN>::fwd_out() <= Pol::LIMIT - 2, "pointwise_lazy: unfolded operand bound");
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r) {
    if (Pol::lazy)
      xa[r] = pointwise_lazy(xa[r], xb[r], ar);
    else
      xa[r] = mulmod_barrett(xa[r], xb[r], ar.q, ar.mu, ar.k);
    if (r & 1) sched_fence();          // two products in flight at a time: bounds the live temporaries
  }
}

}  // namespace tn
