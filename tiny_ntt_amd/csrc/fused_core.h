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

// Timing-ablation switch (developer experiments only; results are wrong when set).
#ifndef TN_ABL_UNIFORM_TW
#define TN_ABL_UNIFORM_TW 0      // 1: every thread uses the phase-0 (wave-uniform) twiddle indices -> no vector twiddle loads
#endif

namespace tn {

template <int B, int E_, typename F> struct StaticFor {
  TN_HD static void run(F& f) { f(std::integral_constant<int, B>()); StaticFor<B + 1, E_, F>::run(f); }
};
template <int E_, typename F> struct StaticFor<E_, E_, F> { TN_HD static void run(F&) {} };
template <int B, int E_, typename F> TN_HD void static_for(F&& f) { StaticFor<B, E_, F>::run(f); }

// Coefficients per thread (log2) of the fused kernels for each supported n = 2^logn (0: not built).
#ifndef TN_FUSED_LPT10
#define TN_FUSED_LPT10 3         // n = 1024 (8 per thread: 219 vs 188 M products/s at 24 bits)
#endif
#ifndef TN_FUSED_LPT12
#define TN_FUSED_LPT12 3         // n = 4096
#endif
constexpr int fused_lpt(int logn) {
  return logn == 8 ? 2 : logn == 9 ? 3 : logn == 10 ? TN_FUSED_LPT10 : logn == 11 ? 3 : logn == 12 ? TN_FUSED_LPT12 : logn == 13 ? 3 : 0;
}

// Per-plan arithmetic constants, passed by value to kernels (lives in SGPRs).
template <typename E> struct Arith {
  E q;
  u64 mu;        // floor(2^(2k)/q)
  int k;         // bitlen(q)
  u32 fold_c;    // 2^k - q (lazy policies only)
  SplitK sk;     // split-constant product (lazy 64-bit lanes): 2^p, 2^(p+32) mod q
  u64 qmul[17];  // K q for K = 0..16 (split policy: the multiples that make differences non-negative; read from the
                 // kernel-argument segment by scalar loads where needed instead of being recomputed or kept in registers)
  typename TwOf<E>::type one;        // w = 1 (used to canonicalise arbitrary inputs); Shoup record
  typename TwOf<E>::type ninv;       // n^-1, Shoup record (constant-geometry kernels)
  // last inverse stage of the fused kernels, in the record format of the plan's fused tables
  // (split for lazy 64-bit plans, Shoup otherwise):
  typename TwOf<E>::type fninv;      // n^-1
  typename TwOf<E>::type fninv_w1;   // n^-1 * psi_inv_brv[1]  (odd half)
};

template <typename E, int LOGN_, int LPT_> struct FusedCfg {
  static constexpr int LOGN = LOGN_, LPT = LPT_;
  static constexpr int N = 1 << LOGN, R = 1 << LPT, THREADS = N >> LPT;
  static constexpr int PHASES = (LOGN + LPT - 1) / LPT;
  static constexpr int pos(int p) { return (LOGN - (p + 1) * LPT) < 0 ? 0 : (LOGN - (p + 1) * LPT); }
  static constexpr int stage_begin(int p) { return p * LPT; }
  static constexpr int stage_end(int p) { return ((p + 1) * LPT) < LOGN ? ((p + 1) * LPT) : LOGN; }
  // LDS image used between phase e and e+1:  addr(j) = j + PAD * (j >> SH)   (in elements).  Padded, not XOR-swizzled, so
  // that the address is ADDITIVE in the register index on both sides of the transpose (ex_base + ex_off below): one
  // address register per side and immediate offsets, instead of one register per coefficient (which, as loop invariants
  // of the persistent row loop, cost ~30 VGPRs and pushed the n = 4096 kernel into scratch).
  // Last exchange (into the phase whose threads own R consecutive coefficients): ONE pad element per R coefficients.
  // Element-wide accesses on both sides are then bank-conflict free for both lane widths (no padding makes the wider
  // 128-bit accesses of the owning side AND the element-wide ones of the other side conflict free at once:
  // tests/test_lds_banks.py, tools/lds_layout_search.py).
  static constexpr int ex_pad(int e) { return pos(e + 1) > 0 ? (1 << pos(e + 1)) : 1; }
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
  TN_HD static constexpr u32 jidx(int p, u32 tau, u32 r) {
    return ((tau >> pos(p)) << (pos(p) + LPT)) | (r << pos(p)) | (tau & ((1u << pos(p)) - 1u));
  }
  TN_HD static constexpr u32 ex_addr(int e, u32 j) {
    if (WB > 0 && ex_wave_local(e))
      return (j >> WSH) * (u32)region_elems() + (j & ((1u << WSH) - 1u)) + (u32)ex_pad(e) * ((j & ((1u << WSH) - 1u)) >> ex_sh(e));
    return j + (u32)ex_pad(e) * (j >> ex_sh(e));
  }
  // ex_addr(e, jidx(p, tau, r)) == ex_base(e, p, tau) + ex_off(e, p, r)  for p in {e, e+1}: the thread bits and the
  // register bits of j are disjoint, and the shift in ex_addr cuts either above or below all register bits
  // (checked exhaustively by tests/test_lds_banks.py)
  TN_HD static constexpr u32 ex_base(int e, int p, u32 tau) { return ex_addr(e, jidx(p, tau, 0)); }
  TN_HD static constexpr u32 ex_off(int e, int p, u32 r) { return ex_addr(e, jidx(p, 0, r)); }
  // Where the twiddles of phase p come from:
  //   TW_UNIFORM  every thread of the workgroup uses the same ones        -> scalar loads
  //   TW_WAVE     the same within a wave (index depends on the wave id)     -> scalar loads
  //   TW_LDS      lane-dependent, table small: staged once per workgroup in LDS
  //   TW_REGS     last phase: one private set per thread, fetched from the L2-resident table into
  //               registers (pre[]) just before the transpose that precedes the phase
  //   TW_VEC      lane-dependent and too many for LDS (table indices >= 2^LDS_TW_MAX_STAGE: only n = 8192, whose
  //               second-to-last phase uses 3584 records): vector loads from the L2-resident table at use
  enum { TW_UNIFORM = 0, TW_WAVE = 1, TW_LDS = 2, TW_REGS = 3, TW_VEC = 4 };
  static constexpr int LDS_TW_MAX_STAGE = 9;         // LDS holds table entries below 2^9 (8 KiB of 16-byte records per direction)
  static constexpr int tw_src(int p) {
    return pos(p) >= LOGN - LPT ? TW_UNIFORM
         : (pos(p) >= 6 ? TW_WAVE : (pos(p) > 0 ? (stage_end(p) <= LDS_TW_MAX_STAGE ? TW_LDS : TW_VEC) : TW_REGS));
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
// Arithmetic policies.
//
// Lazy: values are kept only congruent mod q and bounded at compile time; "fold" (one Barrett step with
//   estimate x>>k) is inserted by a static schedule where a bound would overflow the word.  Needs
//   q = 2^k - c with c small; checked at plan creation (plan_tables.h).
//   * 64-bit lanes: split-constant twiddle product (mul_sp_acc, modarith.h), bounds tracked per register in
//     units of 2^k / 4096 by SplitSched below and verified exactly for the plan's (k, c) by h_split_sched_ok().
//   * 32-bit lanes: Shoup product (< 2q), bounds in multiples of q (Sched below), LIMIT = 64.
// Canonical: every value in [0,q) after every operation; any odd q < 2^62 / 2^31.
template <typename E> struct LazyTraits;
// PW = bound of the lazy pointwise product (pointwise() below), in multiples of q
template <> struct LazyTraits<u64> { static constexpr int LIMIT = 16, TMUL = 4, PW = 2; };   // (TMUL: canonical policy's Shoup product, unused by the split path)
template <> struct LazyTraits<u32> { static constexpr int LIMIT = 64, TMUL = 2, PW = 4; };   // mul_tw_lazy < 2q

// CIN (lazy 64-bit lanes only): the caller PROMISES canonical inputs (TN_PLAN_CANONICAL_INPUTS): load() folds nothing and the
// bound schedule starts from q instead of "any word" (SplitSched<Cfg, true>).
template <typename E, bool LAZY, bool CIN = false> struct Policy {
  typedef typename TwOf<E>::type Tw;
  static constexpr bool lazy = LAZY;
  static constexpr bool canonical_inputs = CIN && LAZY && sizeof(E) == 8;
  static constexpr bool split = LAZY && sizeof(E) == 8;
  static constexpr int LIMIT = LazyTraits<E>::LIMIT, TMUL = LazyTraits<E>::TMUL, PW = LazyTraits<E>::PW;

  // an arbitrary word -> a bounded lazy value (< 2^k + eps) or a canonical one
  TN_HD static E load(E x, const Arith<E>& ar) {
    if (canonical_inputs) return x;
    if (LAZY) return fold(x, ar.k, ar.fold_c);
    return mul_tw(x, ar.one, ar.q);
  }
  TN_HD static E canon(E x, const Arith<E>& ar) {              // any lazy value -> [0,q)
    if (LAZY) { x = fold(x, ar.k, ar.fold_c); return csub(x, ar.q); }
    return x;
  }
  // Cooley-Tukey: (u, v) -> (u + w v, u - w v).  K: the lazy product is below K q (schedule constant).
  //   split: x = u + t' rides the multiply-add chain; the other output is 2u + K q - x = u + K q - t'
  //   (mod 2^64; the true value fits by the schedule's bound).
  template <int K> TN_HD static void ct(E& u, E& v, Tw w, const Arith<E>& ar) {
    if constexpr (split) {
      const u64 x = mul_sp_acc(u, v, w, ar.sk);
      v = ((u << 1) + ar.qmul[K]) - x;
      u = x;
    } else if constexpr (LAZY) {
      const E t = mul_tw_lazy(v, w, ar.q);                       // < 2q
      v = u + ((E)K * ar.q - t);
      u = u + t;
    } else {
      E t = mul_tw(v, w, ar.q);
      E s = u + t;
      v = u >= t ? u - t : u + (ar.q - t);
      u = csub(s, ar.q);
    }
  }
  // Gentleman-Sande: (u, v) -> (u + v, (u - v) w);  v < BND q
  template <int BND> TN_HD static void gs(E& u, E& v, Tw w, const Arith<E>& ar) {
    if constexpr (split) {
      const E d = (u + ar.qmul[BND]) - v;
      u = u + v;
      v = mul_sp(d, w, ar.sk);
    } else if constexpr (LAZY) {
      const E d = u + ((E)BND * ar.q - v);
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
    E d;
    if constexpr (split) d = (u + ar.qmul[BND]) - v;
    else d = LAZY ? (E)(u + ((E)BND * ar.q - v)) : (u >= v ? (E)(u - v) : (E)(u + (ar.q - v)));
    E s = u + v;                                               // fits the word by the schedule's bound
    u = mul_tw_canon(s, ar.fninv, ar);
    v = mul_tw_canon(d, ar.fninv_w1, ar);
  }
  // canonical twiddle product.  Split: the product is below 7 * 2^k for any word; one fold puts it below 2q,
  // so ONE conditional subtraction finishes.
  TN_HD static E mul_tw_canon(E a, Tw w, const Arith<E>& ar) {
    if constexpr (split) return csub(fold(mul_sp(a, w, ar.sk), ar.k, ar.fold_c), ar.q);
    else return mul_tw(a, w, ar.q);
  }
};

// Static fold schedule of the 32-bit lazy policy (and the trivial one of the canonical policy), in multiples
// of q.  fwd: bound grows by TMUL per stage.  inv: bound -> max(2B, TMUL).
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

// Bound schedule of the split-constant policy (lazy 64-bit lanes), per register, in units of 2^k / 4096,
// laid out for the worst case k = 60 (2^64 = 16 * 2^k; smaller k only has more room).  Every value of the
// transform has a compile-time upper bound; the schedule decides where a fold() is needed and which multiple
// K of q makes a difference non-negative.  Within a register phase bounds are tracked per register; an LDS
// transpose mixes registers of different threads, so the next phase starts from their maximum.
//   product of a value below bv:  t' < tmax(bv) = 2^(k+1) + bv/8 + 2^(k+1) + 2^32 cf   (mul_sp_acc)
//   Cooley-Tukey (u, v):  K = ceil(tmax(bv) / q);  needs u + max(tmax, K q) <= 2^64, else u is folded first;
//                         outputs u + tmax(bv), u + K q
//   Gentleman-Sande (u, v):  Kv = ceil(bv / q);  needs u + Kv q <= 2^64 and u + v <= 2^64, else the larger is
//                         folded first; outputs u + v, tmax(u + Kv q)
// The decisions are compile-time guesses in coarse units; h_split_sched_ok() (plan_tables.h) replays them with
// exact 128-bit bounds for the plan's (k, c), and a plan whose modulus fails that replay is not lazy.
template <typename Cfg, bool CIN = false> struct SplitSched {
  static constexpr int LOGN = Cfg::LOGN, R = Cfg::R;
  static constexpr long U = 4096, CAP = 16 * U;
  static constexpr long FOLDED = U + 1;            // fold(): < 2^k + 2^(64-k) c
  static constexpr long PW_OUT = 2 * U;            // mulmod_solinas_lazy: < 2q
  static constexpr long PW_IN = 14 * (U - 1);      // ... for an unfolded operand below 14 q
  static constexpr long tmax(long bv) { return 4 * U + (bv + 7) / 8 + 2; }
  static constexpr int kq(long b) { return (int)((b + U - 2) / (U - 1)); }      // smallest K with K q >= b  (q >= (U-1) units)
  // The multiples K q live in SGPR pairs; only a few distinct ones are used so that they stay resident across the
  // persistent row loop: 6q or 7q in the forward butterflies, multiples of 4q in the inverse ones (one more fold per
  // inverse transform than with exact multiples).
  static constexpr int kf(long t) { return kq(t) <= 6 ? 6 : kq(t); }
  static constexpr int ki(long b) { return (kq(b) + 3) / 4 * 4; }
  static constexpr int phase_of(int s) { return s / Cfg::LPT; }
  static constexpr int bpos_of(int s) { return (LOGN - 1 - s) - Cfg::pos(phase_of(s)); }
  struct Data {
    bool ffold[LOGN][R] = {};          // forward stage s: fold register r (a "u" of the stage) first
    unsigned char fk[LOGN][R] = {};    // forward stage s: K of the butterfly whose u is register r
    long fout = 0;                     // bound of every forward output
    bool pw_fold_b = false;            // pointwise: the second operand must be folded too
    bool ifold[LOGN][R] = {};          // inverse stage g (execution order): fold register r first
    unsigned char ik[LOGN][R] = {};    // inverse stage g: Kv of the butterfly whose u is register r
  };
  static constexpr Data build() {
    Data d;
    long b[R] = {};
    // forward: load_reduce() folds the registers that enter stage 0 as "u" (the low half), the rest are raw words;
    // promised-canonical inputs: every register is below q
    for (int r = 0; r < R; ++r) b[r] = CIN ? U : (r < R / 2 ? FOLDED : CAP);
    for (int s = 0; s < LOGN; ++s) {
      if (s > 0 && phase_of(s) != phase_of(s - 1)) {
        long m = 0;
        for (int r = 0; r < R; ++r) m = b[r] > m ? b[r] : m;
        for (int r = 0; r < R; ++r) b[r] = m;
      }
      const int bit = 1 << bpos_of(s);
      for (int r = 0; r < R; ++r) {
        if (r & bit) continue;
        const long t = tmax(b[r | bit]);
        const int K = kf(t);
        const long grow = t > K * U ? t : K * U;
        if (b[r] + grow > CAP) { d.ffold[s][r] = true; b[r] = FOLDED; }
        d.fk[s][r] = (unsigned char)K;
        const long bu = b[r];
        b[r] = bu + t;
        b[r | bit] = bu + K * U;
      }
    }
    for (int r = 0; r < R; ++r) d.fout = b[r] > d.fout ? b[r] : d.fout;
    d.pw_fold_b = d.fout > PW_IN;
    // inverse, execution order g (g = 0 undoes forward stage LOGN-1)
    for (int r = 0; r < R; ++r) b[r] = PW_OUT;
    for (int g = 0; g < LOGN; ++g) {
      const int s = LOGN - 1 - g;
      if (g > 0 && phase_of(s) != phase_of(s + 1)) {
        long m = 0;
        for (int r = 0; r < R; ++r) m = b[r] > m ? b[r] : m;
        for (int r = 0; r < R; ++r) b[r] = m;
      }
      const int bit = 1 << bpos_of(s);
      for (int r = 0; r < R; ++r) {
        if (r & bit) continue;
        const int v = r | bit;
        for (int it = 0; it < 2; ++it) {
          if (b[r] + ki(b[v]) * U <= CAP && b[r] + b[v] <= CAP) break;
          if (b[r] >= b[v]) { d.ifold[g][r] = true; b[r] = FOLDED; }
          else { d.ifold[g][v] = true; b[v] = FOLDED; }
        }
        const int Kv = ki(b[v]);
        d.ik[g][r] = (unsigned char)Kv;
        const long bd = b[r] + Kv * U;
        b[r] = b[r] + b[v];
        b[v] = tmax(bd);
      }
    }
    return d;
  }
  static constexpr Data D = build();
};

// What the phase loops ask the schedule of their policy.
template <typename Pol, typename Cfg> struct SchedOf {
  typedef Sched<Pol, Cfg::LOGN> S;
  static constexpr bool fwd_fold(int s, int r) { return S::fwd_fold(s) && !(r & (1 << SplitSched<Cfg>::bpos_of(s))); }
  static constexpr int fwd_k(int, int) { return Pol::TMUL; }
  static constexpr bool inv_fold(int g, int) { return S::inv_fold(g); }
  static constexpr int inv_k(int g, int) { return S::inv_bnd(g); }
  static constexpr bool pw_fold_b() { return false; }
  static constexpr bool pw_ok() { return !Pol::lazy || S::fwd_out() <= Pol::LIMIT - 2; }
};
template <typename Cfg, bool CIN> struct SchedOf<Policy<u64, true, CIN>, Cfg> {
  typedef SplitSched<Cfg, CIN> S;
  static constexpr bool fwd_fold(int s, int r) { return S::D.ffold[s][r]; }
  static constexpr int fwd_k(int s, int r) { return S::D.fk[s][r]; }
  static constexpr bool inv_fold(int g, int r) { return S::D.ifold[g][r]; }
  static constexpr int inv_k(int g, int r) { return S::D.ik[g][r]; }
  static constexpr bool pw_fold_b() { return S::D.pw_fold_b; }
  static constexpr bool pw_ok() { return true; }
};

// The three places a phase can take its twiddles from (see FusedCfg::tw_src).
template <typename E> struct TwRefs {
  typedef typename TwOf<E>::type Tw;
  const Tw* __restrict__ glob;     // full table psi^brv(i) (or inverse) in global memory / L2
  const Tw* lds;                   // entries [lds_tw_lo, lds_tw_hi) of it, staged in LDS
  Tw* pre;                         // the calling thread's last-phase twiddles, in registers
  const Tw* mid = nullptr;         // optional: the calling thread's twiddles of the LDS-sourced phase, already in registers
  u32 zero = 0;                    // opaque_zero() of the current row, added to the index of scalar (wave-uniform) twiddle loads:
                                   // keeps those loads inside the persistent row loop (see polymul_fused_kernel)
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
  if (Cfg::tw_src(PH) == Cfg::TW_VEC && t.mid) return t.mid[(1 << (S_ - Cfg::stage_begin(PH))) - 1 + g];   // prefetched (tw_fetch_vec)
  return t.glob[idx + t.zero];
}

// The calling thread's twiddles of a FULL vector-loaded phase PH (TW_VEC), requested from the L2-resident table into
// registers ahead of the transpose that precedes the phase (same register layout as tw_fetch_mid).
template <typename E, typename Cfg, int PH>
TN_HD void tw_fetch_vec(typename TwOf<E>::type (&mid)[Cfg::R], u32 tau, const typename TwOf<E>::type* __restrict__ glob) {
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    constexpr int cnt = Cfg::R >> (bpos + 1), off = (1 << (s - Cfg::stage_begin(PH))) - 1;
    const u32 boff = ((1u << s) + (thi << (Cfg::LPT - bpos - 1))) * (u32)sizeof(typename TwOf<E>::type);
    const typename TwOf<E>::type* base = reinterpret_cast<const typename TwOf<E>::type*>(reinterpret_cast<const char*>(glob) + boff);
#pragma unroll
    for (int g = 0; g < cnt; ++g) mid[off + g] = base[g];
  });
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

// Fetch the calling thread's last-phase twiddles of stages [S0, S1) into registers (issued ahead of their use).
template <typename E, typename Cfg, int S0, int S1>
TN_HD void tw_prefetch_stages(typename TwOf<E>::type* pre, u32 tau, const typename TwOf<E>::type* __restrict__ glob) {
  constexpr int PH = Cfg::PHASES - 1;
  if (Cfg::tw_src(PH) != Cfg::TW_REGS) return;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<S0, S1>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    // byte offset computed in 32 bits: the access is then "uniform table base + one 32-bit thread offset + immediate"
    // (a 64-bit index makes every record's address a VGPR pair that lives across the persistent row loop)
    const u32 off = ((1u << s) + (thi << (Cfg::LPT - bpos - 1))) * (u32)sizeof(typename TwOf<E>::type);
    const typename TwOf<E>::type* base = reinterpret_cast<const typename TwOf<E>::type*>(reinterpret_cast<const char*>(glob) + off);
#pragma unroll
    for (int g = 0; g < Cfg::pre_count(s); ++g) pre[Cfg::pre_off(s) + g] = base[g];
  });
}
template <typename E, typename Cfg>
TN_HD void tw_prefetch_raw(typename TwOf<E>::type* pre, u32 tau, const typename TwOf<E>::type* __restrict__ glob) {
  tw_prefetch_stages<E, Cfg, Cfg::stage_begin(Cfg::PHASES - 1), Cfg::LOGN>(pre, tau, glob);
}

template <typename E, typename Cfg>
TN_HD void tw_prefetch(typename TwOf<E>::type (&pre)[Cfg::NPRE], u32 tau, const typename TwOf<E>::type* __restrict__ glob) {
  tw_prefetch_raw<E, Cfg>(pre, tau, glob);
}

// ---------------------------------------------------------------------------
// One forward phase on a thread's registers.
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void fwd_phase(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw, const Arith<E>& ar) {
  typedef SchedOf<Pol, Cfg> SO;
  typedef typename TwOf<E>::type Tw;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<Cfg::stage_begin(PH), Cfg::stage_end(PH)>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    // only the "u" side of a butterfly can need its bound back: the "v" side goes through the twiddle
    // multiply, which accepts any word
    static_for<0, Cfg::R>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      if constexpr (Pol::lazy && SO::fwd_fold(s, r)) x[r] = fold(x[r], ar.k, ar.fold_c);
    });
    static_for<0, Cfg::R>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      if constexpr (!(r & (1 << bpos))) {
        const Tw w = tw_get<E, Cfg, PH, s>(tw, thi, r >> (bpos + 1));
        Pol::template ct<SO::fwd_k(s, r)>(x[r], x[r | (1 << bpos)], w, ar);
      }
    });
  });
}

// One inverse phase (stages of phase PH in reverse order).
template <typename E, typename Cfg, typename Pol, int PH>
TN_HD void inv_phase(E (&x)[Cfg::R], u32 tau, const TwRefs<E>& tw, const Arith<E>& ar) {
  typedef SchedOf<Pol, Cfg> SO;
  typedef typename TwOf<E>::type Tw;
  const u32 thi = Cfg::thi(PH, tau);
  static_for<0, Cfg::stage_end(PH) - Cfg::stage_begin(PH)>([&](auto i_) {
    constexpr int s = Cfg::stage_end(PH) - 1 - decltype(i_)::value;   // forward stage number being undone
    constexpr int g = Cfg::LOGN - 1 - s;                               // execution order of the inverse
    constexpr int bpos = (Cfg::LOGN - 1 - s) - Cfg::pos(PH);
    static_for<0, Cfg::R>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      if constexpr (Pol::lazy && SO::inv_fold(g, r)) x[r] = fold(x[r], ar.k, ar.fold_c);
    });
    static_for<0, Cfg::R>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      if constexpr (!(r & (1 << bpos))) {
        constexpr int BND = SO::inv_k(g, r);
        if constexpr (s == 0) Pol::template gs_last<BND>(x[r], x[r | (1 << bpos)], ar);
        else {
          const Tw w = tw_get<E, Cfg, PH, s>(tw, thi, r >> (bpos + 1));
          Pol::template gs<BND>(x[r], x[r | (1 << bpos)], w, ar);
        }
      }
    });
  });
}

// Reduction of freshly loaded operand words before the first forward stage: only the registers that enter
// stage 0 as "u" (top register bit clear) must be bounded / canonical; the others are multiplied first.
template <typename E, typename Cfg, typename Pol>
TN_HD void load_reduce(E (&x)[Cfg::R], const Arith<E>& ar) {
#pragma unroll
  for (int r = 0; r < Cfg::R / 2; ++r) x[r] = Pol::load(x[r], ar);
}

// LDS transposes.  EX = exchange index (between phase EX and EX+1); PH = the phase whose register layout is
// written / read.  One address per thread and side, immediate offsets per register.
template <typename E, typename Cfg, int EX, int PH>
TN_HD void ex_store(const E (&x)[Cfg::R], u32 tau, E* lds) {
  E* base = lds + Cfg::ex_base(EX, PH, tau);
  static_for<0, Cfg::R>([&](auto r_) { constexpr int r = decltype(r_)::value; base[Cfg::ex_off(EX, PH, r)] = x[r]; });
}
template <typename E, typename Cfg, int EX, int PH>
TN_HD void ex_load(E (&x)[Cfg::R], u32 tau, const E* lds) {
  const E* base = lds + Cfg::ex_base(EX, PH, tau);
  static_for<0, Cfg::R>([&](auto r_) { constexpr int r = decltype(r_)::value; x[r] = base[Cfg::ex_off(EX, PH, r)]; });
}

// Pointwise product in the last phase's register layout.  Canonical policy: canonical result.
// Lazy policy: operands are folded below 2^k + eps and the product is left below LazyTraits::PW q
// (the inverse schedule starts from that bound).  64-bit lanes: split-and-fold product; a plan is only lazy
// if its (k, c) passes h_pw_fast_ok().  Only ONE operand needs folding first: the other may be any value
// below 14 q, which the forward schedule guarantees or repairs (SchedOf::pw_fold_b).
TN_HD u64 pointwise_lazy(u64 a, u64 b, const Arith<u64>& ar) {
  return mulmod_solinas_lazy(fold(a, ar.k, ar.fold_c), b, ar.k, ar.fold_c);       // < 2q
}
TN_HD u32 pointwise_lazy(u32 a, u32 b, const Arith<u32>& ar) {
  return mulmod_barrett_lazy(fold(a, ar.k, ar.fold_c), fold(b, ar.k, ar.fold_c), ar.q, ar.mu, ar.k);   // < 4q
}
template <typename E, typename Cfg, typename Pol>
TN_HD void pointwise(E (&xa)[Cfg::R], const E (&xb)[Cfg::R], const Arith<E>& ar) {
  typedef SchedOf<Pol, Cfg> SO;
  static_assert(SO::pw_ok(), "pointwise_lazy: unfolded operand bound");
#pragma unroll
  for (int r = 0; r < Cfg::R; ++r) {
    if (Pol::lazy)
      xa[r] = pointwise_lazy(xa[r], SO::pw_fold_b() ? fold(xb[r], ar.k, ar.fold_c) : xb[r], ar);
    else
      xa[r] = mulmod_barrett(xa[r], xb[r], ar.q, ar.mu, ar.k);
    if (r & 1) sched_fence();          // two products in flight at a time: bounds the live temporaries
  }
}

}  // namespace tn
