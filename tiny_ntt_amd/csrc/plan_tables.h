// plan_tables.h — host-side, exact generation of every constant a plan needs.
// Shared by capi.cpp (uploads them) and tests/emu (steps the kernels on the CPU).
#pragma once
#include <vector>
#include "fused_core.h"

namespace tn {

struct HostTables {
  u32 n = 0, logn = 0;
  u64 q = 0, psi = 0, omega = 0, mu = 0;
  int k = 0, elem_bytes = 8;
  bool lazy = false;
  bool cg_lazy = false;      // lazy, 64-bit lanes, and the constant-geometry kernels may run lazy butterflies (h_cg_lazy_ok)
  bool cg_sched = false;     // ... with the static fold schedule (h_cg_sched_ok; even log2 n only)
  bool cin_ok = false;       // the fused product kernel's schedule for promised-canonical inputs is valid (h_split_sched_cin_ok)
  u32 fold_c = 0;
  u64 n_inv = 0, ninv_w1 = 0;
  std::vector<u64> psi_pow, psi_inv_pow, psi_inv_ninv, psi_brv, psi_inv_brv, omega_pow, omega_inv_pow;
  // merged twiddles of the CYCLIC transform (x^n - 1 factorisation tree): node m+i (m = 2^s nodes at level s) splits
  // x^(n/m) - zeta with sqrt(zeta) = psi^(brv(m+i) - n/(2m)); same butterflies, same bit-reversed output order as
  // the negacyclic table psi_brv, so cg_ntt / cg_intt run on the fused kernel without a pre- or post-twist
  std::vector<u64> cyc_brv, cyc_inv_brv;
};

inline u32 h_brv(u32 v, u32 bits) {
  u32 r = 0;
  for (u32 i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

// Lazy-reduction eligibility: q = 2^k - c with LIMIT*q <= 2^W and a fold (one Barrett
// step with estimate x >> k) that lands below 2q for every word x.
inline bool h_lazy_ok(u64 q, int elem_bytes, u32* fold_c) {
  const int W = elem_bytes * 8, k = h_bitlen(q);
  const unsigned __int128 two_w = ((unsigned __int128)1) << W;
  const int LIMIT = elem_bytes == 8 ? LazyTraits<u64>::LIMIT : LazyTraits<u32>::LIMIT;
  const u64 c = (((u64)1) << k) - q;
  const bool fits = (unsigned __int128)LIMIT * q <= two_w;
  const unsigned __int128 top = (two_w >> k);                   // max (x >> k) + 1
  const bool fold_ok = c < ((u64)1 << 32) && (unsigned __int128)c * (top + 1) <= q && (elem_bytes == 4 || k >= 32);
  if (fold_c) *fold_c = (fits && fold_ok) ? (u32)c : 0;
  return fits && fold_ok;
}

// Is mulmod_solinas_lazy (the lazy 64-bit pointwise product) valid for this (k, c)?  Worst-case value of every intermediate of
// that routine for this (k, c), with one operand at the fold() output bound and the other at (LIMIT-2) q; all must fit their words and the
// result must stay below 2q.
inline bool h_pw_fast_ok(u64 q, int k, u64 c) {
  typedef unsigned __int128 u128;
  if (k < 32 || k > 60 || c == 0 || c >= ((u64)1 << 31)) return false;
  const int s = k - 32;
  const u128 one = 1, w64 = one << 64, w32 = one << 32;
  const u128 amax = (one << k) - 1 + ((one << (64 - k)) - 1) * c;          // fold() output bound
  if (amax >= (one << 62)) return false;
  const u128 bmax = (u128)(LazyTraits<u64>::LIMIT - 2) * q;                // the unfolded operand (Sched::fwd_out)
  if (bmax >= w64) return false;
  const u128 pmax = amax * bmax;
  const u128 phmax = pmax >> k;
  if (phmax >= w64) return false;
  const u128 ph1max = phmax >> 32;
  const u128 umax = ph1max * c;
  if (umax >= w64) return false;
  const u128 uhmax = umax >> s;
  const u128 tmax = (one << k) - 1 + (w32 - 1) * c;
  if (tmax >= w64) return false;
  const u128 vhmax = (tmax >> 32) + ((one << s) - 1);
  if (vhmax >= w32) return false;
  const u128 topmax = (vhmax >> s) + uhmax;
  if (topmax >= w32) return false;
  const u128 rmax = (one << k) - 1 + topmax * c;
  return rmax < (u128)2 * q;
}

// ---- split-constant policy (lazy 64-bit lanes): exact replay of SplitSched for the plan's (k, c) -------------
// All bounds are EXCLUSIVE upper bounds held in 128 bits.
struct SplitExact {
  typedef unsigned __int128 u128;
  int k, p;
  u64 q, c, cf;
  u128 two64;
  bool ok = true;
  SplitExact(int k_, u64 c_) : k(k_), p(k_ - 31), c(c_) {
    q = (((u64)1) << k) - c;
    two64 = ((u128)1) << 64;
    cf = p >= 1 ? (u64)((((u128)1) << (p + 32)) % q) : 0;
    if (k < 32 || k > 60 || cf >= ((u64)1 << 32) || c >= ((u64)1 << 32)) ok = false;
  }
  // t' of mul_sp_acc for a value below bv (h_sp_tmax): exclusive bound; also checks that H fits 64 bits
  u128 tmax(u128 bv) {
    const u128 m32 = (((u128)1) << 32) - 1;
    u128 a1 = (bv - 1) >> 32; if (a1 > m32) a1 = m32;
    const u128 wlo = (((u128)1) << p) - 1, whi = (u128)((q - 1) >> p);
    const u128 H = m32 * whi + a1 * whi;
    if (H >= two64 || whi > m32) ok = false;
    return m32 * wlo + a1 * wlo + (m32 << p) + (H >> 32) * cf + 1;
  }
  u128 folded(u128 b) { return (((u128)1) << k) + ((b - 1) >> k) * c; }           // fold() of a value below b
  void fits(u128 exclusive) { if (exclusive > two64) ok = false; }
};

template <typename Cfg, bool CIN = false> inline bool h_split_sched_replay(int k, u64 c) {
  typedef SplitSched<Cfg, CIN> S;
  typedef unsigned __int128 u128;
  SplitExact x(k, c);
  if (!x.ok) return false;
  constexpr int R = Cfg::R, LOGN = Cfg::LOGN;
  const auto& D = S::D;
  u128 b[R];
  auto level = [&]() { u128 m = 0; for (int r = 0; r < R; ++r) m = b[r] > m ? b[r] : m; for (int r = 0; r < R; ++r) b[r] = m; };
  for (int r = 0; r < R; ++r) b[r] = CIN ? (u128)x.q : (r < R / 2 ? x.folded(x.two64) : x.two64);   // load_reduce(): low half folded, rest raw words (promised-canonical inputs: all below q)
  for (int s = 0; s < LOGN; ++s) {
    if (s > 0 && S::phase_of(s) != S::phase_of(s - 1)) level();
    const int bit = 1 << S::bpos_of(s);
    for (int r = 0; r < R; ++r) {
      if (r & bit) continue;
      if (D.ffold[s][r]) b[r] = x.folded(b[r]);
      const u128 t = x.tmax(b[r | bit]), kq = (u128)D.fk[s][r] * x.q;
      if (kq + 1 < t) x.ok = false;                      // K q >= t'max: the difference output is non-negative
      x.fits(b[r] - 1 + t); x.fits(b[r] + kq);
      const u128 bu = b[r];
      b[r] = bu + t - 1; b[r | bit] = bu + kq;
    }
  }
  u128 fout = 0;
  for (int r = 0; r < R; ++r) fout = b[r] > fout ? b[r] : fout;
  if (x.folded(fout) > (u128)2 * x.q) x.ok = false;    // Policy::canon(): fold, one conditional subtraction
  if (!D.pw_fold_b && fout > (u128)14 * x.q) x.ok = false;       // mulmod_solinas_lazy's unfolded operand (h_pw_fast_ok)
  for (int r = 0; r < R; ++r) b[r] = (u128)2 * x.q;    // pointwise product: < 2q (h_pw_fast_ok); folded loads are below that too
  if (x.folded(x.two64) > (u128)2 * x.q) x.ok = false;
  for (int g = 0; g < LOGN; ++g) {
    const int s = LOGN - 1 - g;
    if (g > 0 && S::phase_of(s) != S::phase_of(s + 1)) level();
    const int bit = 1 << S::bpos_of(s);
    for (int r = 0; r < R; ++r) if (D.ifold[g][r]) b[r] = x.folded(b[r]);
    for (int r = 0; r < R; ++r) {
      if (r & bit) continue;
      const int v = r | bit;
      const u128 kq = (u128)D.ik[g][r] * x.q;
      if (kq + 1 < b[v]) x.ok = false;                   // Kv q >= v: u + Kv q - v is non-negative
      x.fits(b[r] + kq); x.fits(b[r] + b[v] - 1);
      const u128 bd = b[r] + kq;
      b[r] = b[r] + b[v] - 1;
      b[v] = x.tmax(bd);
    }
  }
  // last inverse stage (gs_last): both outputs go through mul_tw_canon = csub(fold(mul_sp(any word)))
  if (x.folded(x.tmax(x.two64)) > (u128)2 * x.q) x.ok = false;
  return x.ok;
}

// Lazy butterflies of the constant-geometry kernels (cg_butterfly_lazy in kernels.hip): left folded, product riding,
// difference 2 left + 6q - x; every stage input below B = fold bound + 6q.  Exact replay for (k, c).
inline bool h_cg_lazy_ok(int k, u64 c) {
  typedef unsigned __int128 u128;
  SplitExact x(k, c);
  if (!x.ok) return false;
  const u128 bu = x.folded(x.two64);                // fold() of any word
  const u128 kq = (u128)6 * x.q;
  const u128 B = bu + kq;                           // bound of the difference output; the sum output must stay below it too
  const u128 t = x.tmax(B);
  if (kq + 1 < t) x.ok = false;                     // 6q >= t'max
  x.fits(bu - 1 + t); x.fits(B);
  if (bu + t - 1 > B) x.ok = false;
  if (x.folded(B) > (u128)2 * x.q) x.ok = false;    // canonicalisation afterwards: fold, one conditional subtraction
  if (x.tmax(x.two64) > B) x.ok = false;            // lazy twist on load: the bare product of any word is a valid stage input
  if (B > (u128)14 * (x.q - 1)) x.ok = false;       // lazy pointwise product (pointwise_lazy): second operand below 14q, result < 2q <= B
  // merged twist + stage-1 butterfly (CgArith::bf_first): left folded, right ANY word, K = 7; its outputs enter stage 2 (left folded again)
  {
    const u128 t1 = x.tmax(x.two64), k7 = (u128)7 * x.q;
    if (k7 + 1 < t1) x.ok = false;
    x.fits(bu - 1 + t1); x.fits(bu + k7);
    const u128 out1 = (bu + t1 - 1) > (bu + k7) ? (bu + t1 - 1) : (bu + k7);
    const u128 t2 = x.tmax(out1);
    if (kq + 1 < t2) x.ok = false;                  // stage 2 still uses 6q
    x.fits(bu - 1 + t2);
    if (bu + t2 - 1 > B) x.ok = false;
  }
  return x.ok;
}

// Scheduled lazy butterflies of the constant-geometry kernels (CgArith<.., CGA_SPLIT_SCHED>, cg_core.h): odd stages run on the
// raw left input with 5q, even stages fold it and use 6q.  Exact replay of the two-stage cycle for (k, c), started from the
// worst stage-1 input (the bare twist product of any word) and iterated to its fixed point.
inline bool h_cg_sched_ok(int k, u64 c) {
  typedef unsigned __int128 u128;
  SplitExact x(k, c);
  if (!x.ok) return false;
  u128 b_odd = x.tmax(x.two64);                       // input bound of an odd stage: twist output (folded / pointwise inputs are smaller)
  const u128 fold_any = x.folded(x.two64);
  if (fold_any > b_odd) b_odd = fold_any;
  // merged twist + stage-1 butterfly (CgArith::bf_first): left = the bare twist product (< tmax(2^64)), right ANY word, K = 7;
  // its outputs are the inputs of stage 2 (even: left folded, 6q)
  u128 out_merged = 0;
  {
    const u128 bl = x.tmax(x.two64), t1 = x.tmax(x.two64), k7 = (u128)7 * x.q;
    if (k7 + 1 < t1) x.ok = false;
    x.fits(bl - 1 + t1); x.fits(bl + k7);
    out_merged = (bl + t1 - 1) > (bl + k7) ? (bl + t1 - 1) : (bl + k7);
  }
  for (int it = 0; it < 4; ++it) {
    // odd stage: u raw (< b_odd), v < b_odd
    const u128 t1 = x.tmax(b_odd), k5 = (u128)5 * x.q;
    if (k5 + 1 < t1) x.ok = false;
    x.fits(b_odd - 1 + t1); x.fits(b_odd + k5);
    const u128 s1 = b_odd + t1 - 1, d1 = b_odd + k5;
    u128 b_even = s1 > d1 ? s1 : d1;
    if (out_merged > b_even) b_even = out_merged;      // stage 2 may follow the merged first stage
    // even stage: u folded, v < b_even
    const u128 bu = x.folded(b_even), t2 = x.tmax(b_even), k6 = (u128)6 * x.q;
    if (k6 + 1 < t2) x.ok = false;
    x.fits(bu - 1 + t2); x.fits(bu + k6);
    const u128 s2 = bu + t2 - 1, d2 = bu + k6;
    const u128 out = s2 > d2 ? s2 : d2;
    if (out <= b_odd) {
      // outputs of the last (even) stage: pointwise_lazy's unfolded operand (< 14q), canonicalisation = fold + one subtraction
      if (b_odd > (u128)14 * (x.q - 1)) x.ok = false;
      if (x.folded(b_odd) > (u128)2 * x.q) x.ok = false;
      return x.ok;
    }
    b_odd = out;
  }
  return false;
}

// ... and of the schedule for promised-canonical inputs (TN_PLAN_CANONICAL_INPUTS; built for n = 4096 only)
inline bool h_split_sched_cin_ok(u32 logn, int k, u64 c) {
  return logn == 12 && h_split_sched_replay<FusedCfg<u64, 12, fused_lpt(12)>, true>(k, c);
}

// Is the split-constant lazy policy valid for this (n, k, c)?  (false too when no fused kernel is built for n)
inline bool h_split_sched_ok(u32 logn, int k, u64 c) {
  switch (logn) {
    case 8: return h_split_sched_replay<FusedCfg<u64, 8, fused_lpt(8)>>(k, c);
    case 9: return h_split_sched_replay<FusedCfg<u64, 9, fused_lpt(9)>>(k, c);
    case 10: return h_split_sched_replay<FusedCfg<u64, 10, fused_lpt(10)>>(k, c);
    case 11: return h_split_sched_replay<FusedCfg<u64, 11, fused_lpt(11)>>(k, c);
    case 12: return h_split_sched_replay<FusedCfg<u64, 12, fused_lpt(12)>>(k, c);
    case 13: return h_split_sched_replay<FusedCfg<u64, 13, fused_lpt(13)>>(k, c);
    default: return false;
  }
}

// Preconditions (checked by the caller): n = 2^logn >= 4, q odd prime < 2^62, psi^n == -1.
inline HostTables h_build_tables(u32 n, u64 q, u64 psi, bool allow_lazy) {
  HostTables t;
  u32 logn = 0;
  while (((u32)1 << logn) < n) ++logn;
  t.n = n; t.logn = logn; t.q = q; t.psi = psi % q; t.omega = h_mulmod(t.psi, t.psi, q);
  t.elem_bytes = q < ((u64)1 << 31) ? 4 : 8;
  t.k = h_bitlen(q);
  t.mu = (u64)((((unsigned __int128)1) << (2 * t.k)) / q);
  t.lazy = h_lazy_ok(q, t.elem_bytes, &t.fold_c) && allow_lazy;
  if (t.lazy && t.elem_bytes == 8 && !h_pw_fast_ok(q, t.k, t.fold_c)) t.lazy = false;     // 64-bit lazy pointwise product needs it
  if (t.lazy && t.elem_bytes == 8 && !h_split_sched_ok(logn, t.k, t.fold_c)) t.lazy = false;   // ... and the butterflies this
  if (!t.lazy) t.fold_c = 0;
  t.cg_lazy = t.lazy && t.elem_bytes == 8 && h_cg_lazy_ok(t.k, t.fold_c);
  t.cg_sched = t.cg_lazy && (logn & 1) == 0 && h_cg_sched_ok(t.k, t.fold_c);
  t.cin_ok = t.lazy && t.elem_bytes == 8 && h_split_sched_cin_ok(logn, t.k, t.fold_c);
  const u64 psi_inv = h_powmod(t.psi, q - 2, q);                // modinv: cg_ntt.py:9-10, :91
  const u64 omega_inv = h_powmod(t.omega, q - 2, q);            // :72
  t.n_inv = h_powmod(n % q, q - 2, q);                          // :74
  std::vector<u64>& psi_inv_pow = t.psi_inv_pow;
  psi_inv_pow.resize(n);
  t.psi_pow.resize(n); t.psi_inv_ninv.resize(n); t.psi_brv.resize(n); t.psi_inv_brv.resize(n);
  t.omega_pow.resize(n / 2 + 1); t.omega_inv_pow.resize(n / 2 + 1);      // one entry more than cg_ntt.py uses: omega^(n/2) = -1,
  u64 f = 1, g = 1;                                                          // so that omega^-i = -omega^(n/2 - i) for EVERY i < n/2 (cg_core.h)
  for (u32 i = 0; i < n; ++i) { t.psi_pow[i] = f; psi_inv_pow[i] = g; f = h_mulmod(f, t.psi, q); g = h_mulmod(g, psi_inv, q); }
  for (u32 i = 0; i < n; ++i) {
    t.psi_brv[i] = t.psi_pow[h_brv(i, logn)];
    t.psi_inv_brv[i] = psi_inv_pow[h_brv(i, logn)];
    t.psi_inv_ninv[i] = h_mulmod(psi_inv_pow[i], t.n_inv, q);
  }
  t.cyc_brv.assign(n, 1); t.cyc_inv_brv.assign(n, 1);
  for (u32 i = 1; i < n; ++i) {
    u32 m = 1;
    while (2 * m <= i) m *= 2;
    const u32 e = h_brv(i, logn) - n / (2 * m);
    t.cyc_brv[i] = t.psi_pow[e];
    t.cyc_inv_brv[i] = psi_inv_pow[e];
  }
  u64 w = 1, wi = 1;
  for (u32 j = 0; j <= n / 2; ++j) { t.omega_pow[j] = w; t.omega_inv_pow[j] = wi; w = h_mulmod(w, t.omega, q); wi = h_mulmod(wi, omega_inv, q); }
  t.ninv_w1 = h_mulmod(t.n_inv, t.psi_inv_brv[1], q);
  return t;
}

// Tables of an OMEGA-ONLY plan: cg_ntt(a, omega_n, modulus) / cg_intt for ANY omega_n and any modulus >= 2 (cg_ntt.py:29-75
// only evaluates the butterflies with pow(omega_n, k (i // k), modulus); the inverse uses modinv(omega_n) = omega_n^(q-2), :72,
// and n^(q-2), :74, whatever omega_n and the modulus are).  No psi: only the constant-geometry transforms exist for such a
// plan.  Canonical policy (Shoup records).
inline HostTables h_build_omega_tables(u32 n, u64 q, u64 omega) {
  HostTables t;
  u32 logn = 0;
  while (((u32)1 << logn) < n) ++logn;
  t.n = n; t.logn = logn; t.q = q; t.psi = 0; t.omega = omega % q;
  t.elem_bytes = q < ((u64)1 << 31) ? 4 : 8;
  t.k = h_bitlen(q);
  t.mu = (u64)((((unsigned __int128)1) << (2 * t.k)) / q);
  t.lazy = false; t.cg_lazy = false; t.fold_c = 0;
  const u64 omega_inv = h_powmod(t.omega, q - 2, q);             // modinv(omega_n): cg_ntt.py:9-10, :72
  t.n_inv = h_powmod(n % q, q - 2, q);                           // :74
  t.omega_pow.resize(n / 2 + 1); t.omega_inv_pow.resize(n / 2 + 1);
  u64 w = 1 % q, wi = 1 % q;
  for (u32 j = 0; j <= n / 2; ++j) { t.omega_pow[j] = w; t.omega_inv_pow[j] = wi; w = h_mulmod(w, t.omega, q); wi = h_mulmod(wi, omega_inv, q); }
  t.ninv_w1 = t.n_inv;
  return t;
}

// Tables of a GENERAL plan: nwc_poly_mult(a, b, psi_2n) for ANY psi_2n and any modulus >= 2, computed literally as
// cg_ntt.py:78-92 does — psi^i (:82-83), omega = psi^2 (:85), "modinv" = pow(v, q-2, q) whether or not that is an inverse
// (:9-10, :72, :74, :91), psi_inv^i (:92).  Nothing is validated; canonical policy (Shoup records); constant-geometry kernels only.
inline HostTables h_build_general_tables(u32 n, u64 q, u64 psi) {
  HostTables t;
  u32 logn = 0;
  while (((u32)1 << logn) < n) ++logn;
  t.n = n; t.logn = logn; t.q = q; t.psi = psi % q; t.omega = h_mulmod(t.psi, t.psi, q);
  t.elem_bytes = q < ((u64)1 << 31) ? 4 : 8;
  t.k = h_bitlen(q);
  t.mu = (u64)((((unsigned __int128)1) << (2 * t.k)) / q);
  t.lazy = false; t.cg_lazy = false; t.cg_sched = false; t.fold_c = 0;
  const u64 psi_inv = h_powmod(t.psi, q - 2, q), omega_inv = h_powmod(t.omega, q - 2, q);
  t.n_inv = h_powmod(n % q, q - 2, q);
  t.psi_pow.resize(n); t.psi_inv_pow.resize(n); t.psi_inv_ninv.resize(n);
  u64 f = 1 % q, g = 1 % q;
  for (u32 i = 0; i < n; ++i) {
    t.psi_pow[i] = f; t.psi_inv_pow[i] = g; t.psi_inv_ninv[i] = h_mulmod(g, t.n_inv, q);
    f = h_mulmod(f, t.psi, q); g = h_mulmod(g, psi_inv, q);
  }
  t.omega_pow.resize(n / 2 + 1); t.omega_inv_pow.resize(n / 2 + 1);
  u64 w = 1 % q, wi = 1 % q;
  for (u32 j = 0; j <= n / 2; ++j) { t.omega_pow[j] = w; t.omega_inv_pow[j] = wi; w = h_mulmod(w, t.omega, q); wi = h_mulmod(wi, omega_inv, q); }
  t.ninv_w1 = t.n_inv;
  return t;
}

template <typename E> inline typename TwOf<E>::type h_make_tw(u64 w, u64 q);
template <> inline Tw64 h_make_tw<u64>(u64 w, u64 q) { return h_make_tw64(w, q); }
template <> inline Tw32 h_make_tw<u32>(u64 w, u64 q) { return h_make_tw32(w, q); }

// Shoup records {w, floor(w 2^W / q)}: constant-geometry kernels, canonical policy, 32-bit lanes
template <typename E> inline std::vector<typename TwOf<E>::type> h_tw_table(const std::vector<u64>& v, u64 q) {
  std::vector<typename TwOf<E>::type> t(v.size());
  for (size_t i = 0; i < v.size(); ++i) t[i] = h_make_tw<E>(v[i], q);
  return t;
}

// Records of the FUSED kernels' tables: split constants (mul_sp_acc) when the plan is lazy with 64-bit lanes, Shoup otherwise.
inline bool h_uses_split(const HostTables& t) { return t.lazy && t.elem_bytes == 8; }
template <typename E> inline typename TwOf<E>::type h_make_fused_tw(u64 w, const HostTables& t) { return h_make_tw<E>(w, t.q); }
template <> inline Tw64 h_make_fused_tw<u64>(u64 w, const HostTables& t) {
  return h_uses_split(t) ? h_make_tw64_split(w, t.q, t.k) : h_make_tw64(w, t.q);
}
template <typename E> inline std::vector<typename TwOf<E>::type> h_fused_table(const std::vector<u64>& v, const HostTables& t) {
  std::vector<typename TwOf<E>::type> r(v.size());
  for (size_t i = 0; i < v.size(); ++i) r[i] = h_make_fused_tw<E>(v[i], t);
  return r;
}

template <typename E> inline Arith<E> h_make_arith(const HostTables& t) {
  Arith<E> ar;
  ar.q = (E)t.q; ar.mu = t.mu; ar.k = t.k; ar.fold_c = t.fold_c;
  ar.sk.mulp = 0; ar.sk.cf = 0;
  for (int K = 0; K <= 16; ++K) ar.qmul[K] = (u64)(((unsigned __int128)K * t.q) & ~(u64)0);      // K q < 2^64 wherever the schedule uses it
  if (h_uses_split(t)) {
    const int p = t.k - 31;
    ar.sk.mulp = (u32)1 << p;
    ar.sk.cf = (u32)((((unsigned __int128)1) << (p + 32)) % t.q);
  }
  ar.one = h_make_tw<E>(1, t.q);
  ar.ninv = h_make_tw<E>(t.n_inv, t.q);
  ar.fninv = h_make_fused_tw<E>(t.n_inv, t);
  ar.fninv_w1 = h_make_fused_tw<E>(t.ninv_w1, t);
  return ar;
}

}  // namespace tn
