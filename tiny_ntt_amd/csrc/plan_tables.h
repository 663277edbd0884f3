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
  if (!t.lazy) t.fold_c = 0;
  const u64 psi_inv = h_powmod(t.psi, q - 2, q);                // modinv: cg_ntt.py:9-10, :91
  const u64 omega_inv = h_powmod(t.omega, q - 2, q);            // :72
  t.n_inv = h_powmod(n % q, q - 2, q);                          // :74
  std::vector<u64>& psi_inv_pow = t.psi_inv_pow;
  psi_inv_pow.resize(n);
  t.psi_pow.resize(n); t.psi_inv_ninv.resize(n); t.psi_brv.resize(n); t.psi_inv_brv.resize(n);
  t.omega_pow.resize(n / 2); t.omega_inv_pow.resize(n / 2);
  u64 f = 1, g = 1;
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
  for (u32 j = 0; j < n / 2; ++j) { t.omega_pow[j] = w; t.omega_inv_pow[j] = wi; w = h_mulmod(w, t.omega, q); wi = h_mulmod(wi, omega_inv, q); }
  t.ninv_w1 = h_mulmod(t.n_inv, t.psi_inv_brv[1], q);
  return t;
}

template <typename E> inline typename TwOf<E>::type h_make_tw(u64 w, u64 q);
template <> inline Tw64 h_make_tw<u64>(u64 w, u64 q) { return h_make_tw64(w, q); }
template <> inline Tw32 h_make_tw<u32>(u64 w, u64 q) { return h_make_tw32(w, q); }

template <typename E> inline std::vector<typename TwOf<E>::type> h_tw_table(const std::vector<u64>& v, u64 q) {
  std::vector<typename TwOf<E>::type> t(v.size());
  for (size_t i = 0; i < v.size(); ++i) t[i] = h_make_tw<E>(v[i], q);
  return t;
}

template <typename E> inline Arith<E> h_make_arith(const HostTables& t) {
  Arith<E> ar;
  ar.q = (E)t.q; ar.mu = t.mu; ar.k = t.k; ar.fold_c = t.fold_c;
  ar.one = h_make_tw<E>(1, t.q);
  ar.ninv = h_make_tw<E>(t.n_inv, t.q);
  ar.ninv_w1 = h_make_tw<E>(t.ninv_w1, t.q);
  return ar;
}

}  // namespace tn
