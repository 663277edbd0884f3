// modarith.h — exact modular arithmetic shared by the HIP kernels (device) and
// the plan builder / CPU emulation of the kernels (host).  Pure integer code.
//
// Reduction family: Barrett.  Three forms are used, all exact (bit-identical to
// the `%` the reference uses: new_reference/cg_ntt.py:57-59, :75, :88, :92;
// software_benchmark/benchmark_ntt_60bit.cpp:75-77):
//
//  * mul_tw_*   — Barrett with the quotient factor precomputed PER CONSTANT
//                 (twiddle w carries wp = floor(w * 2^W / q), W = word bits).
//                 This is the butterfly multiply: every twiddle, twist factor
//                 and scale factor is a plan-time constant.
//  * mulmod_barrett* — classic two-operand Barrett with mu = floor(2^(2k)/q),
//                 k = bitlen(q): q1 = p >> (k-1); q2 = (q1*mu) >> (k+1);
//                 r = p - q2*q; conditional subtracts.  The recipe of
//                 scripts/precompute_constants.py:38-46 and
//                 rtl/barrett_reduction.v:23-29; used for the pointwise product
//                 where both operands are data.
//  * mul_sp_*   — SPLIT-CONSTANT product for q = 2^k - c on the lazy 64-bit path (the butterfly multiply of
//                 the throughput kernel): the twiddle w is stored as w = wlo + whi 2^p and x = w 2^32 mod q =
//                 xlo + xhi 2^p with p = k - 31, so  a w == a0 w + a1 x (mod q)  is four 32x32+64 multiply-adds
//                 in two columns (weights 1 and 2^p), and the 64-bit high column H comes back with two more:
//                 lo32(H) 2^p (a shift done by the multiplier) and hi32(H) 2^(p+32) == hi32(H) 2c.  Six
//                 v_mad_u64_u32 and not one other instruction; the butterfly's "+ u" is the first addend.
//  * fold_*     — for moduli just below a power of two (q = 2^k - c, c small;
//                 both reference moduli are: 2^23-2^13+1, 2^60-2^14+1) one
//                 Barrett step with the quotient estimate x >> k:
//                 x - (x>>k)*q = (x mod 2^k) + (x>>k)*c.  Keeps lazy values bounded.
//
// gfx950 cost model (tools/ubench/valu_rates.hip, measured): v_mad_u64_u32,
// v_mul_lo/hi_u32, v_add_co/addc, v_add3, v_alignbit are all one 4-cycle issue
// slot per wave; only v_and/or/xor/add_u32/sub_u32/lshrrev/mov/cndmask are
// 2-cycle.  So a 32x32+64 multiply-add costs the same as a carry add, and the
// formulations below are mad-heavy on purpose.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TN_HD __host__ __device__ __forceinline__
#else
#define TN_HD inline
#endif

namespace tn {

typedef uint32_t u32;
typedef uint64_t u64;

// ----------------------------------------------------------------------------
// Twiddle records: the constant and its precomputed Barrett quotient factor.
struct alignas(16) Tw64 { u64 w, wp; };   // wp = floor(w * 2^64 / q); 16-byte aligned: one s_load_dwordx4 / ds_read_b128 / global_load_dwordx4 per record
struct Tw32 { u32 w, wp; };   // wp = floor(w * 2^32 / q)

template <typename E> struct TwOf;
template <> struct TwOf<u64> { typedef Tw64 type; };
template <> struct TwOf<u32> { typedef Tw32 type; };

// 64x64 -> high 64.  Device: __umul64hi (4 v_mad_u64_u32); host: __int128.
TN_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// ---- 64-bit lanes -----------------------------------------------------------
// Makes a value opaque to LLVM at this point (no instruction is emitted).  Used so that a
// chain of 32x32+64 multiply-adds whose consumer needs only the low dword is NOT narrowed
// into v_mul_lo_u32 + v_add3_u32 (6 issue slots) but stays 4 v_mad_u64_u32.
TN_HD u64 opaque64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
  return x;
}
TN_HD u32 opaque32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
  return x;
}

// The same value as far as the hardware is concerned, a fresh one as far as the compiler is: used on the thread index at
// the top of the persistent row loop so that (table or operand pointer + thread offset) is NOT a loop invariant.  Hoisted,
// each such sum is a 64-bit VGPR pair that lives across the whole loop (and ends up in scratch); kept inside the loop the
// access is "scalar base + 32-bit thread offset", which needs one VGPR for all of them.  No instruction is emitted.
TN_HD u32 opaque_copy(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
  return x;
}

// A wave-uniform zero the compiler cannot see through, (re)defined at the point of the call (one scalar move, no
// vector-ALU slot).  Adding it to a loop-invariant scalar address or index pins every load that depends on the sum
// after this point: otherwise the compiler hoists all of them out of the persistent row loop, where there are more such
// values than SGPRs and they come back through v_readlane (a vector-ALU slot each), while a scalar load that hits the
// scalar cache is free.  Host: 0.
TN_HD u32 opaque_zero() {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
#else
  return 0;
#endif
}

// Scheduling fence (device only): the machine scheduler may not move instructions across it.
// Used between groups of butterflies to bound how many are in flight (live registers).
TN_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// The same for vector-ALU work only: loads (LDS, global) and scalar instructions may still move across, so twiddle reads
// can be issued ahead of the butterflies that use them while the live temporaries of the arithmetic stay bounded.
TN_HD void sched_fence_valu() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0x0124);       // SALU | VMEM read | DS read may cross
#endif
}

// Value known to be identical in all lanes of a wave: tells the compiler so (v_readfirstlane),
// which turns loads indexed by it into scalar loads.  Host: identity.
TN_HD u32 wave_uniform(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (u32)__builtin_amdgcn_readfirstlane((int)x);
#else
  return x;
#endif
}

// reverse the low `bits` bits of v (bit_reverse, new_reference/cg_ntt.py:13-18)
TN_HD u32 bitrev(u32 v, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(v) >> (32 - bits);
#else
  u32 r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
  return r;
#endif
}

TN_HD u32 mulhi32(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }

// In {T-2, T-1, T} for T = floor(a*wp / 2^64): the high partial product plus the high
// halves of the two middle ones; what is dropped (their low halves and a0*wp0) is < 3*2^64... / 2^64 < 3,
// and floor() of the kept part can fall at most 2 below.  v_mul_hi_u32 x2 + v_mad_u64_u32 + 64-bit add.
TN_HD u64 mulhi64_lo2(u64 a, u64 wp) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
  return ((u64)a1 * p1 + mulhi32(a0, p1)) + mulhi32(a1, p0);
}

// u + a*w - qh*q (mod 2^64) with qh ~ floor(a*w/q):  the low dword pair accumulates
// a0*w0 + qh0*nq0 on top of u (nq = 2^64 - q; two mads with a 64-bit addend, so the add
// of u is free); the cross terms a0*w1 + a1*w0 + qh0*nq1 + qh1*nq0 only matter mod 2^32
// and are summed in a second mad chain whose low dword is added to the high dword.
// a: ANY u64.  Result == u + a*w (mod q); as an integer it is u + (a*w mod q) + j*q with
// j in {0,1,2,3} (Shoup's quotient is at most 1 low, mulhi64_lo2 at most 2 more),
// provided that fits in 64 bits.
TN_HD u64 mul_tw_acc(u64 u, u64 a, Tw64 t, u64 q) {
  const u64 qh = mulhi64_lo2(a, t.wp);
  const u64 nq = (u64)0 - q;
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), w0 = (u32)t.w, w1 = (u32)(t.w >> 32);
  const u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  u64 lo = (u64)a0 * w0 + u;
  lo = (u64)h0 * n0 + lo;
  u64 hi = (u64)a0 * w1;
  hi = (u64)a1 * w0 + hi;
  hi = (u64)h0 * n1 + hi;
  hi = (u64)h1 * n0 + hi;
  hi = opaque64(hi);
  // one 32-bit add on the high dword; opaque, or LLVM rewrites it as lo + (hi << 32): two moves
  // and a 64-bit add (8 issue cycles instead of 2)
  const u32 rh = opaque32((u32)(lo >> 32) + (u32)hi);
  return ((u64)rh << 32) | (u32)lo;
}

// a: ANY u64;  result == a*w (mod q), in [0, 4q).  Needs 4q <= 2^64.
TN_HD u64 mul_tw_lazy(u64 a, Tw64 t, u64 q) { return mul_tw_acc(0, a, t, q); }

TN_HD u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }

// canonical [0,q)
TN_HD u64 mul_tw(u64 a, Tw64 t, u64 q) {
  u64 r = mul_tw_lazy(a, t, q);
  r = csub(r, 2 * q);
  return csub(r, q);
}

// ---- split-constant twiddle product (lazy 64-bit lanes, q = 2^k - c, 32 <= k <= 60) ----------------------
// The 16-byte record is reused: .w = wlo | whi << 32, .wp = xlo | xhi << 32  with  w = wlo + whi 2^p,
// x = (w 2^32) mod q = xlo + xhi 2^p,  wlo, xlo < 2^p,  whi, xhi < 2^(k-p) = 2^31,  p = k - 31.
struct SplitK { u32 mulp, cf; };     // 2^p  and  2^(p+32) mod q (= 2c)
// u + a w (mod q) for ANY 64-bit a, as the integer u + t' with
//   t' = a0 wlo + a1 xlo + lo32(H) 2^p + hi32(H) cf,   H = a0 whi + a1 xhi < 2^64,
//   t' < 2^(k+1) + (a >> 32) 2^p + 2^(k+1) + 2^32 cf   (h_sp_tmax() in plan_tables.h evaluates it exactly).
// The caller's bound schedule guarantees u + t' < 2^64 (SplitSched in fused_core.h, verified for the plan's
// (k, c) on the host by h_split_sched_ok()).  mulp is a run-time value on purpose: with a literal 2^p the
// compiler turns that multiply-add into a 64-bit shift and a 64-bit add (two instructions).
#ifndef TN_SOLINAS5
#define TN_SOLINAS5 0            // developer A/B only (profiles/r2_g_five_multiply_ab.txt): 1 = the 5-multiply form below instead of the 6
#endif
#if TN_SOLINAS5
// The 5-multiply split-constant product proposed in the round-1 review, for k = 60: record {w, x = w 2^32 mod q} as plain
// 64-bit words; a w == a0 w + a1 x is four multiply-adds into a 94-bit sum S = H2 2^32 + lo32(L2) with the carries chained
// through the high column, and ONE fold at bit 62 (2^62 == 4c).  Fewer multiplies, but the column carries need glue: two
// zero-extended pairs, a 64-bit add, a mask and a funnel shift; and the butterfly's "+ u" cannot ride (the fold comes after).
TN_HD u64 mul_sp_acc(u64 u, u64 a, Tw64 t, SplitK sk) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
  const u32 w0 = (u32)t.w, w1 = (u32)(t.w >> 32), x0 = (u32)t.wp, x1 = (u32)(t.wp >> 32);
  const u64 L1 = (u64)a0 * w0;
  const u64 H1 = (u64)a0 * w1 + (L1 >> 32);
  const u64 L2 = (u64)a1 * x0 + (u32)L1;
  const u64 H2 = (u64)a1 * x1 + (H1 + (L2 >> 32));             // S < 2^94  ->  H2 < 2^62
  const u32 top = (u32)(H2 >> 30);                               // S >> 62
  const u64 low = ((u64)((u32)H2 & 0x3FFFFFFFu) << 32) | (u32)L2;   // S mod 2^62
  return u + (low + (u64)top * (sk.cf << 1));                    // cf = 2c  ->  2^62 == 4c;  product < 2^62 + 2^32 4c
}
#else
TN_HD u64 mul_sp_acc(u64 u, u64 a, Tw64 t, SplitK sk) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
  u64 L = (u64)a0 * (u32)t.w + u;
  L = (u64)a1 * (u32)t.wp + L;
  u64 H = (u64)a0 * (u32)(t.w >> 32);
  H = (u64)a1 * (u32)(t.wp >> 32) + H;
  u64 r = (u64)(u32)H * sk.mulp + L;
  r = (u64)(u32)(H >> 32) * sk.cf + r;
  return r;
}
#endif
TN_HD u64 mul_sp(u64 a, Tw64 t, SplitK sk) { return mul_sp_acc(0, a, t, sk); }

// Two-operand Barrett (A9).  a, b in [0, q); k = bitlen(q) in [2, 62]; mu = floor(2^(2k)/q) (<= k+1 bits).
TN_HD u64 mulmod_barrett(u64 a, u64 b, u64 q, u64 mu, int k) {
  u64 plo = a * b, phi = mulhi64(a, b);                       // p < 2^(2k)
  u64 q1 = (k - 1 == 0) ? plo : ((phi << (64 - (k - 1))) | (plo >> (k - 1)));   // p >> (k-1), <= k+1 bits
  u64 mlo = q1 * mu, mhi = mulhi64(q1, mu);                   // q1*mu < 2^(2k+2)
  u64 q2 = (mhi << (64 - (k + 1))) | (mlo >> (k + 1));        // >> (k+1)
  u64 r = plo - q2 * q;                                       // true value in [0, 3q)
  r = csub(r, q);
  return csub(r, q);                                          // second subtract kept: single-subtract bound unproven
}

// One Barrett step with quotient estimate x >> k, for q = 2^k - c (k >= 32):  result == x (mod q),
// < 2^k + (x >> k) * c.  Written on the high dword so it is 2 two-cycle ops + one v_mad_u64_u32.
TN_HD u64 fold(u64 x, int k, u32 c) {
  const u32 top = (u32)(x >> 32) >> (k - 32);
  const u64 lowmask = ((u64)((1u << (k - 32)) - 1u) << 32) | 0xFFFFFFFFull;   // k >= 32: the low dword needs no masking
  return (x & lowmask) + (u64)top * c;
}

// Two-operand product for q = 2^k - c with SMALL c (2^k == c mod q), used by the pointwise step of the lazy policy:
// the 128-bit product is split at bit k and the high part folded back twice.  9 multiplies instead of the
// 18 of a two-operand Barrett.  a: output of fold() (< 2^k + 2^(64-k) c); b: any value < 14q.  Result == a*b (mod q), < 2q.
// Every intermediate bound is verified for the plan's (k, c) on the host: h_pw_fast_ok() in plan_tables.h.
TN_HD u64 mulmod_solinas_lazy(u64 a, u64 b, int k, u32 c) {
  const int s = k - 32;                                          // 0 <= s <= 28
  const u32 ms = (1u << s) - 1u;
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 m0 = (u64)a0 * b0;
  const u64 m1 = (u64)a0 * b1 + (m0 >> 32);
  const u64 m2 = (u64)a1 * b0 + (u32)m1;
  const u64 m3 = (u64)a1 * b1 + (m1 >> 32) + (m2 >> 32);         // P = m3 2^64 + lo32(m2) 2^32 + lo32(m0)
  const u32 p0 = (u32)m0, p1 = (u32)m2;
  const u32 ph0 = (u32)((((u64)(u32)m3 << 32) | p1) >> s);       // Ph = P >> k  (two dwords)
  const u32 ph1 = (u32)(m3 >> s);
  const u64 pl = ((u64)(p1 & ms) << 32) | p0;                    // Pl = P mod 2^k
  const u64 t = (u64)ph0 * c + pl;                               // P == Pl + Ph c = t + (ph1 c) 2^32
  const u64 u = (u64)ph1 * c;
  const u32 uh = (u32)(u >> s);                                  // (u 2^32) >> k
  const u32 vh = (u32)(t >> 32) + ((u32)u & ms);                 // high dword of V = t + ((u 2^32) mod 2^k); low dword = t's
  const u32 top = (vh >> s) + uh;                                // (V >> k) + ((u 2^32) >> k)
  return (((u64)(vh & ms) << 32) | (u32)t) + (u64)top * c;
}

// ---- 32-bit lanes -----------------------------------------------------------
// a: ANY u32; result == a*w (mod q) in [0, 2q).  Needs 2q < 2^32.
TN_HD u32 mul_tw_lazy(u32 a, Tw32 t, u32 q) {
  u32 qh = (u32)(((u64)a * t.wp) >> 32);                      // exact high half: in {Q-1, Q}
  return a * t.w - qh * q;
}

TN_HD u32 csub(u32 x, u32 q) { return x >= q ? x - q : x; }

TN_HD u32 mul_tw(u32 a, Tw32 t, u32 q) { return csub(mul_tw_lazy(a, t, q), q); }

// a, b in [0,q), k = bitlen(q) <= 31, mu = floor(2^(2k)/q) (<= k+1 bits <= 32)
TN_HD u32 mulmod_barrett(u32 a, u32 b, u32 q, u64 mu, int k) {
  u64 p = (u64)a * b;
  u64 q1 = p >> (k - 1);
  u64 q2 = (q1 * mu) >> (k + 1);                              // q1, mu <= 32 bits: product fits 64
  u32 r = (u32)p - (u32)q2 * q;
  r = csub(r, q);
  return csub(r, q);
}

TN_HD u32 mulmod_barrett_lazy(u32 a, u32 b, u32 q, u64 mu, int k) {
  u64 p = (u64)a * b;
  u64 q1 = p >> (k - 1);
  u64 q2 = (q1 * mu) >> (k + 1);
  return (u32)p - (u32)q2 * q;
}

TN_HD u32 fold(u32 x, int k, u32 c) {
  u32 lowmask = (((u32)1) << k) - 1;
  return (x & lowmask) + (x >> k) * c;
}

// ---- host-side helpers (plan building; exact, slow path is fine) -------------
inline u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((unsigned __int128)a * b) % q); }
inline u64 h_powmod(u64 b, u64 e, u64 q) {
  u64 r = 1 % q; b %= q;
  while (e) { if (e & 1) r = h_mulmod(r, b, q); b = h_mulmod(b, b, q); e >>= 1; }
  return r;
}
inline int h_bitlen(u64 x) { int n = 0; while (x) { ++n; x >>= 1; } return n; }
inline Tw64 h_make_tw64(u64 w, u64 q) {
  Tw64 t; t.w = w; t.wp = (u64)((((unsigned __int128)w) << 64) / q); return t;
}
inline Tw64 h_make_tw64_split(u64 w, u64 q, int k) {       // see mul_sp_acc
#if TN_SOLINAS5
  Tw64 t5; t5.w = w; t5.wp = (u64)((((unsigned __int128)w) << 32) % q); (void)k; return t5;
#endif
  const int p = k - 31;
  const u64 x = (u64)((((unsigned __int128)w) << 32) % q), m = (((u64)1) << p) - 1;
  Tw64 t; t.w = (w & m) | ((w >> p) << 32); t.wp = (x & m) | ((x >> p) << 32); return t;
}
inline u64 h_split_value(Tw64 t, int k) { return TN_SOLINAS5 ? t.w : (u64)(u32)t.w + ((t.w >> 32) << (k - 31)); }   // w back from its record
inline Tw32 h_make_tw32(u64 w, u64 q) {
  Tw32 t; t.w = (u32)w; t.wp = (u32)((w << 32) / q); return t;
}
// deterministic Miller-Rabin for 64-bit (bases cover all n < 2^64)
inline bool h_is_prime(u64 n) {
  if (n < 2) return false;
  static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (u64 p : small) { if (n == p) return true; if (n % p == 0) return false; }
  u64 d = n - 1; int s = 0;
  while ((d & 1) == 0) { d >>= 1; ++s; }
  for (u64 a : small) {
    u64 x = h_powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int i = 1; i < s; ++i) { x = h_mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
    if (comp) return false;
  }
  return true;
}

}  // namespace tn
