// emu_kernels.cpp — TEST INFRASTRUCTURE: steps the product kernels' per-thread
// code (tiny_ntt_amd/csrc/fused_core.h, modarith.h, plan_tables.h — the very
// headers the gfx950 kernels are compiled from) on the CPU, one emulated thread
// at a time with barriers between phases, so index maps, LDS layouts, table
// semantics and lazy-reduction bounds can be checked in the GPU-less build
// container.  Built with g++ by tests/emu/Makefile; loaded by tests/test_emu.py.
// It is not a product path and nothing in tiny_ntt_amd/ loads it.
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../tiny_ntt_amd/csrc/plan_tables.h"
#include "../../tiny_ntt_amd/csrc/cg_core.h"

using namespace tn;

namespace {

template <typename E, int LOGN, int LPT, bool LAZY, bool CIN = false>
int fused_polymul_emu(const HostTables& t, const u64* a, const u64* b, u64* c, size_t batch, bool cyclic) {
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  typedef Policy<E, LAZY, CIN> Pol;
  typedef typename TwOf<E>::type Tw;
  Arith<E> ar = h_make_arith<E>(t);
  if (cyclic) ar.fninv_w1 = ar.fninv;                      // as launch_fused_t: product in Z_q[x]/(x^n - 1)
  const std::vector<Tw> psi_brv = h_fused_table<E>(cyclic ? t.cyc_brv : t.psi_brv, t),
                        psi_inv_brv = h_fused_table<E>(cyclic ? t.cyc_inv_brv : t.psi_inv_brv, t);
  std::vector<E> lds(Cfg::lds_elems());
  struct Regs { E x[Cfg::R]; };
  std::vector<Regs> xa(Cfg::THREADS), xb(Cfg::THREADS);

  // host copies of what the kernel stages in LDS / prefetches into registers
  std::vector<Tw> lds_fwd(psi_brv.begin() + Cfg::lds_tw_lo(), psi_brv.begin() + Cfg::lds_tw_hi());
  std::vector<Tw> lds_inv(psi_inv_brv.begin() + Cfg::lds_tw_lo(), psi_inv_brv.begin() + Cfg::lds_tw_hi());
  struct Pre { Tw t[Cfg::NPRE]; };
  std::vector<Pre> pre(Cfg::THREADS);

  auto forward = [&](std::vector<Regs>& x) {
    for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) tw_prefetch<E, Cfg>(pre[tau].t, tau, psi_brv.data());
    static_for<0, Cfg::PHASES>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) {
        const TwRefs<E> tw = {psi_brv.data(), lds_fwd.data(), pre[tau].t};
        fwd_phase<E, Cfg, Pol, p>(x[tau].x, tau, tw, ar);
      }
      if constexpr (p + 1 < Cfg::PHASES) {
        for (auto& v : lds) v = (E)0xDEADBEEFu;
        for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) ex_store<E, Cfg, p, p>(x[tau].x, tau, lds.data());
        for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) ex_load<E, Cfg, p, p + 1>(x[tau].x, tau, lds.data());
      }
    });
  };
  auto inverse = [&](std::vector<Regs>& x) {
    for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) tw_prefetch<E, Cfg>(pre[tau].t, tau, psi_inv_brv.data());
    static_for<0, Cfg::PHASES>([&](auto i_) {
      constexpr int p = Cfg::PHASES - 1 - decltype(i_)::value;
      for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) {
        const TwRefs<E> tw = {psi_inv_brv.data(), lds_inv.data(), pre[tau].t};
        inv_phase<E, Cfg, Pol, p>(x[tau].x, tau, tw, ar);
      }
      if constexpr (p > 0) {
        for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) ex_store<E, Cfg, p - 1, p>(x[tau].x, tau, lds.data());
        for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) ex_load<E, Cfg, p - 1, p - 1>(x[tau].x, tau, lds.data());
      }
    });
  };

  for (size_t row = 0; row < batch; ++row) {
    const size_t off = row << LOGN;
    for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau)
    {
      for (int r = 0; r < Cfg::R; ++r) {
        xa[tau].x[r] = (E)a[off + Cfg::jidx(0, tau, r)];
        xb[tau].x[r] = (E)b[off + Cfg::jidx(0, tau, r)];
      }
      load_reduce<E, Cfg, Pol>(xa[tau].x, ar);
      load_reduce<E, Cfg, Pol>(xb[tau].x, ar);
    }
    forward(xa);
    forward(xb);
    for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau) pointwise<E, Cfg, Pol>(xa[tau].x, xb[tau].x, ar);
    inverse(xa);
    for (u32 tau = 0; tau < (u32)Cfg::THREADS; ++tau)
      for (int r = 0; r < Cfg::R; ++r) c[off + Cfg::jidx(0, tau, r)] = xa[tau].x[r];
  }
  return 0;
}

// Same steps as ntt_fused_kernel (kernels.hip), one emulated thread at a time.
template <typename E, int LOGN, int LPT, bool LAZY>
int fused_ntt_emu(const HostTables& t, int mode, const u64* in, u64* out) {
  typedef FusedCfg<E, LOGN, LPT> Cfg;
  typedef Policy<E, LAZY> Pol;
  typedef typename TwOf<E>::type Tw;
  Arith<E> ar = h_make_arith<E>(t);
  if (mode == 2) ar.fninv_w1 = ar.fninv;                    // cyc_inv_brv[1] = 1 (as launch_nttf_t does)
  const std::vector<Tw> tab = h_fused_table<E>(mode == 2 ? t.cyc_inv_brv : (mode == 1 ? t.cyc_brv : t.psi_brv), t);
  std::vector<Tw> lds_tab(tab.begin() + Cfg::lds_tw_lo(), tab.begin() + Cfg::lds_tw_hi());
  std::vector<E> lds(Cfg::lds_elems());
  struct Regs { E x[Cfg::R]; };
  struct Pre { Tw t[Cfg::NPRE]; };
  std::vector<Regs> x(Cfg::THREADS);
  std::vector<Pre> pre(Cfg::THREADS);
  constexpr int LAST = Cfg::PHASES - 1;
  const u32 T = Cfg::THREADS;
  for (u32 tau = 0; tau < T; ++tau) tw_prefetch<E, Cfg>(pre[tau].t, tau, tab.data());
  if (mode == 2) {
    for (u32 tau = 0; tau < T; ++tau) for (int r = 0; r < Cfg::R; ++r) lds[Cfg::nat_addr(Cfg::jidx(0, tau, r))] = Pol::load((E)in[Cfg::jidx(0, tau, r)], ar);
    for (u32 tau = 0; tau < T; ++tau) for (int r = 0; r < Cfg::R; ++r) x[tau].x[r] = lds[Cfg::nat_addr(bitrev(Cfg::jidx(LAST, tau, r), LOGN))];
    static_for<0, Cfg::PHASES>([&](auto i_) {
      constexpr int p = Cfg::PHASES - 1 - decltype(i_)::value;
      for (u32 tau = 0; tau < T; ++tau) { const TwRefs<E> tw = {tab.data(), lds_tab.data(), pre[tau].t}; inv_phase<E, Cfg, Pol, p>(x[tau].x, tau, tw, ar); }
      if constexpr (p > 0) {
        for (u32 tau = 0; tau < T; ++tau) ex_store<E, Cfg, p - 1, p>(x[tau].x, tau, lds.data());
        for (u32 tau = 0; tau < T; ++tau) ex_load<E, Cfg, p - 1, p - 1>(x[tau].x, tau, lds.data());
      }
    });
    for (u32 tau = 0; tau < T; ++tau) for (int r = 0; r < Cfg::R; ++r) out[Cfg::jidx(0, tau, r)] = x[tau].x[r];
    return 0;
  }
  for (u32 tau = 0; tau < T; ++tau) for (int r = 0; r < Cfg::R; ++r) {
    const u32 j = Cfg::jidx(0, tau, r);
    x[tau].x[r] = (LAZY && r >= Cfg::R / 2) ? (E)in[j] : Pol::load((E)in[j], ar);     // as ntt_fused_kernel: only the "u" half is reduced
  }
  static_for<0, Cfg::PHASES>([&](auto p_) {
    constexpr int p = decltype(p_)::value;
    for (u32 tau = 0; tau < T; ++tau) { const TwRefs<E> tw = {tab.data(), lds_tab.data(), pre[tau].t}; fwd_phase<E, Cfg, Pol, p>(x[tau].x, tau, tw, ar); }
    if constexpr (p + 1 < Cfg::PHASES) {
      for (u32 tau = 0; tau < T; ++tau) ex_store<E, Cfg, p, p>(x[tau].x, tau, lds.data());
      for (u32 tau = 0; tau < T; ++tau) ex_load<E, Cfg, p, p + 1>(x[tau].x, tau, lds.data());
    }
  });
  for (u32 tau = 0; tau < T; ++tau) for (int r = 0; r < Cfg::R; ++r)
    lds[Cfg::nat_addr(bitrev(Cfg::jidx(LAST, tau, r), LOGN))] = LAZY ? Pol::canon(x[tau].x[r], ar) : x[tau].x[r];
  for (u32 tau = 0; tau < T; ++tau) for (int r = 0; r < Cfg::R; ++r) out[Cfg::jidx(0, tau, r)] = lds[Cfg::nat_addr(Cfg::jidx(0, tau, r))];
  return 0;
}

template <typename E, bool LAZY>
int fused_ntt_dispatch(const HostTables& t, int mode, const u64* in, u64* out) {
  switch (t.logn) {
    case 8: return fused_ntt_emu<E, 8, fused_lpt(8), LAZY>(t, mode, in, out);
    case 9: return fused_ntt_emu<E, 9, fused_lpt(9), LAZY>(t, mode, in, out);
    case 10: return fused_ntt_emu<E, 10, fused_lpt(10), LAZY>(t, mode, in, out);
    case 11: return fused_ntt_emu<E, 11, fused_lpt(11), LAZY>(t, mode, in, out);
    case 12: return fused_ntt_emu<E, 12, fused_lpt(12), LAZY>(t, mode, in, out);
    case 13: return fused_ntt_emu<E, 13, fused_lpt(13), LAZY>(t, mode, in, out);
    default: return 7;
  }
}

template <typename E, bool LAZY>
int fused_dispatch(const HostTables& t, const u64* a, const u64* b, u64* c, size_t batch, bool cyclic) {
  switch (t.logn) {
    case 8: return fused_polymul_emu<E, 8, fused_lpt(8), LAZY>(t, a, b, c, batch, cyclic);
    case 9: return fused_polymul_emu<E, 9, fused_lpt(9), LAZY>(t, a, b, c, batch, cyclic);
    case 10: return fused_polymul_emu<E, 10, fused_lpt(10), LAZY>(t, a, b, c, batch, cyclic);
    case 11: return fused_polymul_emu<E, 11, fused_lpt(11), LAZY>(t, a, b, c, batch, cyclic);
    case 12: return fused_polymul_emu<E, 12, fused_lpt(12), LAZY>(t, a, b, c, batch, cyclic);
    case 13: return fused_polymul_emu<E, 13, fused_lpt(13), LAZY>(t, a, b, c, batch, cyclic);
    default: return 7;
  }
}

// Sequential restatement of cg_kernel's arithmetic with the product's tables
// and modarith (thread mapping there is a plain strided loop).
template <typename E>
int cg_emu(const HostTables& t, int mode, const u64* a, const u64* b, u64* out, u64* trace) {
  typedef typename TwOf<E>::type Tw;
  const Arith<E> ar = h_make_arith<E>(t);
  const u32 n = t.n, logn = t.logn, pairs = n / 2;
  const std::vector<Tw> omega = h_tw_table<E>(t.omega_pow, t.q), omega_inv = h_tw_table<E>(t.omega_inv_pow, t.q),
                        psi_pow = h_tw_table<E>(t.psi_pow, t.q), psi_inv_ninv = h_tw_table<E>(t.psi_inv_ninv, t.q);
  auto load_brv = [&](const u64* in, const Tw* tw) {
    std::vector<E> buf(n);
    for (u32 i = 0; i < n; ++i) buf[h_brv(i, logn)] = tw ? mul_tw((E)in[i], tw[i], ar.q) : mul_tw((E)in[i], ar.one, ar.q);
    return buf;
  };
  auto stages = [&](std::vector<E> src, const Tw* tab, u64* tr) {
    std::vector<E> dst(n);
    for (u32 stage = 1; stage <= logn; ++stage) {
      const u32 k = n >> stage;
      for (u32 i = 0; i < pairs; ++i) {
        const E left = src[2 * i], right = src[2 * i + 1];
        const E tt = mul_tw(right, tab[i & ~(k - 1)], ar.q);
        dst[i] = csub((E)(left + tt), ar.q);
        dst[i + pairs] = left >= tt ? (E)(left - tt) : (E)(left + (ar.q - tt));
      }
      if (tr) for (u32 i = 0; i < n; ++i) tr[(size_t)(stage - 1) * n + i] = dst[i];
      src.swap(dst);
    }
    return src;
  };
  if (mode == 0 || mode == 3) {
    auto r = stages(load_brv(a, mode == 3 ? psi_pow.data() : nullptr), omega.data(), trace);
    for (u32 i = 0; i < n; ++i) out[i] = r[i];
  } else if (mode == 1) {
    auto r = stages(load_brv(a, nullptr), omega_inv.data(), nullptr);
    for (u32 i = 0; i < n; ++i) out[i] = mul_tw(r[i], ar.ninv, ar.q);
  } else {
    auto ra = stages(load_brv(a, psi_pow.data()), omega.data(), nullptr);
    auto rb = stages(load_brv(b, psi_pow.data()), omega.data(), nullptr);
    std::vector<E> cc(n);
    for (u32 i = 0; i < n; ++i) cc[h_brv(i, logn)] = mulmod_barrett(ra[i], rb[i], ar.q, ar.mu, ar.k);
    auto rc = stages(cc, omega_inv.data(), nullptr);
    for (u32 i = 0; i < n; ++i) out[i] = mul_tw(rc[i], psi_inv_ninv[i], ar.q);
  }
  return 0;
}


// The constant-geometry kernels' trips (cg_core.h / cg_kernels.hip), one emulated lane-step at a time with the LDS image and
// the LDS twiddle table between them.  mode: 0 cg_ntt, 1 cg_intt, 2 nwc_poly_mult, 3 twist + cg_ntt, 4 cyclic product.
// flags: bit 0 = wave-uniform trips take their twiddles from the global table (scalar loads in the kernel) instead of LDS;
//        bit 1 = no reversal: the inverse transform of a product reads a re-staged inverse table (plans whose omega is not a root)
template <typename E, int GROUP, int LAYOUT, int AM>
int cgm_emu(const HostTables& t, int mode, int flags, const u64* a, const u64* b, u64* out, u64* trace) {
  typedef CgGeom<GROUP> Ge;
  typedef CgMap<E, GROUP, LAYOUT> M;
  typedef CgArith<E, AM> A;
  typedef typename TwOf<E>::type Tw;
  constexpr int R = Ge::R, L = Ge::L;
  const u32 n = t.n, logn = t.logn;
  if ((int)logn < L) return 7;
  if (trace && (AM == CGA_SPLIT_LAZY || AM == CGA_SPLIT_SCHED)) return 7;
  if (AM == CGA_SPLIT_SCHED && !t.cg_sched) return 7;
  const Arith<E> ar = h_make_arith<E>(t);
  const u32 TP = n >> L, ntrips = Ge::ntrips(logn), r1 = Ge::first_stages(logn);
  const bool split = AM != CGA_SHOUP;
  auto table = [&](const std::vector<u64>& v) { return split ? h_fused_table<E>(v, t) : h_tw_table<E>(v, t.q); };
  const std::vector<Tw> fwd = table(t.omega_pow), inv = table(t.omega_inv_pow), psi_pow = table(t.psi_pow), psi_inv_ninv = table(t.psi_inv_ninv);
  const Tw ninv = split ? ar.fninv : ar.ninv;
  const bool big = n >= 512 && TP >= 256;
  std::vector<E> img(M::span(n) + 4, (E)0xDEADBEEFu);
  std::vector<Tw> ldstab(n / 2 + 1);
  auto stage_table = [&](const std::vector<Tw>& g) { for (u32 j = 0; j <= n / 2; ++j) ldstab[cg_twmap<GROUP, LAYOUT>(j, big)] = g[j]; };
  struct Regs { E x[R]; Tw w0[GROUP]; };
  constexpr bool CAN_MERGE = TN_CG_MERGE_TWIST && A::LAZY;           // cg_kernel_impl.h: the twist of stage 1's right inputs rides the butterflies

  // x: registers in bit-reversed order (x[brvL(e')] = element ls + e' TP of the input list) -> natural order (x[e] = output ls + e TP)
  auto transform = [&](std::vector<Regs>& x, bool inverse, bool rev, u64* tr, bool merged = false) {
    const Tw* glob = inverse ? inv.data() : fwd.data();
    auto run_first = [&](auto nst_) {
      constexpr int NST = decltype(nst_)::value;
      for (u32 ls = 0; ls < TP; ++ls) {
        const u32 T = h_brv(ls, logn - L);
        auto tw = [&](auto j_, auto h_) { return glob[(u32)decltype(h_)::value * (n >> (decltype(j_)::value + 1))]; };
        auto after = [&](auto j_) { if (tr) for (u32 e = 0; e < (u32)R; ++e) tr[(size_t)decltype(j_)::value * n + Ge::pos(logn, decltype(j_)::value + 1, T, e)] = x[ls].x[e]; };
        if constexpr (CAN_MERGE) {
          if (merged) cg_trip<E, GROUP, AM, NST, false, 0, true>(x[ls].x, ar, tw, after, [&](auto g_) { return x[ls].w0[decltype(g_)::value]; });
          else cg_trip<E, GROUP, AM, NST, false>(x[ls].x, ar, tw, after);
        } else cg_trip<E, GROUP, AM, NST, false>(x[ls].x, ar, tw, after);
        if (ntrips > 1) for (u32 e = 0; e < (u32)R; ++e) img[M::at(Ge::pos(logn, NST, T, e))] = x[ls].x[e];
      }
    };
    switch (r1) { case 1: run_first(std::integral_constant<int, 1>()); break;
                  case 2: if constexpr (L >= 2) run_first(std::integral_constant<int, 2>()); break;
                  case 3: if constexpr (L >= 3) run_first(std::integral_constant<int, 3>()); break;
                  case 4: if constexpr (L >= 4) run_first(std::integral_constant<int, 4>()); break; }
    u32 s0 = r1;
    for (u32 trip = 1; trip < ntrips; ++trip, s0 += L) {
      for (u32 ls = 0; ls < TP; ++ls) for (u32 e = 0; e < (u32)R; e += 2) {
        const CgPair<E> v = *reinterpret_cast<const CgPair<E>*>(&img[M::step(M::at(R * ls), e)]);
        x[ls].x[e] = v.lo; x[ls].x[e + 1] = v.hi;
      }
      for (auto& v : img) v = (E)0xDEADBEEFu;
      const bool uniform = (flags & 1) && (int)logn - (int)s0 - L >= 6;     // the trip's twiddles depend on T >> 6 only
      for (u32 ls = 0; ls < TP; ++ls) {
        const u32 T = ls, base0 = cg_tw_base0<GROUP>(logn, s0, T);
        auto after = [&](auto j_) { if (tr) for (u32 e = 0; e < (u32)R; ++e) tr[(size_t)(s0 + decltype(j_)::value) * n + Ge::pos(logn, decltype(j_)::value + 1, T, e)] = x[ls].x[e]; };
        auto idx = [&](int j, u32 h) { return h * (n >> (j + 1)) + (base0 >> j); };
        auto run = [&](auto par_) {
          constexpr int PAR = decltype(par_)::value;
          if (uniform) cg_trip<E, GROUP, AM, L, false, PAR>(x[ls].x, ar, [&](auto j_, auto h_) { return glob[idx(decltype(j_)::value, decltype(h_)::value)]; }, after);
          else if (rev) cg_trip<E, GROUP, AM, L, true, PAR>(x[ls].x, ar, [&](auto j_, auto h_) { return ldstab[cg_twmap<GROUP, LAYOUT>(n / 2 - idx(decltype(j_)::value, decltype(h_)::value), big)]; }, after);
          else cg_trip<E, GROUP, AM, L, false, PAR>(x[ls].x, ar, [&](auto j_, auto h_) { return ldstab[cg_twmap<GROUP, LAYOUT>(idx(decltype(j_)::value, decltype(h_)::value), big)]; }, after);
        };
        if (AM == CGA_SPLIT_SCHED && (s0 & 1u)) run(std::integral_constant<int, 1>()); else run(std::integral_constant<int, 0>());
      }
      if (trip + 1 < ntrips) for (u32 ls = 0; ls < TP; ++ls) for (u32 e = 0; e < (u32)R; ++e) img[M::at(Ge::pos(logn, L, ls, e))] = x[ls].x[e];
    }
  };
  auto load = [&](std::vector<Regs>& x, const u64* in, const Tw* twist) {       // returns: the first stage is the merged one
    for (u32 ls = 0; ls < TP; ++ls) for (u32 e = 0; e < (u32)R; ++e) {
      const u32 i = ls + e * TP;
      if (twist && CAN_MERGE && e >= (u32)GROUP) { x[ls].x[Ge::brvL(e)] = (E)in[i]; x[ls].w0[Ge::brvL(e) >> 1] = twist[i]; }
      else x[ls].x[Ge::brvL(e)] = twist ? A::in_mul((E)in[i], twist[i], ar) : A::in_red((E)in[i], ar);
    }
    return twist && CAN_MERGE;
  };
  std::vector<Regs> xa(TP), xb(TP);
  if (mode == 0 || mode == 3) {
    stage_table(fwd);
    const bool m = load(xa, a, mode == 3 ? psi_pow.data() : nullptr);
    transform(xa, false, false, trace, m);
    for (u32 ls = 0; ls < TP; ++ls) for (u32 e = 0; e < (u32)R; ++e) out[ls + e * TP] = A::out_canon(xa[ls].x[e], ar);
  } else if (mode == 1) {
    stage_table(inv);
    load(xa, a, nullptr);
    transform(xa, true, false, nullptr);
    for (u32 ls = 0; ls < TP; ++ls) for (u32 e = 0; e < (u32)R; ++e) out[ls + e * TP] = A::out_mul(xa[ls].x[e], ninv, ar);
  } else {
    const Tw* twist = mode == 4 ? nullptr : psi_pow.data();
    const bool rev = !(flags & 2);
    stage_table(fwd);
    const bool ma = load(xa, a, twist); transform(xa, false, false, nullptr, ma);
    const bool mb = load(xb, b, twist); transform(xb, false, false, nullptr, mb);
    for (u32 ls = 0; ls < TP; ++ls) {
      E c[R];
      for (u32 e = 0; e < (u32)R; ++e) c[e] = A::pointwise(xa[ls].x[e], xb[ls].x[e], ar);
      for (u32 e = 0; e < (u32)R; ++e) xa[ls].x[Ge::brvL(e)] = c[e];
    }
    if (!rev) stage_table(inv);
    transform(xa, true, rev, nullptr);
    for (u32 ls = 0; ls < TP; ++ls) for (u32 e = 0; e < (u32)R; ++e) {
      const u32 i = ls + e * TP;
      out[i] = A::out_mul(xa[ls].x[e], twist ? psi_inv_ninv[i] : ninv, ar);
    }
  }
  return 0;
}

template <typename E, int AM>
int cgm_dispatch(const HostTables& t, int mode, int group, int layout, int flags, const u64* a, const u64* b, u64* out, u64* trace) {
#define TN_CGM(G, LY) if (group == G && layout == LY) return cgm_emu<E, G, LY, AM>(t, mode, flags, a, b, out, trace);
  TN_CGM(1, 0) TN_CGM(1, 1) TN_CGM(1, 2) TN_CGM(2, 0) TN_CGM(2, 1) TN_CGM(2, 2)
  TN_CGM(4, 0) TN_CGM(4, 1) TN_CGM(4, 2) TN_CGM(8, 0) TN_CGM(8, 1) TN_CGM(8, 2)
#undef TN_CGM
  return 7;
}

// Layout probes for the LDS bank-conflict simulator (tests/test_lds_banks.py).
// what: 0 THREADS, 1 R, 2 PHASES, 3 lds_elems, 4 ex_wave_local(arg0), 5 jidx(arg0=phase, arg1=tau, arg2=r),
//       6 ex_addr(arg0=exchange, arg1=j), 7 pos(arg0)
template <typename E, int LOGN, int LPT> static long cfg_probe(int what, unsigned a0, unsigned a1, unsigned a2) {
  typedef FusedCfg<E, LOGN, LPT> C;
  switch (what) {
    case 0: return C::THREADS;
    case 1: return C::R;
    case 2: return C::PHASES;
    case 3: return C::lds_elems();
    case 4: return C::ex_wave_local((int)a0);
    case 5: return C::jidx((int)a0, a1, a2);
    case 6: return C::ex_addr((int)a0, a1);
    case 7: return C::pos((int)a0);
  }
  return -1;
}
bool params_ok(u32 n, u64 q, u64 psi) {
  if (n < 4 || (n & (n - 1)) || q < 3 || !(q & 1) || q >= ((u64)1 << 62)) return false;
  return h_powmod(psi % q, n, q) == q - 1;
}

}  // namespace

extern "C" {

// 0 ok, 2 bad params, 7 unsupported n.  Coefficients travel as uint64 regardless of lane width.
// flags: bit 0 = canonical policy, bit 1 = cyclic product (x^n - 1) instead of negacyclic, bit 2 = inputs promised canonical.
int emu_fused_poly_mult(uint32_t n, uint64_t q, uint64_t psi, int flags, const uint64_t* a, const uint64_t* b,
                        uint64_t* c, size_t batch) {
  if (!params_ok(n, q, psi)) return 2;
  const bool cyc = (flags & 2) != 0;
  const HostTables t = h_build_tables(n, q, psi, !(flags & 1));
  if (flags & 4) {                       // promised-canonical inputs (TN_PLAN_CANONICAL_INPUTS): n = 4096, lazy 64-bit lanes, schedule replayed
    if (!(t.cin_ok && t.logn == 12 && t.lazy && t.elem_bytes == 8) || cyc) return 7;
    return fused_polymul_emu<u64, 12, fused_lpt(12), true, true>(t, a, b, c, batch, false);
  }
  if (t.elem_bytes == 8) return t.lazy ? fused_dispatch<u64, true>(t, a, b, c, batch, cyc) : fused_dispatch<u64, false>(t, a, b, c, batch, cyc);
  return t.lazy ? fused_dispatch<u32, true>(t, a, b, c, batch, cyc) : fused_dispatch<u32, false>(t, a, b, c, batch, cyc);
}

// mode: 0 twist + forward (natural out), 1 cg_ntt, 2 cg_intt — the register-tiled standalone transforms
int emu_fused_ntt(uint32_t n, uint64_t q, uint64_t psi, int force_canonical, int mode, const uint64_t* in, uint64_t* out) {
  if (!params_ok(n, q, psi)) return 2;
  const HostTables t = h_build_tables(n, q, psi, !force_canonical);
  if (t.elem_bytes == 8) return t.lazy ? fused_ntt_dispatch<u64, true>(t, mode, in, out) : fused_ntt_dispatch<u64, false>(t, mode, in, out);
  return t.lazy ? fused_ntt_dispatch<u32, true>(t, mode, in, out) : fused_ntt_dispatch<u32, false>(t, mode, in, out);
}

int emu_is_lazy(uint32_t n, uint64_t q, uint64_t psi) {
  if (!params_ok(n, q, psi)) return -1;
  return h_build_tables(n, q, psi, true).lazy ? 1 : 0;
}

// mode: 0 cg_ntt, 1 cg_intt, 2 nwc_poly_mult, 3 twist + cg_ntt.  trace may be NULL ([logn][n] otherwise, mode 0/3).
int emu_cg(uint32_t n, uint64_t q, uint64_t psi, int mode, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t* trace) {
  if (!params_ok(n, q, psi)) return 2;
  const HostTables t = h_build_tables(n, q, psi, true);
  return t.elem_bytes == 8 ? cg_emu<u64>(t, mode, a, b, out, trace) : cg_emu<u32>(t, mode, a, b, out, trace);
}

// The trips of cg_kernels.hip stepped on the CPU.  am: 0 Shoup records, 1 split records canonical, 2 split records lazy
// 3 split records lazy with the static fold schedule (1, 2, 3 need a plan that is lazy with 64-bit lanes; 3 an even log2 n).  7 = unsupported combination.
int emu_cgm(uint32_t n, uint64_t q, uint64_t psi, int mode, int group, int layout, int am, int flags, const uint64_t* a,
            const uint64_t* b, uint64_t* out, uint64_t* trace) {
  if (!params_ok(n, q, psi)) return 2;
  const HostTables t = h_build_tables(n, q, psi, true);
  if (am < 0 || am > 3 || (am != 0 && !(t.lazy && t.elem_bytes == 8 && (am == 1 || t.cg_lazy)))) return 7;
  if (t.elem_bytes == 8) {
    if (am == 0) return cgm_dispatch<u64, CGA_SHOUP>(t, mode, group, layout, flags, a, b, out, trace);
    if (am == 1) return cgm_dispatch<u64, CGA_SPLIT_CANON>(t, mode, group, layout, flags, a, b, out, trace);
    if (am == 2) return cgm_dispatch<u64, CGA_SPLIT_LAZY>(t, mode, group, layout, flags, a, b, out, trace);
    return cgm_dispatch<u64, CGA_SPLIT_SCHED>(t, mode, group, layout, flags, a, b, out, trace);
  }
  return cgm_dispatch<u32, CGA_SHOUP>(t, mode, group, layout, flags, a, b, out, trace);
}

// Layout probes of the constant-geometry kernels (64-bit lanes): what 0 = CgMap::at(x), 1 = cg_twmap(x) (n >= 512), 2 = CgMap::span(x)
long emu_cgm_probe(int group, int layout, int what, unsigned x) {
#define TN_CGP(G, LY) if (group == G && layout == LY) return what == 0 ? (long)CgMap<u64, G, LY>::at(x) : what == 1 ? (long)cg_twmap<G, LY>(x, true) : (long)CgMap<u64, G, LY>::span(x);
  TN_CGP(1, 0) TN_CGP(1, 1) TN_CGP(1, 2) TN_CGP(2, 0) TN_CGP(2, 1) TN_CGP(2, 2)
  TN_CGP(4, 0) TN_CGP(4, 1) TN_CGP(4, 2) TN_CGP(8, 0) TN_CGP(8, 1) TN_CGP(8, 2)
#undef TN_CGP
  return -1;
}

// see cfg_probe() above for `what`
long emu_cfg_probe(int logn, int elem_bytes, int what, unsigned a0, unsigned a1, unsigned a2) {
  if (elem_bytes == 8) {
    if (logn == 8) return cfg_probe<u64, 8, fused_lpt(8)>(what, a0, a1, a2);
    if (logn == 9) return cfg_probe<u64, 9, fused_lpt(9)>(what, a0, a1, a2);
    if (logn == 10) return cfg_probe<u64, 10, fused_lpt(10)>(what, a0, a1, a2);
    if (logn == 11) return cfg_probe<u64, 11, fused_lpt(11)>(what, a0, a1, a2);
    if (logn == 12) return cfg_probe<u64, 12, fused_lpt(12)>(what, a0, a1, a2);
    if (logn == 13) return cfg_probe<u64, 13, fused_lpt(13)>(what, a0, a1, a2);
  } else {
    if (logn == 8) return cfg_probe<u32, 8, fused_lpt(8)>(what, a0, a1, a2);
    if (logn == 9) return cfg_probe<u32, 9, fused_lpt(9)>(what, a0, a1, a2);
    if (logn == 10) return cfg_probe<u32, 10, fused_lpt(10)>(what, a0, a1, a2);
    if (logn == 11) return cfg_probe<u32, 11, fused_lpt(11)>(what, a0, a1, a2);
    if (logn == 12) return cfg_probe<u32, 12, fused_lpt(12)>(what, a0, a1, a2);
    if (logn == 13) return cfg_probe<u32, 13, fused_lpt(13)>(what, a0, a1, a2);
  }
  return -1;
}

// Direct probes of the arithmetic primitives (for property tests).
uint64_t emu_mul_tw64(uint64_t a, uint64_t w, uint64_t q) { return mul_tw(a, h_make_tw64(w, q), q); }
uint64_t emu_mul_tw64_lazy(uint64_t a, uint64_t w, uint64_t q) { return mul_tw_lazy(a, h_make_tw64(w, q), q); }
uint32_t emu_mul_tw32(uint32_t a, uint32_t w, uint32_t q) { return mul_tw(a, h_make_tw32(w, q), q); }
uint64_t emu_barrett64(uint64_t a, uint64_t b, uint64_t q) {
  const int k = h_bitlen(q);
  return mulmod_barrett(a, b, q, (u64)((((unsigned __int128)1) << (2 * k)) / q), k);
}
uint32_t emu_barrett32(uint32_t a, uint32_t b, uint32_t q) {
  const int k = h_bitlen(q);
  return mulmod_barrett(a, b, q, (u64)((((unsigned __int128)1) << (2 * k)) / q), k);
}
uint64_t emu_fold64(uint64_t x, uint64_t q) { const int k = h_bitlen(q); return fold(x, k, (u32)((((u64)1) << k) - q)); }
// lazy 64-bit pointwise product on two ARBITRARY words (folds them first, as pointwise() does); -1 if (k, c) is not admissible
int emu_pw_fast_ok(uint64_t q) { const int k = h_bitlen(q); return h_pw_fast_ok(q, k, (((u64)1) << k) - q) ? 1 : 0; }
uint64_t emu_pointwise_lazy64(uint64_t a, uint64_t b, uint64_t q) {
  const int k = h_bitlen(q);
  Arith<u64> ar; ar.q = q; ar.k = k; ar.fold_c = (u32)((((u64)1) << k) - q); ar.mu = 0;
  return pointwise_lazy(a, b, ar);
}
// split-constant product (lazy 64-bit lanes): u + a*w as mul_sp_acc computes it; *ok = 0 if (k, c) is not admissible
uint64_t emu_mul_sp_acc(uint64_t u, uint64_t a, uint64_t w, uint64_t q) {
  const int k = h_bitlen(q), p = k - 31;
  SplitK sk; sk.mulp = (u32)1 << p; sk.cf = (u32)((((unsigned __int128)1) << (p + 32)) % q);
  return mul_sp_acc(u, a, h_make_tw64_split(w, q, k), sk);
}
int emu_split_sched_ok(uint32_t logn, uint64_t q) { const int k = h_bitlen(q); return h_split_sched_ok(logn, k, (((u64)1) << k) - q) ? 1 : 0; }
// schedule statistics for one shape: what 0 folds in a forward transform per thread, 1 folds in an inverse, 2 forward output bound (units of 2^k/4096), 3 pw_fold_b
long emu_split_sched_stat(int logn, int what) {
  auto stat = [&](auto cfg) -> long {
    typedef decltype(cfg) C; typedef SplitSched<C> S;
    long f = 0, i = 0;
    for (int s = 0; s < C::LOGN; ++s) for (int r = 0; r < C::R; ++r) { f += S::D.ffold[s][r]; i += S::D.ifold[s][r]; }
    return what == 0 ? f : what == 1 ? i : what == 2 ? S::D.fout : (long)S::D.pw_fold_b;
  };
  switch (logn) {
    case 8: return stat(FusedCfg<u64, 8, fused_lpt(8)>());
    case 9: return stat(FusedCfg<u64, 9, fused_lpt(9)>());
    case 10: return stat(FusedCfg<u64, 10, fused_lpt(10)>());
    case 11: return stat(FusedCfg<u64, 11, fused_lpt(11)>());
    case 12: return stat(FusedCfg<u64, 12, fused_lpt(12)>());
    case 13: return stat(FusedCfg<u64, 13, fused_lpt(13)>());
  }
  return -1;
}
uint32_t emu_fold32(uint32_t x, uint32_t q) { const int k = h_bitlen(q); return fold(x, k, (u32)((((u64)1) << k) - q)); }

}  // extern "C"
