// Test infrastructure: the CPU stepping of the kernels' per-thread code (emu_kernels.cpp: the product's own headers compiled for the
// host) run under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the GPU pool, so this is where an
// out-of-bounds LDS image / table index or a shift by the word size in fused_core.h / cg_core.h / plan_tables.h would show.
// Built and run by tests/test_emu.py::test_kernel_bodies_under_address_and_ub_sanitizers (tests/emu/Makefile: sanitize).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

extern "C" {
int emu_fused_poly_mult(uint32_t n, uint64_t q, uint64_t psi, int flags, const uint64_t* a, const uint64_t* b, uint64_t* c, size_t batch);
int emu_fused_ntt(uint32_t n, uint64_t q, uint64_t psi, int force_canonical, int mode, const uint64_t* in, uint64_t* out);
int emu_cg(uint32_t n, uint64_t q, uint64_t psi, int mode, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t* trace);
int emu_cgm(uint32_t n, uint64_t q, uint64_t psi, int mode, int group, int layout, int am, int flags, const uint64_t* a, const uint64_t* b,
            uint64_t* out, uint64_t* trace);
}

namespace {
uint64_t lcg = 0x9E3779B97F4A7C15ull;
uint64_t next() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return lcg; }
uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((unsigned __int128)a * b) % q); }
uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) { uint64_t r = 1; b %= q; while (e) { if (e & 1) r = mulmod(r, b, q); b = mulmod(b, b, q); e >>= 1; } return r; }
int failures = 0, checks = 0;
void expect(bool ok, const char* what, unsigned n, int a, int b, int c, int d) {
  ++checks;
  if (!ok) { ++failures; std::fprintf(stderr, "MISMATCH %s n=%u (%d %d %d %d)\n", what, n, a, b, c, d); }
}
}  // namespace

int main() {
  struct P { uint32_t n; uint64_t q, psi; };
  // psi of the full-size sets squared down to smaller n (psi^(N/n) is a primitive 2n-th root)
  const P base24 = {4096, 8380417ull, 283817ull}, base60 = {4096, 1152921504606830593ull, 431606828070683274ull};
  std::vector<P> sets;
  for (uint32_t n : {16u, 64u, 256u, 1024u, 4096u}) {
    sets.push_back({n, base24.q, powmod(base24.psi, base24.n / n, base24.q)});
    sets.push_back({n, base60.q, powmod(base60.psi, base60.n / n, base60.q)});
  }
  for (const P& p : sets) {
    const uint32_t n = p.n;
    std::vector<uint64_t> a(n), b(n), ref(n), out(n), t1(n), t2(n);
    for (uint32_t i = 0; i < n; ++i) { a[i] = next(); b[i] = next(); }            // any 64-bit word: the kernels reduce
    if (p.q < (1ull << 32)) for (uint32_t i = 0; i < n; ++i) { a[i] &= 0xffffffffull; b[i] &= 0xffffffffull; }
    a[0] = p.q - 1; b[0] = p.q - 1; a[1] = 0; b[n - 1] = ~0ull >> (p.q < (1ull << 32) ? 32 : 0);
    // reference for this row: the one-stage-per-trip constant-geometry stepping (mode 2 = nwc_poly_mult)
    const int rc = emu_cg(n, p.q, p.psi, 2, a.data(), b.data(), ref.data(), nullptr);
    expect(rc == 0, "emu_cg rc", n, rc, 0, 0, 0);
    if (n >= 256) {
      for (int flags : {0, 1}) {
        std::fill(out.begin(), out.end(), 0);
        const int r = emu_fused_poly_mult(n, p.q, p.psi, flags, a.data(), b.data(), out.data(), 1);
        expect(r == 0 && out == ref, "fused product", n, flags, r, 0, 0);
      }
      for (int fc : {0, 1}) {
        int r = emu_fused_ntt(n, p.q, p.psi, fc, 1, a.data(), t1.data());
        r |= emu_fused_ntt(n, p.q, p.psi, fc, 2, t1.data(), t2.data());
        bool ok = r == 0;
        for (uint32_t i = 0; i < n && ok; ++i) ok = t2[i] == a[i] % p.q;
        expect(ok, "fused ntt round trip", n, fc, r, 0, 0);
        r = emu_fused_ntt(n, p.q, p.psi, fc, 0, a.data(), t1.data());
        expect(r == 0, "fused twisted ntt", n, fc, r, 0, 0);
      }
    }
    for (int group : {1, 2, 4, 8}) for (int layout : {0, 1, 2}) for (int am : {0, 1, 2, 3}) for (int flags : {0, 1, 3}) {
      std::fill(out.begin(), out.end(), 0);
      const int r = emu_cgm(n, p.q, p.psi, 2, group, layout, am, flags, a.data(), b.data(), out.data(), nullptr);
      if (r == 7) continue;                                                        // combination not built (e.g. split records on 32-bit lanes)
      expect(r == 0 && out == ref, "cg trips product", n, group, layout, am, flags);
      int r2 = emu_cgm(n, p.q, p.psi, 0, group, layout, am, flags, a.data(), nullptr, t1.data(), nullptr);
      r2 |= emu_cgm(n, p.q, p.psi, 1, group, layout, am, flags, t1.data(), nullptr, t2.data(), nullptr);
      bool ok = r2 == 0;
      for (uint32_t i = 0; i < n && ok; ++i) ok = t2[i] == a[i] % p.q;
      expect(ok, "cg trips ntt round trip", n, group, layout, am, flags);
      const int r3 = emu_cgm(n, p.q, p.psi, 3, group, layout, am, flags, a.data(), nullptr, t1.data(), nullptr);
      expect(r3 == 0, "cg trips twisted ntt", n, group, layout, am, flags);
      if (am <= 1) {                                                               // per-stage traces (canonical arithmetic)
        unsigned logn = 0; while ((1u << logn) < n) ++logn;
        std::vector<uint64_t> tr((size_t)logn * n);
        const int r4 = emu_cgm(n, p.q, p.psi, 0, group, layout, am, flags, a.data(), nullptr, t1.data(), tr.data());
        expect(r4 == 0, "cg trips trace", n, group, layout, am, flags);
      }
    }
  }
  std::printf("sanitize_driver: %zu parameter sets, %d checks, %d failures\n", sets.size(), checks, failures);
  return failures ? 1 : 0;
}
