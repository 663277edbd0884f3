// C++17 client of include/tinyntt.hpp: the reference benchmark's main (software_benchmark/benchmark_ntt_60bit.cpp:207-254 and
// benchmark_ntt.cpp) written against the mirror — same function names, same checks.  Built with g++ by tests/test_c_abi.py.
//   exit 0: both parameter sets reproduce the reference binary's printed checksums and --check passes
//   exit 3: no HIP device (tinyntt::Error with TN_ENODEVICE; no CPU fallback)      exit 1: anything else
#include <cstdio>
#include "tinyntt.hpp"

template <class B>
static int run(const char* name, std::uint64_t want_fwd, std::uint64_t want_mul) {
  const auto a = B::make_poly(1), b = B::make_poly(2);                 // main: make_poly(1), make_poly(2)
  typename B::Poly out{}, fwd{}, ref{};
  B::negacyclic_mul_ntt(a, b, out);
  B::forward_ntt_bench(a, fwd);
  B::negacyclic_mul_reference(a, b, ref);                              // --check
  if (out != ref) { std::fprintf(stderr, "%s: check failed\n", name); return 1; }
  auto rt = fwd;                                                       // ntt<true>(ntt<false>(x)) == x
  auto x = a;
  B::template ntt<false>(x);
  B::template ntt<true>(x);
  if (x != a) { std::fprintf(stderr, "%s: ntt round trip failed\n", name); return 1; }
  std::printf("%s forward_ntt_checksum=%llu checksum=%llu\n", name, (unsigned long long)B::checksum(fwd), (unsigned long long)B::checksum(out));
  if (B::checksum(fwd) != want_fwd || B::checksum(out) != want_mul) { std::fprintf(stderr, "%s: checksum differs from the reference binary's\n", name); return 1; }
  (void)rt;
  return 0;
}

int main() {
  try {
    using B60 = tinyntt::Bench<std::uint64_t, 4096, 1152921504606830593ULL, 431606828070683274ULL>;   // rtl/ntt_poly_mult.sv:16-24
    using B24 = tinyntt::Bench<std::uint32_t, 4096, 8380417ULL, 283817ULL>;                           // CMakeLists.txt:5-7
    // values printed by the reference binaries (oracle/_ref reproduces them)
    if (run<B60>("benchmark_ntt_60bit", 15678418584317678507ULL, 2710933653778106521ULL)) return 1;
    if (run<B24>("benchmark_ntt", 2800297349529693940ULL, 11303505593119465445ULL)) return 1;
    // a parameter set the reference's static_asserts reject (:58-59): psi + 1 is not a 2N-th root
    try {
      using Bad = tinyntt::Bench<std::uint64_t, 4096, 1152921504606830593ULL, 431606828070683275ULL>;
      (void)Bad::plan();
      std::fprintf(stderr, "bad psi accepted\n");
      return 1;
    } catch (const tinyntt::Error& e) {
      if (e.status() != TN_EBADPARAM) return 1;
    }
    std::puts("cpp mirror ok");
    return 0;
  } catch (const tinyntt::Error& e) {
    std::printf("tinyntt::Error %d: %s\n", (int)e.status(), e.what());
    return e.status() == TN_ENODEVICE ? 3 : 1;
  }
}
