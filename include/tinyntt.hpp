// tinyntt.hpp — the functions of the reference's C++ benchmark, over the C ABI of tinyntt.h (header-only, C++17).
//
// software_benchmark/benchmark_ntt_60bit.cpp and benchmark_ntt.cpp are the reference's compiled-code callers of the
// hot path.  Their parameter set is compile-time (BENCH_N / BENCH_Q / BENCH_PSI, CMakeLists.txt:5-7); here it is the
// template arguments of tinyntt::Bench, and every function keeps the reference's name, argument meaning and
// preconditions.  The arithmetic runs on the GPU behind libtinyntt.so; nothing in this header computes a butterfly.
//
//   using B = tinyntt::Bench<std::uint64_t, 4096, 1152921504606830593ULL, 431606828070683274ULL>;
//   B::Poly a = B::make_poly(1), b = B::make_poly(2), out;
//   B::negacyclic_mul_ntt(a, b, out);                 // benchmark_ntt_60bit.cpp:148
//   std::uint64_t s = B::checksum(out);               // :182   == the value the reference binary prints
//
// Errors: the reference rejects a bad (N, Q, PSI) at compile time (static_asserts, :58-59); here the plan is created on
// first use and a rejected parameter set, a missing HIP device or a failed launch throws tinyntt::Error (what() is
// tn_last_error(), status() the tn_status).  Word must be the lane width of the modulus: uint32_t for Q < 2^31
// (benchmark_ntt.cpp), uint64_t otherwise (benchmark_ntt_60bit.cpp).
#ifndef TINYNTT_HPP
#define TINYNTT_HPP

#include <array>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "tinyntt.h"

namespace tinyntt {

class Error : public std::runtime_error {
 public:
  Error(tn_status s, const char* what) : std::runtime_error(what ? what : tn_status_string(s)), status_(s) {}
  tn_status status() const { return status_; }
 private:
  tn_status status_;
};

inline void check(tn_status s) {
  if (s != TN_OK) throw Error(s, tn_last_error());
}

template <class Word, std::size_t N, std::uint64_t Q, std::uint64_t PSI, int Device = 0>
struct Bench {
  static_assert(N > 0 && (N & (N - 1)) == 0, "N must be a power of two");                      // benchmark_ntt_60bit.cpp:17
  static_assert(sizeof(Word) == (Q < (1ULL << 31) ? 4 : 8), "Word = uint32_t for Q < 2^31, uint64_t otherwise");
  using Poly = std::array<Word, N>;                                                            // :21

  // The plan of this parameter set (tables = the reference's constexpr PsiPowers / OmegaPowers / N_INV, :43-64).
  static tn_plan* plan() {
    static Holder h;
    return h.p;
  }

  // make_poly (:79-87; 24-bit variant benchmark_ntt.cpp:82-90): the harness' LCG, on the host.
  static Poly make_poly(std::uint64_t seed) {
    Poly out{};
    std::uint64_t x = seed;
    for (auto& v : out) {
      x = 6364136223846793005ULL * x + 1442695040888963407ULL;
      v = static_cast<Word>(sizeof(Word) == 8 ? x % Q : (x >> 17) % Q);
    }
    return out;
  }

  // checksum (:182-188; the 32-bit program wraps mod 2^64 before the %, benchmark_ntt.cpp:228-233): on the host.
  static std::uint64_t checksum(const Poly& poly) {
    std::uint64_t acc = 0;
    for (Word v : poly) {
      if (sizeof(Word) == 8) acc = static_cast<std::uint64_t>((static_cast<unsigned __int128>(acc) * 1315423911ULL + v) % 0xffffffffffffffc5ULL);
      else acc = (acc * 1315423911ULL + v) % 0xffffffffffffffc5ULL;
    }
    return acc;
  }

  // negacyclic_mul_ntt (:148-159): out = a * b in Z_Q[x]/(x^N + 1); a, b in [0, Q) like the reference's mod_* helpers expect
  // (any word is accepted and taken mod Q).
  static void negacyclic_mul_ntt(const Poly& a, const Poly& b, Poly& out) { negacyclic_mul_ntt(a.data(), b.data(), out.data(), 1); }
  // The same for `batch` polynomials stored back to back (the reference loops over reps; a GPU wants them in one call).
  static void negacyclic_mul_ntt(const Word* a, const Word* b, Word* out, std::size_t batch) {
    check(tn_poly_mult_host(plan(), a, b, out, batch, TN_VARIANT_AUTO));
  }

  // forward_ntt_bench (:161-165): out = ntt<false>(twist(a)), natural order.
  static void forward_ntt_bench(const Poly& a, Poly& out) { check(tn_twisted_ntt_forward_host(plan(), a.data(), out.data(), 1, TN_VARIANT_AUTO)); }

  // ntt<Inverse> (:107-128): in-place cyclic transform with omega = PSI^2, natural order in and out; the inverse scales by N^-1.
  template <bool Inverse>
  static void ntt(Poly& a) {
    Poly t;
    check(Inverse ? tn_ntt_inverse_host(plan(), a.data(), t.data(), 1, TN_VARIANT_AUTO)
                  : tn_ntt_forward_host(plan(), a.data(), t.data(), 1, TN_VARIANT_AUTO));
    a = t;
  }

  // negacyclic_mul_reference (:167-180) / negacyclic_mul_scalar (benchmark_simple_60bit.cpp:45-58): the O(N^2) direct product,
  // computed on the device by the independent checker kernel (what --check compares against).
  static void negacyclic_mul_reference(const Poly& a, const Poly& b, Poly& out) { check(tn_schoolbook_host(plan(), a.data(), b.data(), out.data(), 1)); }

 private:
  struct Holder {
    tn_plan* p = nullptr;
    Holder() { check(tn_plan_create(&p, static_cast<std::uint32_t>(N), Q, PSI, Device, TN_PLAN_DEFAULT)); }
    ~Holder() { if (p) tn_plan_destroy(p); }
    Holder(const Holder&) = delete;
    Holder& operator=(const Holder&) = delete;
  };
};

}  // namespace tinyntt

#endif  // TINYNTT_HPP
