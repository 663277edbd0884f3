/*
 * cg_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's golden model for the hot path
 * (new_reference/cg_ntt.py and cg_ntt_8butterfly.py in orhosko/tiny-ntt).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the shipped library (tiny_ntt_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below
 * against the tests/golden fixtures, which were produced by importing the reference
 * Python module in the build container (tests/golden/make_golden.py), and
 * against the checksums the reference C++ benchmark prints.
 *
 * Arithmetic: exact, uint64 storage, unsigned __int128 products reduced with
 * `%` — the same canonical residue Python's big-int `%` yields.  Valid for any
 * odd modulus q < 2^63 (the sums a+t stay below 2^64).
 *
 * Every function names the reference lines it follows.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define TN_ORACLE_OK 0
#define TN_ORACLE_EBADLEN 1   /* reference: ValueError("Expected N coefficients") */
#define TN_ORACLE_ENOMEM 2

static inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

/* pow(base, exp, q) — Python builtin used at cg_ntt.py:10,51,54,82,85,92 */
u64 tn_oracle_powmod(u64 base, u64 exp, u64 q) {
    u64 r = 1 % q;
    base %= q;
    while (exp) {
        if (exp & 1) r = mulmod(r, base, q);
        base = mulmod(base, base, q);
        exp >>= 1;
    }
    return r;
}

/* modinv — cg_ntt.py:9-10 (Fermat: value^(q-2) mod q, q prime) */
u64 tn_oracle_modinv(u64 value, u64 q) { return tn_oracle_powmod(value, q - 2, q); }

/* bit_reverse — cg_ntt.py:13-18 */
static size_t bit_reverse(size_t value, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (value & 1); value >>= 1; }
    return r;
}

static unsigned log2_exact(size_t n) { unsigned b = 0; while (((size_t)1 << b) < n) ++b; return b; }

/* bit_reverse_list — cg_ntt.py:21-26: reordered[rev(idx)] = values[idx] */
void tn_oracle_bit_reverse_list(const u64 *values, u64 *reordered, size_t n) {
    unsigned bits = log2_exact(n);
    for (size_t idx = 0; idx < n; ++idx) reordered[bit_reverse(idx, bits)] = values[idx];
}

/*
 * cg_ntt — cg_ntt.py:29-65.  Natural-order in, natural-order out.
 * Constant-geometry (Pease) schedule: stage s = 1..log2 n, k = n >> s,
 * omega_s = omega^k; butterfly i in [0, n/2):
 *     w = omega_s^(i / k); t = w * a[2i+1]
 *     A[i] = a[2i] + t;  A[i + n/2] = a[2i] - t        (all mod q, in [0,q))
 * `group` selects how many consecutive butterflies are issued together
 * (1 = cg_ntt.py, 8 = cg_ntt_8butterfly.py:61-89); the arithmetic is identical
 * and the pad lanes (0,0,1) of :79-83 contribute nothing that is stored.
 * `trace`, if non-NULL, receives every stage's full output (log2 n rows of n),
 * the data cg_ntt.py prints 16-at-a-time under verbose=True (:60-62).
 */
static int cg_ntt_impl(const u64 *a_prime, u64 *out, size_t n, u64 omega_n, u64 q,
                       unsigned group, u64 *trace) {
    unsigned log_n = log2_exact(n);
    if (n < 2 || ((size_t)1 << log_n) != n) return TN_ORACLE_EBADLEN;
    u64 *a = (u64 *)malloc(n * sizeof(u64));
    u64 *A = (u64 *)malloc(n * sizeof(u64));
    u64 *wtab = (u64 *)malloc((n / 2) * sizeof(u64));   /* omega_s^j, j < 2^(stage-1) */
    if (!a || !A || !wtab) { free(a); free(A); free(wtab); return TN_ORACLE_ENOMEM; }
    tn_oracle_bit_reverse_list(a_prime, a, n);                       /* :39 */
    size_t pairs = n / 2;
    for (unsigned stage = 1; stage <= log_n; ++stage) {              /* :49 */
        size_t k = n >> stage;                                       /* :50 */
        u64 omega_s = tn_oracle_powmod(omega_n, k, q);               /* :51 */
        /* pow(omega_s, i // k) of :54, tabulated once per stage (same exact values) */
        wtab[0] = 1 % q;
        for (size_t j = 1; j < (pairs + k - 1) / k; ++j) wtab[j] = mulmod(wtab[j - 1], omega_s, q);
        for (size_t base = 0; base < pairs; base += group) {
            u64 left[8], right[8], w[8];
            unsigned lanes = 0;
            for (unsigned off = 0; off < group && base + off < pairs; ++off, ++lanes) {
                size_t i = base + off;
                w[off] = wtab[i / k];                                /* :54 */
                left[off] = a[2 * i] % q;                            /* :55 */
                right[off] = a[2 * i + 1] % q;                       /* :56 */
            }
            for (unsigned off = 0; off < lanes; ++off) {
                size_t i = base + off;
                u64 t = mulmod(w[off], right[off], q);               /* :57 */
                u64 s = left[off] + t;                               /* :58 */
                A[i] = s >= q ? s - q : s;
                A[i + pairs] = left[off] >= t ? left[off] - t : left[off] + q - t;  /* :59 */
            }
        }
        if (trace) memcpy(trace + (size_t)(stage - 1) * n, A, n * sizeof(u64));
        memcpy(a, A, n * sizeof(u64));                               /* :63-64 */
    }
    if (log_n == 0) memcpy(A, a, n * sizeof(u64));
    memcpy(out, A, n * sizeof(u64));
    free(a); free(A); free(wtab);
    return TN_ORACLE_OK;
}

int tn_oracle_cg_ntt(const u64 *a_prime, u64 *out, size_t n, u64 omega_n, u64 q, u64 *trace) {
    return cg_ntt_impl(a_prime, out, n, omega_n, q, 1, trace);
}

/* cg_ntt_8butterfly — cg_ntt_8butterfly.py:41-97 */
int tn_oracle_cg_ntt_8butterfly(const u64 *a_prime, u64 *out, size_t n, u64 omega_n, u64 q, u64 *trace) {
    return cg_ntt_impl(a_prime, out, n, omega_n, q, 8, trace);
}

/* cg_intt — cg_ntt.py:68-75: cg_ntt with omega^-1, then * n^-1 */
static int cg_intt_impl(const u64 *A, u64 *out, size_t n, u64 omega_n, u64 q, unsigned group) {
    u64 omega_inv = tn_oracle_modinv(omega_n, q);                    /* :72 */
    int rc = cg_ntt_impl(A, out, n, omega_inv, q, group, NULL);      /* :73 */
    if (rc) return rc;
    u64 n_inv = tn_oracle_modinv((u64)n % q, q);                     /* :74 */
    for (size_t i = 0; i < n; ++i) out[i] = mulmod(out[i], n_inv, q);/* :75 */
    return TN_ORACLE_OK;
}

int tn_oracle_cg_intt(const u64 *A, u64 *out, size_t n, u64 omega_n, u64 q) {
    return cg_intt_impl(A, out, n, omega_n, q, 1);
}

/* cg_intt_8butterfly — cg_ntt_8butterfly.py:100-104 */
int tn_oracle_cg_intt_8butterfly(const u64 *A, u64 *out, size_t n, u64 omega_n, u64 q) {
    return cg_intt_impl(A, out, n, omega_n, q, 8);
}

/*
 * nwc_poly_mult — cg_ntt.py:78-92 (group=1) / cg_ntt_8butterfly.py:107-121 (group=8).
 * c = a*b in Z_q[x]/(x^n+1): twist by psi^i, two forward NTTs with omega=psi^2,
 * pointwise product, inverse NTT, untwist by psi^-i.
 */
static int nwc_impl(const u64 *a, const u64 *b, u64 *c, size_t n, u64 q, u64 psi_2n, unsigned group) {
    u64 *buf = (u64 *)malloc(4 * n * sizeof(u64));
    if (!buf) return TN_ORACLE_ENOMEM;
    u64 *at = buf, *bt = buf + n, *A = buf + 2 * n, *B = buf + 3 * n;
    u64 p = 1 % q;
    for (size_t i = 0; i < n; ++i) {                                 /* :82-83 */
        at[i] = mulmod(a[i] % q, p, q);
        bt[i] = mulmod(b[i] % q, p, q);
        p = mulmod(p, psi_2n % q, q);
    }
    u64 omega_n = mulmod(psi_2n % q, psi_2n % q, q);                 /* :85 */
    int rc = cg_ntt_impl(at, A, n, omega_n, q, group, NULL);         /* :86 */
    if (!rc) rc = cg_ntt_impl(bt, B, n, omega_n, q, group, NULL);    /* :87 */
    if (!rc) {
        for (size_t i = 0; i < n; ++i) A[i] = mulmod(A[i], B[i], q); /* :88 */
        rc = cg_intt_impl(A, at, n, omega_n, q, group);              /* :90 */
    }
    if (!rc) {
        u64 psi_inv = tn_oracle_modinv(psi_2n % q, q);               /* :91 */
        u64 pi = 1 % q;
        for (size_t i = 0; i < n; ++i) {                             /* :92 */
            c[i] = mulmod(at[i], pi, q);
            pi = mulmod(pi, psi_inv, q);
        }
    }
    free(buf);
    return rc;
}

int tn_oracle_nwc_poly_mult(const u64 *a, const u64 *b, u64 *c, size_t n, u64 q, u64 psi_2n) {
    return nwc_impl(a, b, c, n, q, psi_2n, 1);
}

int tn_oracle_nwc_poly_mult_8butterfly(const u64 *a, const u64 *b, u64 *c, size_t n, u64 q, u64 psi_2n) {
    return nwc_impl(a, b, c, n, q, psi_2n, 8);
}

/* Batched convenience for the parity tests: rows are independent. */
int tn_oracle_nwc_poly_mult_batch(const u64 *a, const u64 *b, u64 *c, size_t batch, size_t n, u64 q, u64 psi_2n) {
    for (size_t r = 0; r < batch; ++r) {
        int rc = nwc_impl(a + r * n, b + r * n, c + r * n, n, q, psi_2n, 1);
        if (rc) return rc;
    }
    return TN_ORACLE_OK;
}

/*
 * negacyclic_convolution — new_reference/test_cg_ntt.py:11-21 (O(n^2) schoolbook
 * with sign flip on wrap-around); also software_benchmark's
 * negacyclic_mul_reference (benchmark_ntt_60bit.cpp:167-180).
 */
void tn_oracle_negacyclic_schoolbook(const u64 *a, const u64 *b, u64 *out, size_t n, u64 q) {
    memset(out, 0, n * sizeof(u64));
    for (size_t i = 0; i < n; ++i) {
        u64 ai = a[i] % q;
        if (!ai) continue;
        for (size_t j = 0; j < n; ++j) {
            u64 term = mulmod(ai, b[j] % q, q);
            size_t k = i + j;
            if (k >= n) { k -= n; term = term ? q - term : 0; }
            u64 s = out[k] + term;
            out[k] = s >= q ? s - q : s;
        }
    }
}
