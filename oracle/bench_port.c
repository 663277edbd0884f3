/*
 * bench_port.c — CPU ORACLE #2 / CPU baseline "port" (test infrastructure).
 *
 * Restates, in plain C with run-time parameters, the algorithm that the
 * reference's C++ benchmark times (software_benchmark/benchmark_ntt_60bit.cpp
 * and benchmark_ntt.cpp in orhosko/tiny-ntt): table twiddles, in-place
 * bit-reversal + iterative Cooley-Tukey, `%`-reduced products, twist ->
 * 2x forward -> pointwise -> inverse (+ n^-1) -> inverse twist; together with
 * its deterministic input generator and its output checksum.
 *
 * Used by: tests/ (second, independent oracle; LCG inputs + checksums G1-G3 of
 * SURVEY.md §8c), bench.py's cpu_baseline leg when oracle/_ref is absent
 * (kind "port").  Never linked into the shipped library.
 *
 * Parity status: PINNED by the reference's own printed checksums (G1-G3),
 * asserted in tests/test_oracle.py.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct {
    size_t n;
    u64 q, psi;
    u64 n_inv;
    u64 *psi_pow, *psi_inv_pow, *omega_pow, *omega_inv_pow;   /* A11: .cpp:61-64 */
} tn_port_plan;

static inline u64 mm(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }          /* mod_mul :75-77 */
static inline u64 madd(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; } /* mod_add :66-69 */
static inline u64 msub(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }        /* mod_sub :71-73 */

static u64 pw(u64 b, u64 e, u64 q) {                                 /* pow_mod :27-37 */
    u64 r = 1;
    while (e) { if (e & 1) r = mm(r, b, q); b = mm(b, b, q); e >>= 1; }
    return r;
}

static u64 *power_table(u64 root, size_t n, u64 q) {                 /* make_power_table :43-51 */
    u64 *t = (u64 *)malloc(n * sizeof(u64));
    u64 v = 1;
    for (size_t i = 0; t && i < n; ++i) { t[i] = v; v = mm(v, root, q); }
    return t;
}

/* Returns NULL unless n is a power of two and psi^n == -1 (the static_asserts at :58-59). */
tn_port_plan *tn_port_plan_create(size_t n, u64 q, u64 psi) {
    if (n < 2 || (n & (n - 1)) || q < 3 || pw(psi % q, n, q) != q - 1) return NULL;
    tn_port_plan *p = (tn_port_plan *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->n = n; p->q = q; p->psi = psi % q;
    u64 omega = mm(p->psi, p->psi, q);                                /* :53 */
    u64 psi_inv = pw(p->psi, q - 2, q), omega_inv = pw(omega, q - 2, q);  /* :54-55 */
    p->n_inv = pw((u64)n % q, q - 2, q);                              /* :56 */
    p->psi_pow = power_table(p->psi, n, q);
    p->psi_inv_pow = power_table(psi_inv, n, q);
    p->omega_pow = power_table(omega, n, q);
    p->omega_inv_pow = power_table(omega_inv, n, q);
    return p;
}

void tn_port_plan_destroy(tn_port_plan *p) {
    if (!p) return;
    free(p->psi_pow); free(p->psi_inv_pow); free(p->omega_pow); free(p->omega_inv_pow); free(p);
}

/* make_poly — 60-bit: v = x % Q (.cpp:79-87); <=32-bit: v = (x >> 17) % Q (benchmark_ntt.cpp:82-90) */
void tn_port_make_poly(u64 seed, u64 *out, size_t n, u64 q) {
    u64 x = seed;
    int narrow = q < ((u64)1 << 32);
    for (size_t i = 0; i < n; ++i) {
        x = 6364136223846793005ULL * x + 1442695040888963407ULL;
        out[i] = narrow ? (x >> 17) % q : x % q;
    }
}

/* checksum — 60-bit: 128-bit intermediate (.cpp:182-188); <=32-bit: wraps mod 2^64 first (benchmark_ntt.cpp:228-233) */
u64 tn_port_checksum(const u64 *poly, size_t n, u64 q) {
    const u64 M = 0xffffffffffffffc5ULL;
    int narrow = q < ((u64)1 << 32);
    u64 acc = 0;
    for (size_t i = 0; i < n; ++i)
        acc = narrow ? (acc * 1315423911ULL + poly[i]) % M
                     : (u64)(((u128)acc * 1315423911ULL + poly[i]) % M);
    return acc;
}

static void bitrev_permute(u64 *a, size_t n) {                       /* :89-105 */
    for (size_t i = 0, j = 0; i < n; ++i) {
        if (i < j) { u64 t = a[i]; a[i] = a[j]; a[j] = t; }
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j |= bit;
    }
}

/* ntt<Inverse> — :107-128: bit-reverse, then len = 2,4,..,n with w = table[j * (n/len)] */
static void ntt_inplace(const tn_port_plan *p, u64 *a, int inverse) {
    const size_t n = p->n; const u64 q = p->q;
    const u64 *table = inverse ? p->omega_inv_pow : p->omega_pow;
    bitrev_permute(a, n);
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, step = n / len;
        for (size_t base = 0; base < n; base += len)
            for (size_t j = 0; j < half; ++j) {
                u64 u = a[base + j];
                u64 v = mm(a[base + j + half], table[j * step], q);
                a[base + j] = madd(u, v, q);
                a[base + j + half] = msub(u, v, q);
            }
    }
    if (inverse) for (size_t i = 0; i < n; ++i) a[i] = mm(a[i], p->n_inv, q);   /* :123-127 */
}

/* Untwisted transforms, natural order in/out (for cg_ntt / cg_intt parity). */
void tn_port_ntt(const tn_port_plan *p, const u64 *in, u64 *out, int inverse) {
    if (out != in) memcpy(out, in, p->n * sizeof(u64));
    ntt_inplace(p, out, inverse);
}

/* forward_ntt_bench — :161-165: copy, twist, forward */
void tn_port_forward_ntt_bench(const tn_port_plan *p, const u64 *a, u64 *out) {
    for (size_t i = 0; i < p->n; ++i) out[i] = mm(a[i], p->psi_pow[i], p->q);    /* twist :130-134 */
    ntt_inplace(p, out, 0);
}

/* negacyclic_mul_ntt — :148-159 */
int tn_port_negacyclic_mul_ntt(const tn_port_plan *p, const u64 *a, const u64 *b, u64 *out) {
    const size_t n = p->n; const u64 q = p->q;
    u64 *rhs = (u64 *)malloc(n * sizeof(u64));
    if (!rhs) return 2;
    for (size_t i = 0; i < n; ++i) {
        out[i] = mm(a[i], p->psi_pow[i], q);
        rhs[i] = mm(b[i], p->psi_pow[i], q);
    }
    ntt_inplace(p, out, 0);
    ntt_inplace(p, rhs, 0);
    for (size_t i = 0; i < n; ++i) out[i] = mm(out[i], rhs[i], q);               /* pointwise :142-146 */
    ntt_inplace(p, out, 1);
    for (size_t i = 0; i < n; ++i) out[i] = mm(out[i], p->psi_inv_pow[i], q);    /* inverse_twist :136-140 */
    free(rhs);
    return 0;
}

int tn_port_negacyclic_mul_batch(const tn_port_plan *p, const u64 *a, const u64 *b, u64 *c, size_t batch) {
    for (size_t r = 0; r < batch; ++r) {
        int rc = tn_port_negacyclic_mul_ntt(p, a + r * p->n, b + r * p->n, c + r * p->n);
        if (rc) return rc;
    }
    return 0;
}

/*
 * Timed loop in the reference's style (main :225-239): the same pair every rep,
 * steady clock, average ns.  Returns avg ns per poly-mult; *fwd_avg_ns gets the
 * forward_ntt_bench average.  Checksums are written so runs can be diffed.
 */
double tn_port_time_reps(const tn_port_plan *p, int reps, double *fwd_avg_ns, u64 *fwd_checksum, u64 *checksum) {
    const size_t n = p->n;
    u64 *a = (u64 *)malloc(3 * n * sizeof(u64));
    if (!a) return -1.0;
    u64 *b = a + n, *out = a + 2 * n;
    tn_port_make_poly(1, a, n, p->q);
    tn_port_make_poly(2, b, n, p->q);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r) tn_port_forward_ntt_bench(p, a, out);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (fwd_avg_ns) *fwd_avg_ns = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / reps;
    if (fwd_checksum) *fwd_checksum = tn_port_checksum(out, n, p->q);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r) tn_port_negacyclic_mul_ntt(p, a, b, out);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double avg = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / reps;
    if (checksum) *checksum = tn_port_checksum(out, n, p->q);
    free(a);
    return avg;
}

/*
 * SURVEY.md §8(d) item 3, the all-core figure: `rows` DISJOINT rows of the synthetic batch the GPU gets (global row r =
 * make_poly(2r+1) x make_poly(2r+2)), generated before the clock starts, multiplied once each per pass.  Returns avg ns per
 * poly-mult; *xor_checksum = XOR of the rows' checksums of the last pass (row 0 alone reproduces the reference's printed one).
 */
double tn_port_time_rows(const tn_port_plan *p, size_t first_row, size_t rows, int passes, u64 *xor_checksum) {
    const size_t n = p->n;
    u64 *a = (u64 *)malloc(3 * rows * n * sizeof(u64));
    if (!a) return -1.0;
    u64 *b = a + rows * n, *out = a + 2 * rows * n;
    for (size_t r = 0; r < rows; ++r) {
        tn_port_make_poly(2 * (first_row + r) + 1, a + r * n, n, p->q);
        tn_port_make_poly(2 * (first_row + r) + 2, b + r * n, n, p->q);
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < passes; ++k) tn_port_negacyclic_mul_batch(p, a, b, out, rows);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    u64 x = 0;
    for (size_t r = 0; r < rows; ++r) x ^= tn_port_checksum(out + r * n, n, p->q);
    if (xor_checksum) *xor_checksum = x;
    free(a);
    return ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / ((double)passes * (double)rows);
}

#ifdef TN_PORT_MAIN
/* CLI with the reference's key=value output (main :241-248) and exit codes 0/1/2. */
int main(int argc, char **argv) {
    size_t n = 4096; u64 q = 1152921504606830593ULL, psi = 431606828070683274ULL;
    int reps = 10, check = 0;
    size_t rows = 0, first_row = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--check")) check = 1;
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) { reps = atoi(argv[++i]); if (reps < 1) reps = 1; }
        else if (!strcmp(argv[i], "--n") && i + 1 < argc) n = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--q") && i + 1 < argc) q = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--psi") && i + 1 < argc) psi = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--rows") && i + 1 < argc) rows = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--first-row") && i + 1 < argc) first_row = strtoull(argv[++i], NULL, 10);
        else { fprintf(stderr, "usage: bench_port [--check] [--reps count] [--n N --q Q --psi PSI] [--rows R [--first-row F]]\n"); return 2; }
    }
    tn_port_plan *p = tn_port_plan_create(n, q, psi);
    if (!p) { fprintf(stderr, "bad parameters: psi^n != -1 mod q or n not a power of two\n"); return 2; }
    if (check) {
        extern void tn_oracle_negacyclic_schoolbook(const u64 *, const u64 *, u64 *, size_t, u64);
        u64 *a = (u64 *)malloc(4 * n * sizeof(u64)), *b = a + n, *c = a + 2 * n, *r = a + 3 * n;
        tn_port_make_poly(1, a, n, q); tn_port_make_poly(2, b, n, q);
        tn_oracle_negacyclic_schoolbook(a, b, r, n, q);
        tn_port_negacyclic_mul_ntt(p, a, b, c);
        if (memcmp(c, r, n * sizeof(u64))) { fprintf(stderr, "correctness check failed\n"); return 1; }
        free(a);
    }
    if (rows) {                      /* disjoint rows of the GPU's batch, `reps` passes over them */
        u64 x;
        double avg_rows = tn_port_time_rows(p, first_row, rows, reps, &x);
        if (avg_rows < 0) { fprintf(stderr, "out of memory\n"); return 1; }
        printf("bench_port\nN=%zu Q=%llu reps=%d rows=%zu first_row=%zu\n", n, (unsigned long long)q, reps, rows, first_row);
        printf("avg_ns=%.0f\nxor_checksum=%llu\n", avg_rows, (unsigned long long)x);
        tn_port_plan_destroy(p);
        return 0;
    }
    double fwd; u64 fc, cs;
    double avg = tn_port_time_reps(p, reps, &fwd, &fc, &cs);
    printf("bench_port\nN=%zu Q=%llu reps=%d\n", n, (unsigned long long)q, reps);
    printf("forward_ntt_total_ns=%.0f\nforward_ntt_avg_ns=%.0f\nforward_ntt_checksum=%llu\n", fwd * reps, fwd, (unsigned long long)fc);
    printf("total_ns=%.0f\navg_ns=%.0f\nchecksum=%llu\n", avg * reps, avg, (unsigned long long)cs);
    tn_port_plan_destroy(p);
    return 0;
}
#endif
