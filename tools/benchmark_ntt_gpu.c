/* benchmark_ntt_gpu — the reference's software_benchmark CLI on the GPU, through the C ABI only (C99).
 *
 * Same flags and the same key=value report as software_benchmark/benchmark_ntt_60bit.cpp (main, :190-240):
 *     benchmark_ntt_gpu [--check] [--simple] [--reps count] [--batch rows] [--n N --q Q --psi PSI]
 * so its output diffs against the reference binary's: the checksums are those of row 0, which is the
 * reference's own pair make_poly(1) x make_poly(2); the *_avg_ns lines are per product / per transform
 * (total time of reps launches divided by reps * batch), with the operands resident in HBM.
 * Row r of the batch is make_poly(2r+1) x make_poly(2r+2) (generated on the device).
 * --check compares the first rows with the O(n^2) direct product (negacyclic_mul_reference, :167-180) on device.
 * --simple is the other benchmark family (software_benchmark/benchmark_simple_60bit.cpp, benchmark_simple.cpp): it times
 * the O(n^2) direct product itself (negacyclic_mul_scalar, :45-58) and prints that program's five lines.
 *
 * Build:  gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/benchmark_ntt_gpu.c \
 *             -L tiny_ntt_amd/lib -ltinyntt -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o benchmark_ntt_gpu
 * (make -C tiny_ntt_amd/csrc benchmark_ntt_gpu)
 */
#define _POSIX_C_SOURCE 199309L
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "tinyntt.h"

static double now_ns(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

#define TN_CHECK(call) do { tn_status s_ = (call); if (s_ != TN_OK) { fprintf(stderr, "%s: %s (%s)\n", #call, tn_last_error(), tn_status_string(s_)); return 1; } } while (0)
#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    /* defaults = software_benchmark/CMakeLists.txt:5-7 (60-bit target), rtl/ntt_poly_mult.sv:18-19 */
    uint64_t n = 4096, q = 1152921504606830593ULL, psi = 431606828070683274ULL;
    long reps = 100, batch = 65536;
    int check = 0, simple = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--check")) check = 1;
        else if (!strcmp(argv[i], "--simple")) simple = 1;
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) { reps = atol(argv[++i]); if (reps < 1) reps = 1; }
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) { batch = atol(argv[++i]); if (batch < 1) batch = 1; }
        else if (!strcmp(argv[i], "--n") && i + 1 < argc) n = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--q") && i + 1 < argc) q = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--psi") && i + 1 < argc) psi = strtoull(argv[++i], NULL, 10);
        else { fprintf(stderr, "usage: benchmark_ntt_gpu [--check] [--simple] [--reps count] [--batch rows] [--n N --q Q --psi PSI]\n"); return 2; }
    }
    tn_plan *plan = NULL;
    tn_status st = tn_plan_create(&plan, (uint32_t)n, q, psi, 0, TN_PLAN_DEFAULT);
    if (st != TN_OK) { fprintf(stderr, "tn_plan_create: %s (%s)\n", tn_last_error(), tn_status_string(st)); return st == TN_ENODEVICE ? 3 : 2; }
    const size_t bytes = (size_t)batch * n * tn_plan_elem_bytes(plan);
    void *a = NULL, *b = NULL, *c = NULL, *f = NULL;
    uint64_t *sums = NULL;
    HIP_CHECK(hipMalloc(&a, bytes)); HIP_CHECK(hipMalloc(&b, bytes)); HIP_CHECK(hipMalloc(&c, bytes)); HIP_CHECK(hipMalloc(&f, bytes));
    HIP_CHECK(hipMalloc((void **)&sums, 2 * sizeof(uint64_t)));
    TN_CHECK(tn_fill_lcg_dev(plan, a, (size_t)batch, 1, 2, NULL));      /* row r: make_poly(2r+1) */
    TN_CHECK(tn_fill_lcg_dev(plan, b, (size_t)batch, 2, 2, NULL));      /*        make_poly(2r+2) */
    TN_CHECK(tn_plan_synchronize(plan));

    if (simple) {                                   /* benchmark_simple*.cpp main (:84-108): reps x negacyclic_mul_scalar, checksum of the result */
        TN_CHECK(tn_schoolbook_dev(plan, a, b, c, (size_t)batch, NULL));
        TN_CHECK(tn_plan_synchronize(plan));
        const double s0 = now_ns();
        for (long r = 0; r < reps; ++r) TN_CHECK(tn_schoolbook_dev(plan, a, b, c, (size_t)batch, NULL));
        TN_CHECK(tn_plan_synchronize(plan));
        const double sns = now_ns() - s0;
        uint64_t hs;
        TN_CHECK(tn_checksum_rows_dev(plan, c, sums, 1, NULL));
        TN_CHECK(tn_plan_synchronize(plan));
        HIP_CHECK(hipMemcpy(&hs, sums, sizeof hs, hipMemcpyDeviceToHost));
        printf("benchmark_simple_gpu\nN=%llu Q=%llu reps=%ld batch=%ld\ntotal_ns=%.0f\navg_ns=%.2f\nchecksum=%llu\n", (unsigned long long)n,
               (unsigned long long)q, reps, batch, sns, sns / ((double)reps * (double)batch), (unsigned long long)hs);
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(f); (void)hipFree(sums);
        tn_plan_destroy(plan);
        return 0;
    }
    printf("benchmark_ntt_gpu\nN=%llu Q=%llu reps=%ld batch=%ld\n", (unsigned long long)n, (unsigned long long)q, reps, batch);

    /* forward_ntt_bench (:161-165): twist + forward transform of a */
    TN_CHECK(tn_twisted_ntt_forward_dev(plan, a, f, (size_t)batch, TN_VARIANT_AUTO, NULL));
    TN_CHECK(tn_plan_synchronize(plan));
    double t0 = now_ns();
    for (long r = 0; r < reps; ++r) TN_CHECK(tn_twisted_ntt_forward_dev(plan, a, f, (size_t)batch, TN_VARIANT_AUTO, NULL));
    TN_CHECK(tn_plan_synchronize(plan));
    double fwd_ns = now_ns() - t0;

    /* negacyclic_mul_ntt (:148-159) */
    TN_CHECK(tn_poly_mult_dev(plan, a, b, c, (size_t)batch, TN_VARIANT_AUTO, NULL));
    TN_CHECK(tn_plan_synchronize(plan));
    t0 = now_ns();
    for (long r = 0; r < reps; ++r) TN_CHECK(tn_poly_mult_dev(plan, a, b, c, (size_t)batch, TN_VARIANT_AUTO, NULL));
    TN_CHECK(tn_plan_synchronize(plan));
    double mul_ns = now_ns() - t0;

    uint64_t h[2];
    TN_CHECK(tn_checksum_rows_dev(plan, f, sums, 1, NULL));
    TN_CHECK(tn_checksum_rows_dev(plan, c, sums + 1, 1, NULL));
    TN_CHECK(tn_plan_synchronize(plan));
    HIP_CHECK(hipMemcpy(h, sums, sizeof h, hipMemcpyDeviceToHost));

    if (check) {
        const size_t rows = batch < 4 ? (size_t)batch : 4, cb = rows * n * tn_plan_elem_bytes(plan);
        void *ref = NULL;
        HIP_CHECK(hipMalloc(&ref, cb));
        TN_CHECK(tn_schoolbook_dev(plan, a, b, ref, rows, NULL));
        TN_CHECK(tn_plan_synchronize(plan));
        unsigned char *x = (unsigned char *)malloc(cb), *y = (unsigned char *)malloc(cb);
        if (!x || !y) return 1;
        HIP_CHECK(hipMemcpy(x, c, cb, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(y, ref, cb, hipMemcpyDeviceToHost));
        if (memcmp(x, y, cb) != 0) { fprintf(stderr, "check failed: NTT product differs from the direct product\n"); return 1; }
        printf("check=ok rows=%zu\n", rows);
        free(x); free(y); (void)hipFree(ref);
    }
    const double units = (double)reps * (double)batch;
    printf("forward_ntt_total_ns=%.0f\nforward_ntt_avg_ns=%.2f\nforward_ntt_checksum=%llu\n", fwd_ns, fwd_ns / units, (unsigned long long)h[0]);
    printf("total_ns=%.0f\navg_ns=%.2f\nchecksum=%llu\n", mul_ns, mul_ns / units, (unsigned long long)h[1]);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(f); (void)hipFree(sums);
    tn_plan_destroy(plan);
    return 0;
}
