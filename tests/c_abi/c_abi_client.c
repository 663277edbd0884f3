/* Plain-C client of include/tinyntt.h (C99, no C++/HIP/torch types): what a maintainer of the reference's
 * C++ benchmark would write.  Built with gcc by tests/test_c_abi.py.
 *   usage: c_abi_client            -> exit 0 if a product on the GPU matches the schoolbook result computed here,
 *                                     exit 3 if no HIP device is visible (library reports TN_ENODEVICE, no CPU fallback),
 *                                     exit 1 on any mismatch / unexpected status. */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "tinyntt.h"

#define N 256
static const uint64_t Q = 8380417, PSI = 1239911;   /* new_reference/test_cg_ntt.py:7 */

int main(void) {
    tn_plan *plan = NULL;
    /* parameter validation happens before any device use */
    if (tn_plan_create(&plan, 100, Q, PSI, 0, TN_PLAN_DEFAULT) != TN_EBADLEN) return 1;
    if (tn_plan_create(&plan, N, Q, PSI + 1, 0, TN_PLAN_DEFAULT) != TN_EBADPARAM) return 1;
    tn_status st = tn_plan_create(&plan, N, Q, PSI, 0, TN_PLAN_DEFAULT);
    if (st == TN_ENODEVICE) { printf("no device: %s\n", tn_last_error()); return 3; }
    if (st != TN_OK) { fprintf(stderr, "%s\n", tn_last_error()); return 1; }
    if (tn_plan_elem_bytes(plan) != 4 || tn_plan_n(plan) != N || tn_plan_q(plan) != Q) return 1;

    uint32_t a[N], b[N], c[N], ref[N];
    uint64_t x = 1;
    for (int i = 0; i < N; ++i) { x = 6364136223846793005ULL * x + 1442695040888963407ULL; a[i] = (uint32_t)((x >> 17) % Q); }
    for (int i = 0; i < N; ++i) { x = 6364136223846793005ULL * x + 1442695040888963407ULL; b[i] = (uint32_t)((x >> 17) % Q); }
    for (int k = 0; k < N; ++k) ref[k] = 0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {            /* negacyclic schoolbook: test_cg_ntt.py:11-21 */
            uint64_t t = (uint64_t)a[i] * b[j] % Q;
            int k = i + j;
            if (k >= N) { k -= N; t = (Q - t) % Q; }
            ref[k] = (uint32_t)((ref[k] + t) % Q);
        }
    if (tn_poly_mult_host(plan, a, b, c, 1, TN_VARIANT_AUTO) != TN_OK) { fprintf(stderr, "%s\n", tn_last_error()); return 1; }
    for (int k = 0; k < N; ++k) if (c[k] != ref[k]) { fprintf(stderr, "mismatch at %d\n", k); return 1; }
    if (tn_poly_mult_host(plan, a, b, c, 1, TN_VARIANT_CG8) != TN_OK) return 1;
    for (int k = 0; k < N; ++k) if (c[k] != ref[k]) return 1;
    if (tn_poly_mult_host(plan, a, b, a, 1, TN_VARIANT_AUTO) != TN_EINVAL) return 1;     /* aliasing is rejected */
    tn_plan_destroy(plan);

    /* the same product through the multi-device entry points: two entries (both on device 0 here), 5 rows split 3 + 2 */
    {
        enum { ROWS = 5 };
        static uint32_t ma[ROWS][N], mb[ROWS][N], mc[ROWS][N];
        tn_multi *m = NULL;
        const int devs[2] = {0, 0};
        size_t first = 0, rows = 0;
        if (tn_shard_rows(ROWS, 2, 1, &first, &rows) != TN_OK || first != 3 || rows != 2) return 1;
        if (tn_multi_create(&m, N, Q, PSI, devs, 2, TN_PLAN_DEFAULT) != TN_OK) { fprintf(stderr, "%s\n", tn_multi_last_error()); return 1; }
        if (tn_multi_size(m) != 2 || tn_multi_device(m, 1) != 0 || !tn_multi_plan(m, 1)) return 1;
        for (int r = 0; r < ROWS; ++r)
            for (int i = 0; i < N; ++i) { ma[r][i] = a[(i + r) % N]; mb[r][i] = b[i]; }
        if (tn_multi_poly_mult_host(m, ma, mb, mc, ROWS, TN_VARIANT_AUTO) != TN_OK) { fprintf(stderr, "%s\n", tn_multi_last_error()); return 1; }
        for (int k = 0; k < N; ++k) if (mc[0][k] != ref[k]) { fprintf(stderr, "multi mismatch at %d\n", k); return 1; }
        if (tn_poly_mult_host(tn_multi_plan(m, 0), ma[4], mb[4], c, 1, TN_VARIANT_CG) != TN_OK) return 1;   /* last row again, one plan */
        for (int k = 0; k < N; ++k) if (mc[4][k] != c[k]) return 1;
        tn_multi_destroy(m);
    }
    printf("c abi ok (version %d)\n", tn_version());
    return 0;
}
