/*
 * tinyntt.h — C ABI of libtinyntt.so: batched negacyclic polynomial
 * multiplication in Z_q[x]/(x^n+1) on AMD MI355X (gfx950), bit-exact against
 * orhosko/tiny-ntt's golden model.
 *
 * The reference has no FFI layer for this path: its boundary is the Python
 * signature `nwc_poly_mult(a, b, psi_2n) -> c` (new_reference/cg_ntt.py:78),
 * its 8-butterfly twin (new_reference/cg_ntt_8butterfly.py:107), the transforms
 * `cg_ntt` / `cg_intt` (cg_ntt.py:29,68) and the C++ benchmark entry
 * `negacyclic_mul_ntt(a, b, out)` (software_benchmark/benchmark_ntt_60bit.cpp:148,
 * benchmark_ntt.cpp:194).  Each entry point below names the reference interface
 * it replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - Plain C: opaque plan handle, raw pointers, sizes; no C++/torch types.
 *   - Coefficients: row-major [batch][n], little-endian, uint32_t when q < 2^31,
 *     uint64_t otherwise (tn_plan_elem_bytes tells which).  Natural coefficient
 *     order in and out.  Outputs are canonical residues in [0, q)
 *     (cg_ntt.py:58-59 — Python's % is non-negative).
 *   - Inputs need not be reduced: any word value is taken mod q, as the
 *     reference's `%` does (cg_ntt.py:82-83).
 *   - *_dev entry points take DEVICE pointers (hipMalloc / torch tensors) and a
 *     hipStream_t passed as void* (NULL = the plan's own stream; TN_STREAM_LEGACY =
 *     the device's legacy default stream, hipStreamLegacy); they enqueue and return.  *_host entry points take host pointers, copy in, run, copy
 *     out and synchronise.
 *   - a, b are read-only; c must not alias or overlap a or b (byte ranges are checked: TN_EINVAL).
 *   - Every function returns a tn_status; nothing throws or aborts across the
 *     ABI.  tn_last_error() gives the message for the calling thread.
 *   - A plan's tables are immutable after creation.  *_dev calls on one plan serialise on the stream they are
 *     given and may be issued from several host threads (also on different streams: the fused kernels' row
 *     scheduler hands a counter slot to one launch at a time and falls back to a fixed stride when its ring is
 *     busy).  *_host calls and tn_time_poly_mult_dev use the plan's staging buffers and events and take a
 *     per-plan lock: concurrent callers are served one after the other.  Distinct plans/devices are independent.
 *   - Stream capture: *_dev launches may be captured into a hipGraph (hipStreamBeginCapture on the stream handed in) and
 *     replayed, also beside live launches of the same plan on other streams.  A captured launch never takes a slot of the
 *     row scheduler (it runs the fixed row stride), so no replay can share scheduler state with a live launch.  Plan
 *     creation, the *_host entry points and tn_time_poly_mult_dev synchronise and must not be captured.
 */
#ifndef TINYNTT_H
#define TINYNTT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TN_VERSION 100 /* 0.1.0 */
#define TN_STREAM_LEGACY ((void *)1) /* == hipStreamLegacy: the stream handle-0 callers (e.g. torch's default stream) mean */

typedef enum tn_status {
  TN_OK = 0,
  TN_EBADLEN = 1,   /* n not a supported power of two / wrong length  (ValueError at cg_ntt.py:36-37,:69-70,:79-80) */
  TN_EBADPARAM = 2, /* q even / out of range, psi^n != -1 mod q        (static_asserts at benchmark_ntt_60bit.cpp:58-59) */
  TN_ENODEVICE = 3, /* no usable HIP device: the library never falls back to a CPU path */
  TN_EHIP = 4,      /* a HIP runtime call failed */
  TN_ENOMEM = 5,
  TN_EINVAL = 6,    /* NULL pointer, aliasing, unknown variant */
  TN_EUNSUPPORTED = 7
} tn_status;

/* Kernel schedule selector.  All variants return identical bits. */
typedef enum tn_variant {
  TN_VARIANT_AUTO = 0,  /* fastest available: FUSED when the plan supports it, else CG */
  TN_VARIANT_FUSED = 1, /* register/LDS-tiled merged-twiddle kernel (the throughput path) */
  TN_VARIANT_CG = 2,    /* constant-geometry stage sweep in LDS: the dataflow of cg_ntt.py:49-64 */
  TN_VARIANT_CG8 = 3,   /* same, butterflies issued 8 per lane-step: cg_ntt_8butterfly.py:61-89 */
  TN_VARIANT_CG8_PADDED = 4, /* CG8 with 16 bytes of padding per lane-step in the LDS image; for the rocprof sweep */
  /* the rest of the lane-grouping x LDS-layout sweep (BASELINE config 5): GROUP butterflies per lane-step in {1, 2, 4, 8},
   * layout in {linear, padded, swizzled}; SWIZZLED = XOR-swizzled image, bank-conflict free for every access of the sweep */
  TN_VARIANT_CG_SWIZZLED = 5,
  TN_VARIANT_CG8_SWIZZLED = 6,
  TN_VARIANT_CG2 = 7,
  TN_VARIANT_CG2_PADDED = 8,
  TN_VARIANT_CG2_SWIZZLED = 9,
  TN_VARIANT_CG4 = 10,
  TN_VARIANT_CG4_PADDED = 11,
  TN_VARIANT_CG4_SWIZZLED = 12
} tn_variant;

typedef struct tn_plan tn_plan;

/* Plan flags */
#define TN_PLAN_DEFAULT 0u
#define TN_PLAN_FORCE_CANONICAL 1u /* disable lazy reduction in the FUSED kernel (debug / generic-modulus path) */
/* The caller PROMISES that every input coefficient of tn_poly_mult_* is already in [0, q) — the precondition of the reference's
 * C++ path (mod_add / mod_sub / mod_mul, benchmark_ntt_60bit.cpp:66-77, on make_poly's output :79-87); the Python path reduces
 * with % (cg_ntt.py:82-83) and needs no promise.  The fused product kernel at n = 4096 / 64-bit lanes then skips the input
 * folds (bound schedule started from q, replayed exactly at plan creation); every other kernel ignores the flag.  With the flag
 * set, inputs >= q give undefined results. */
#define TN_PLAN_CANONICAL_INPUTS 2u

/*
 * Create a plan for (n, q, psi) on HIP device `device`.
 * Replaces: module constants N, Q (cg_ntt.py:5-6) + the psi_2n argument of
 * nwc_poly_mult (:78); BENCH_N/BENCH_Q/BENCH_PSI (software_benchmark/CMakeLists.txt:5-7)
 * and the constexpr tables PsiPowers/OmegaPowers/... (benchmark_ntt_60bit.cpp:43-64).
 * Validates n = 2^m (4 <= n <= 8192; fused kernels for 256 <= n <= 8192), q odd prime < 2^62, psi^n == -1 (mod q):
 * the preconditions of the throughput kernels.  Parameters it rejects (TN_EBADPARAM) still have a defined result in the
 * reference: tn_plan_create_general computes it.
 */
tn_status tn_plan_create(tn_plan **out, uint32_t n, uint64_t q, uint64_t psi, int device, uint32_t flags);
/*
 * Plan for the UNTWISTED transforms with an arbitrary omega_n: cg_ntt(a_prime, omega_n, modulus) and
 * cg_intt(A, omega_n, modulus) (cg_ntt.py:29-75) evaluate their butterflies for ANY omega_n — it need not be a primitive
 * n-th root, nor have a square root psi mod q; the inverse uses modinv(omega_n) = omega_n^(q-2) (:72) and n^-1 (:74).
 * Such a plan offers tn_ntt_forward_* / tn_ntt_inverse_* / tn_ntt_forward_trace_host with the constant-geometry variants
 * (plus tn_pointwise_mul_dev, tn_fill_lcg_dev, tn_checksum_rows_dev); everything that needs psi returns TN_EUNSUPPORTED.
 * q: ANY modulus in [2, 2^62), prime or not, odd or even — "modinv" is pow(v, q-2, q) exactly as cg_ntt.py:9-10 computes it.
 */
tn_status tn_plan_create_omega(tn_plan **out, uint32_t n, uint64_t q, uint64_t omega, int device, uint32_t flags);
/*
 * Plan that validates NOTHING beyond the sizes: nwc_poly_mult(a, b, psi_2n) (cg_ntt.py:78-92) never checks psi_2n or the
 * modulus, so it returns a defined list for every psi_2n (roots of unity or not, zero included) and every modulus; this plan
 * computes that list: tables psi^i, omega = psi^2, pow(omega, q-2, q), pow(n, q-2, q), pow(psi, q-2, q)^i built literally,
 * constant-geometry kernels with canonical arithmetic (TN_VARIANT_AUTO = TN_VARIANT_CG; TN_VARIANT_FUSED is unsupported).
 * Offers every entry point except the merged tables of tn_plan_export_table (4, 5).  q in [2, 2^62); 4 <= n <= 8192.
 * The Python mirror falls back to it when tn_plan_create rejects (q, psi).
 */
tn_status tn_plan_create_general(tn_plan **out, uint32_t n, uint64_t q, uint64_t psi, int device, uint32_t flags);
tn_status tn_plan_destroy(tn_plan *plan);

uint32_t tn_plan_n(const tn_plan *plan);
uint64_t tn_plan_q(const tn_plan *plan);
uint64_t tn_plan_psi(const tn_plan *plan);
uint64_t tn_plan_omega(const tn_plan *plan);      /* psi^2: the omega_n that nwc_poly_mult passes to cg_ntt (cg_ntt.py:85) */
uint32_t tn_plan_elem_bytes(const tn_plan *plan); /* 4 or 8 */
int tn_plan_device(const tn_plan *plan);
int tn_plan_has_fused(const tn_plan *plan);       /* 1 if TN_VARIANT_FUSED is available for this (n, q) */
int tn_plan_is_lazy(const tn_plan *plan);         /* 1 if the fused kernel runs with lazy reduction */
int tn_plan_is_general(const tn_plan *plan);      /* 1 for a plan made by tn_plan_create_general */

/*
 * c[r] = a[r] * b[r] in Z_q[x]/(x^n+1) for r < batch.
 * Replaces: nwc_poly_mult(a, b, psi_2n) (cg_ntt.py:78-92), nwc_poly_mult_8butterfly
 * (cg_ntt_8butterfly.py:107-121; use TN_VARIANT_CG8), negacyclic_mul_ntt(a, b, out)
 * (benchmark_ntt_60bit.cpp:148-159).
 */
tn_status tn_poly_mult_dev(tn_plan *plan, const void *a, const void *b, void *c, size_t batch,
                           tn_variant variant, void *stream);
tn_status tn_poly_mult_host(tn_plan *plan, const void *a, const void *b, void *c, size_t batch,
                            tn_variant variant);

/*
 * The *_host entry points cut the batch into chunks that flow H2D -> kernel -> D2H on three
 * streams through a fixed set of device staging slots (copies overlap the kernels when the host
 * buffers are pinned; device staging stays bounded whatever the batch).  rows = rows per chunk,
 * 0 = automatic (32 MiB per operand).  No counterpart in the reference (its callers hand over
 * Python lists / std::vector, cg_ntt.py:78, benchmark_ntt_60bit.cpp:148).
 */
tn_status tn_plan_set_host_chunk_rows(tn_plan *plan, size_t rows);

/*
 * Untwisted cyclic transforms with omega = psi^2, natural order in and out.
 * tn_ntt_forward_*  replaces cg_ntt(a_prime, omega_n, modulus)  (cg_ntt.py:29-65),
 *                   cg_ntt_8butterfly (cg_ntt_8butterfly.py:41-97) with TN_VARIANT_CG8.
 * tn_ntt_inverse_*  replaces cg_intt(A, omega_n, modulus)       (cg_ntt.py:68-75).
 * TN_VARIANT_AUTO / FUSED: register-tiled kernel (merged transform + one extra LDS transpose for natural order);
 * CG variants: the reference's stage sweep.  Identical results.
 */
tn_status tn_ntt_forward_dev(tn_plan *plan, const void *in, void *out, size_t batch, tn_variant variant, void *stream);
tn_status tn_ntt_inverse_dev(tn_plan *plan, const void *in, void *out, size_t batch, tn_variant variant, void *stream);
tn_status tn_ntt_forward_host(tn_plan *plan, const void *in, void *out, size_t batch, tn_variant variant);
tn_status tn_ntt_inverse_host(tn_plan *plan, const void *in, void *out, size_t batch, tn_variant variant);

/*
 * Forward transform of ONE polynomial that also returns every stage's output:
 * trace is [log2 n][n] elements, row s-1 = the list `A` after stage s — the
 * data cg_ntt(..., verbose=True) prints 16-at-a-time (cg_ntt.py:60-62).  CG variants only (AUTO = CG here):
 * the stages observed are the reference's constant-geometry stages.
 */
tn_status tn_ntt_forward_trace_host(tn_plan *plan, const void *in, void *out, void *trace, tn_variant variant);

/*
 * twist + forward transform: the reference's second timed unit,
 * forward_ntt_bench(a, out) (benchmark_ntt_60bit.cpp:161-165).
 */
tn_status tn_twisted_ntt_forward_dev(tn_plan *plan, const void *in, void *out, size_t batch, tn_variant variant, void *stream);
tn_status tn_twisted_ntt_forward_host(tn_plan *plan, const void *in, void *out, size_t batch, tn_variant variant);

/*
 * Untwisted (CYCLIC) product: cg_ntt(a), cg_ntt(b), pointwise, cg_intt with omega = psi^2 —
 * python_poly_mult (test/cocotb_tests/test_ntt_poly_mult.py:38-43), i.e. what the reference's
 * RTL top level / RoCC accelerator computes.  TN_VARIANT_FUSED runs the fused
 * product kernel on the twiddle tables of the x^n - 1 factorisation tree (no twist anywhere).
 */
tn_status tn_cyclic_poly_mult_dev(tn_plan *plan, const void *a, const void *b, void *c, size_t batch,
                                  tn_variant variant, void *stream);

/* c[i] = a[i] * b[i] mod q over batch*n coefficients: pointwise_mul (benchmark_ntt_60bit.cpp:142-146; cg_ntt.py:88). */
tn_status tn_pointwise_mul_dev(tn_plan *plan, const void *a, const void *b, void *c, size_t batch, void *stream);

/*
 * O(n^2) direct negacyclic product on device — negacyclic_mul_reference (benchmark_ntt_60bit.cpp:167-180),
 * the benchmark_simple family, negacyclic_convolution (new_reference/test_cg_ntt.py:11-21).  An
 * independent on-device checker for the NTT path; not a throughput kernel.
 */
tn_status tn_schoolbook_dev(tn_plan *plan, const void *a, const void *b, void *c, size_t batch, void *stream);
tn_status tn_schoolbook_host(tn_plan *plan, const void *a, const void *b, void *c, size_t batch);

/*
 * Copy one of the plan's constant tables to the host as uint64_t values (the constants only,
 * without their Barrett factors).  which: 0 psi^i [n] (= rtl/twiddle_forward*.hex), 1 psi^-i * n^-1 [n],
 * 2 omega^j [n/2], 3 omega^-j [n/2], 4 psi^brv(i) [n], 5 psi^-brv(i) [n], 6 psi^-i [n] (= rtl/twiddle_inverse*.hex).
 */
tn_status tn_plan_export_table(tn_plan *plan, int which, void *host_out);

/*
 * Synthetic inputs and digests, on device, in the reference benchmark's own
 * conventions so runs can be diffed against its printed checksums:
 * tn_fill_lcg_dev: row r gets make_poly(seed0 + r * seed_stride)
 *                  (benchmark_ntt_60bit.cpp:79-87; benchmark_ntt.cpp:82-90 when q < 2^32).
 * tn_checksum_rows_dev: out[r] = checksum(row r)  (benchmark_ntt_60bit.cpp:182-188;
 *                  benchmark_ntt.cpp:228-233 when q < 2^32).  out is a DEVICE uint64_t[batch].
 */
tn_status tn_fill_lcg_dev(tn_plan *plan, void *dst, size_t batch, uint64_t seed0, uint64_t seed_stride, void *stream);
tn_status tn_checksum_rows_dev(tn_plan *plan, const void *src, uint64_t *out, size_t batch, void *stream);

/* Blocks until everything enqueued on the plan's own stream has finished. */
tn_status tn_plan_synchronize(tn_plan *plan);

/*
 * Times `iters` back-to-back launches of tn_poly_mult_dev on the plan's own
 * stream with HIP events recorded on that stream (hipEventRecord); returns the
 * mean milliseconds per launch.  (bench.py uses this so the timed region is
 * measured on the stream the kernel runs on.)
 */
tn_status tn_time_poly_mult_dev(tn_plan *plan, const void *a, const void *b, void *c, size_t batch,
                                tn_variant variant, int iters, float *ms_per_launch);

/* Name of the kernel a variant resolves to for this plan (for matching rocprof rows). */
const char *tn_kernel_name(const tn_plan *plan, tn_variant variant);

/*
 * One call sharded over several devices.  Polynomial pairs are independent: the batch is cut into contiguous
 * row blocks (tn_shard_rows: block sizes differ by at most one row), one block per entry; every entry has its OWN plan and stream
 * on its device; there is no collective and no exchange.  devices == NULL: every visible device once.  A device may be listed
 * several times (each entry still gets its own plan and stream).  The reference has no counterpart (single-threaded, one host).
 *   tn_multi_poly_mult_host  host buffers [batch][n]: one host thread per entry runs the H2D -> kernel -> D2H pipeline of its block.
 *   tn_multi_poly_mult_dev   operands already resident per device: a[i], b[i], c[i] hold rows[i] rows on entry i's device; the
 *                            launches are enqueued on the entries' own streams; tn_multi_synchronize waits for all of them.
 * Errors: the first failing entry's status; tn_multi_last_error() names the entry and carries its message.
 */
typedef struct tn_multi tn_multi;
tn_status tn_shard_rows(size_t batch, int parts, int index, size_t *first_row, size_t *rows);
tn_status tn_multi_create(tn_multi **out, uint32_t n, uint64_t q, uint64_t psi, const int *devices, int ndevices, uint32_t flags);
tn_status tn_multi_destroy(tn_multi *m);
int tn_multi_size(const tn_multi *m);
tn_plan *tn_multi_plan(tn_multi *m, int index);   /* entry index's plan (owned by m) */
int tn_multi_device(const tn_multi *m, int index);
tn_status tn_multi_poly_mult_host(tn_multi *m, const void *a, const void *b, void *c, size_t batch, tn_variant variant);
tn_status tn_multi_poly_mult_dev(tn_multi *m, const void *const *a, const void *const *b, void *const *c, const size_t *rows,
                                 tn_variant variant);
tn_status tn_multi_synchronize(tn_multi *m);
const char *tn_multi_last_error(void);

const char *tn_last_error(void);
const char *tn_status_string(tn_status s);
int tn_version(void);
/* First 16 hex digits of the sha256 of the sources this library was built from (csrc/Makefile); lets a stored
 * measurement (profiles/traffic_latest.json) say which build it was taken on.  No counterpart in the reference. */
const char *tn_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* TINYNTT_H */
