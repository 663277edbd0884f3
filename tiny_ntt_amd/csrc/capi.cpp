// capi.cpp — the extern "C" surface declared in include/tinyntt.h: plan
// creation (validation + exact table generation on the host), the launch entry
// points, host-buffer conveniences, timing helper.  No CPU compute fallback
// exists: without a HIP device every entry point fails with TN_ENODEVICE.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <new>
#include "../../include/tinyntt.h"
#include "plan.h"
#include "plan_tables.h"

using namespace tn;

static thread_local std::string g_err;

static tn_status fail(tn_status s, const std::string& msg) { g_err = msg; return s; }
static tn_status fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return TN_EHIP;
}
// Every entry point runs on the plan's device and leaves the caller's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) { err = hipSetDevice(dev); changed = (err == hipSuccess); }
  }
  ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};
#define TN_ON_DEVICE(p) DeviceGuard tn_guard_((p)->device); if (tn_guard_.err != hipSuccess) return fail_hip(tn_guard_.err, "hipSetDevice")

#define TN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip(e_, #call); } while (0)

extern "C" const char* tn_last_error(void) { return g_err.c_str(); }
extern "C" int tn_version(void) { return TN_VERSION; }
// tn_build_id(): build_id.cpp (its own object, recompiled whenever ANY source of the library changes)
extern "C" const char* tn_status_string(tn_status s) {
  switch (s) {
    case TN_OK: return "ok";
    case TN_EBADLEN: return "bad length";
    case TN_EBADPARAM: return "bad parameter";
    case TN_ENODEVICE: return "no HIP device";
    case TN_EHIP: return "HIP runtime error";
    case TN_ENOMEM: return "out of memory";
    case TN_EINVAL: return "invalid argument";
    case TN_EUNSUPPORTED: return "unsupported";
  }
  return "unknown";
}

// Upload a table of constants as Tw32[] or Tw64[]: the record format h_make_fused_tw picks for this plan (split constants
// when lazy with 64-bit lanes, Shoup records = value + Barrett quotient factor otherwise); fused = false forces Shoup.
static hipError_t upload_tw(const std::vector<u64>& vals, const HostTables& t, bool fused, void** dptr) {
  hipError_t e;
  if (t.elem_bytes == 8) {
    const std::vector<Tw64> r = fused ? h_fused_table<u64>(vals, t) : h_tw_table<u64>(vals, t.q);
    if ((e = hipMalloc(dptr, r.size() * sizeof(Tw64))) != hipSuccess) return e;
    return hipMemcpy(*dptr, r.data(), r.size() * sizeof(Tw64), hipMemcpyHostToDevice);
  }
  const std::vector<Tw32> r = fused ? h_fused_table<u32>(vals, t) : h_tw_table<u32>(vals, t.q);
  if ((e = hipMalloc(dptr, r.size() * sizeof(Tw32))) != hipSuccess) return e;
  return hipMemcpy(*dptr, r.data(), r.size() * sizeof(Tw32), hipMemcpyHostToDevice);
}

// The three plan constructors share everything but the parameter checks and the tables they build.
enum PlanKind { PLAN_PSI = 0, PLAN_OMEGA = 1, PLAN_GENERAL = 2 };
static tn_status plan_new(tn_plan** out, PlanKind kind, uint32_t n, uint64_t q, uint64_t root, int device, uint32_t flags, const char* fn) {
  if (!out) return fail(TN_EINVAL, std::string(fn) + ": out is NULL");
  *out = nullptr;
  u32 logn = 0;
  while (((u32)1 << logn) < n) ++logn;
  if (n < 4 || ((u32)1 << logn) != n) {
    char buf[96]; snprintf(buf, sizeof buf, "Expected a power-of-two length >= 4, got %u", n);
    return fail(TN_EBADLEN, buf);
  }
  if (kind == PLAN_PSI) {
    if (q < 3 || (q & 1) == 0 || q >= ((u64)1 << 62)) return fail(TN_EBADPARAM, "q must be an odd prime below 2^62");
    if (!h_is_prime(q)) return fail(TN_EBADPARAM, "q must be prime (modinv uses Fermat, cg_ntt.py:9-10)");
  } else if (q < 2 || q >= ((u64)1 << 62)) {
    return fail(TN_EBADPARAM, "the modulus must lie in [2, 2^62)");
  }
  const int elem_bytes = q < ((u64)1 << 31) ? 4 : 8;
  if (n > 8192u) {
    char buf[96]; snprintf(buf, sizeof buf, "n = %u exceeds the supported maximum for %d-byte coefficients", n, elem_bytes);
    return fail(TN_EBADLEN, buf);
  }
  root %= q;
  if (kind == PLAN_PSI && h_powmod(root, n, q) != q - 1)
    return fail(TN_EBADPARAM, "psi must satisfy psi^n == -1 mod q (primitive 2n-th root; benchmark_ntt_60bit.cpp:58-59)");

  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || ndev <= 0)
    return fail(TN_ENODEVICE, "no HIP device visible; libtinyntt has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(TN_EINVAL, "device index out of range");
  DeviceGuard guard(device);
  if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");

  tn_plan* p = new (std::nothrow) tn_plan();
  if (!p) return fail(TN_ENOMEM, "plan allocation failed");
  const HostTables t = kind == PLAN_PSI ? h_build_tables(n, q, root, !(flags & TN_PLAN_FORCE_CANONICAL))
                     : kind == PLAN_OMEGA ? h_build_omega_tables(n, q, root) : h_build_general_tables(n, q, root);
  p->n = n; p->logn = logn; p->q = q; p->psi = kind == PLAN_OMEGA ? 0 : t.psi; p->omega = t.omega;
  p->device = device; p->flags = flags; p->elem_bytes = elem_bytes;
  p->k = t.k; p->lazy = t.lazy; p->cg_lazy = t.cg_lazy; p->cg_sched = t.cg_sched;
  p->canonical_inputs = kind == PLAN_PSI && (flags & TN_PLAN_CANONICAL_INPUTS) && t.cin_ok;
  p->omega_only = kind == PLAN_OMEGA; p->general = kind != PLAN_PSI;       // (no reversal trick without omega^(n/2) == -1)
  if (elem_bytes == 8) p->ar64 = h_make_arith<u64>(t); else p->ar32 = h_make_arith<u32>(t);
  p->has_fused = kind == PLAN_PSI && fused_supported(logn, elem_bytes);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) p->num_cus = prop.multiProcessorCount;
  }
  struct Up { const std::vector<u64>* v; void** d; };
  const Up ups[] = {{&t.psi_brv, &p->d_psi_brv}, {&t.psi_inv_brv, &p->d_psi_inv_brv}, {&t.omega_pow, &p->d_omega_pow},
                    {&t.omega_inv_pow, &p->d_omega_inv_pow}, {&t.psi_pow, &p->d_psi_pow}, {&t.psi_inv_ninv, &p->d_psi_inv_ninv},
                    {&t.psi_inv_pow, &p->d_psi_inv_pow}, {&t.cyc_brv, &p->d_cyc_brv}, {&t.cyc_inv_brv, &p->d_cyc_inv_brv}};
  hipError_t e = hipSuccess;
  for (const Up& u : ups)
    if (e == hipSuccess && !u.v->empty()) e = upload_tw(*u.v, t, true, u.d);      // (record format: h_make_fused_tw — Shoup unless lazy 64-bit)
  if (kind == PLAN_PSI) {
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_sched, 2 * tn_plan::SCHED_SLOTS * sizeof(u32));
    if (e == hipSuccess) e = hipMemset(p->d_sched, 0, 2 * tn_plan::SCHED_SLOTS * sizeof(u32));
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&p->ev0);
  if (e == hipSuccess) e = hipEventCreate(&p->ev1);
  if (e != hipSuccess) { tn_plan_destroy(p); return fail_hip(e, "plan table upload"); }
  *out = p;
  return TN_OK;
}

extern "C" tn_status tn_plan_create(tn_plan** out, uint32_t n, uint64_t q, uint64_t psi, int device, uint32_t flags) {
  return plan_new(out, PLAN_PSI, n, q, psi, device, flags, "tn_plan_create");
}
extern "C" tn_status tn_plan_create_omega(tn_plan** out, uint32_t n, uint64_t q, uint64_t omega, int device, uint32_t flags) {
  return plan_new(out, PLAN_OMEGA, n, q, omega, device, flags, "tn_plan_create_omega");
}
extern "C" tn_status tn_plan_create_general(tn_plan** out, uint32_t n, uint64_t q, uint64_t psi, int device, uint32_t flags) {
  return plan_new(out, PLAN_GENERAL, n, q, psi, device, flags, "tn_plan_create_general");
}

extern "C" tn_status tn_plan_destroy(tn_plan* p) {
  if (!p) return TN_OK;
  DeviceGuard guard(p->device);
  void* tabs[] = {p->d_psi_brv, p->d_psi_inv_brv, p->d_omega_pow, p->d_omega_inv_pow, p->d_psi_pow, p->d_psi_inv_ninv, p->d_psi_inv_pow, p->d_cyc_brv, p->d_cyc_inv_brv, p->d_scratch, p->d_sched};
  for (void* t : tabs) if (t) (void)hipFree(t);
  for (unsigned i = 0; i < tn_plan::SCHED_SLOTS; ++i) if (p->sched_ev[i]) (void)hipEventDestroy(p->sched_ev[i]);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  for (int i = 0; i < tn_plan::HOST_SLOTS; ++i) {
    if (p->ev_in[i]) (void)hipEventDestroy(p->ev_in[i]);
    if (p->ev_k[i]) (void)hipEventDestroy(p->ev_k[i]);
    if (p->ev_out[i]) (void)hipEventDestroy(p->ev_out[i]);
  }
  if (p->copy_in) (void)hipStreamDestroy(p->copy_in);
  if (p->copy_out) (void)hipStreamDestroy(p->copy_out);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return TN_OK;
}

extern "C" uint32_t tn_plan_n(const tn_plan* p) { return p ? p->n : 0; }
extern "C" uint64_t tn_plan_q(const tn_plan* p) { return p ? p->q : 0; }
extern "C" uint64_t tn_plan_psi(const tn_plan* p) { return p ? p->psi : 0; }
extern "C" uint64_t tn_plan_omega(const tn_plan* p) { return p ? p->omega : 0; }
extern "C" uint32_t tn_plan_elem_bytes(const tn_plan* p) { return p ? (uint32_t)p->elem_bytes : 0; }
extern "C" int tn_plan_device(const tn_plan* p) { return p ? p->device : -1; }
extern "C" int tn_plan_has_fused(const tn_plan* p) { return p && p->has_fused; }
extern "C" int tn_plan_is_lazy(const tn_plan* p) { return p && p->lazy; }
extern "C" int tn_plan_is_general(const tn_plan* p) { return p && p->general && !p->omega_only; }

static hipStream_t pick_stream(tn_plan* p, void* stream) { return stream ? (hipStream_t)stream : p->stream; }

struct CgSel { int group; int layout; };     // layout: kernels.hip CgLayout (0 linear, 1 padded, 2 swizzled)
static bool cg_sel(tn_variant v, CgSel* s) {
  switch (v) {
    case TN_VARIANT_CG: *s = {1, 0}; return true;
    case TN_VARIANT_CG8: *s = {8, 0}; return true;
    case TN_VARIANT_CG8_PADDED: *s = {8, 1}; return true;
    case TN_VARIANT_CG_SWIZZLED: *s = {1, 2}; return true;
    case TN_VARIANT_CG8_SWIZZLED: *s = {8, 2}; return true;
    case TN_VARIANT_CG2: *s = {2, 0}; return true;
    case TN_VARIANT_CG2_PADDED: *s = {2, 1}; return true;
    case TN_VARIANT_CG2_SWIZZLED: *s = {2, 2}; return true;
    case TN_VARIANT_CG4: *s = {4, 0}; return true;
    case TN_VARIANT_CG4_PADDED: *s = {4, 1}; return true;
    case TN_VARIANT_CG4_SWIZZLED: *s = {4, 2}; return true;
    default: return false;
  }
}

static tn_status check_ptrs(const tn_plan* p, const void* a, const void* b, const void* c, size_t batch, const char* fn) {
  if (!p) return fail(TN_EINVAL, std::string(fn) + ": plan is NULL");
  if (batch > 0x7fffffffull) return fail(TN_EINVAL, std::string(fn) + ": batch too large for one call (max 2^31 - 1 rows)");
  if (batch && (!a || !b || !c)) return fail(TN_EINVAL, std::string(fn) + ": NULL buffer");
  if (batch) {                        // byte ranges: the kernels prefetch the next row's input while a row's result is being stored
    const size_t bytes = batch * (size_t)p->n * (size_t)p->elem_bytes;
    const char *ca = (const char*)a, *cb = (const char*)b, *cc = (const char*)c;
    if ((cc < ca + bytes && ca < cc + bytes) || (cc < cb + bytes && cb < cc + bytes))
      return fail(TN_EINVAL, std::string(fn) + ": output must not alias or overlap an input");
  }
  return TN_OK;
}

extern "C" tn_status tn_poly_mult_dev(tn_plan* p, const void* a, const void* b, void* c, size_t batch, tn_variant variant,
                                      void* stream) {
  tn_status st = check_ptrs(p, a, b, c, batch, "tn_poly_mult_dev");
  if (st) return st;
  if (p->omega_only) return fail(TN_EUNSUPPORTED, "tn_poly_mult_dev: an omega-only plan has no psi (tn_plan_create_omega offers cg_ntt / cg_intt only)");
  TN_ON_DEVICE(p);
  hipStream_t s = pick_stream(p, stream);
  if (variant == TN_VARIANT_AUTO) variant = p->has_fused ? TN_VARIANT_FUSED : TN_VARIANT_CG;
  if (variant == TN_VARIANT_FUSED) {
    if (!p->has_fused) return fail(TN_EUNSUPPORTED, "fused kernel not built for this n; use TN_VARIANT_CG");
    TN_HIP(launch_polymul_fused(p, a, b, c, batch, s));
    return TN_OK;
  }
  CgSel sel;
  if (!cg_sel(variant, &sel)) return fail(TN_EINVAL, "unknown variant");
  TN_HIP(launch_cg(p, CG_POLYMUL, sel.group, sel.layout, a, b, c, nullptr, batch, s));
  return TN_OK;
}

extern "C" tn_status tn_cyclic_poly_mult_dev(tn_plan* p, const void* a, const void* b, void* c, size_t batch, tn_variant variant,
                                             void* stream) {
  tn_status st = check_ptrs(p, a, b, c, batch, "tn_cyclic_poly_mult_dev");
  if (st) return st;
  if (p->omega_only) return fail(TN_EUNSUPPORTED, "tn_cyclic_poly_mult_dev: not available on an omega-only plan");
  TN_ON_DEVICE(p);
  if (variant == TN_VARIANT_AUTO) variant = p->has_fused ? TN_VARIANT_FUSED : TN_VARIANT_CG;
  if (variant == TN_VARIANT_FUSED) {
    if (!p->has_fused) return fail(TN_EUNSUPPORTED, "tn_cyclic_poly_mult_dev: fused kernel not built for this n; use TN_VARIANT_CG");
    TN_HIP(launch_polymul_fused(p, a, b, c, batch, pick_stream(p, stream), /*cyclic=*/true));
    return TN_OK;
  }
  CgSel sel;
  if (!cg_sel(variant, &sel)) return fail(TN_EINVAL, "tn_cyclic_poly_mult_dev: unknown variant");
  TN_HIP(launch_cg(p, CG_CYCLIC_POLYMUL, sel.group, sel.layout, a, b, c, nullptr, batch, pick_stream(p, stream)));
  return TN_OK;
}

extern "C" tn_status tn_pointwise_mul_dev(tn_plan* p, const void* a, const void* b, void* c, size_t batch, void* stream) {
  if (!p) return fail(TN_EINVAL, "tn_pointwise_mul_dev: plan is NULL");
  if (batch > 0xffffffffull) return fail(TN_EINVAL, "tn_pointwise_mul_dev: batch too large");
  if (batch && (!a || !b || !c)) return fail(TN_EINVAL, "tn_pointwise_mul_dev: NULL buffer");
  TN_ON_DEVICE(p);
  TN_HIP(launch_pointwise(p, a, b, c, batch, pick_stream(p, stream)));
  return TN_OK;
}

extern "C" tn_status tn_schoolbook_dev(tn_plan* p, const void* a, const void* b, void* c, size_t batch, void* stream) {
  tn_status st = check_ptrs(p, a, b, c, batch, "tn_schoolbook_dev");
  if (st) return st;
  TN_ON_DEVICE(p);
  TN_HIP(launch_schoolbook(p, a, b, c, batch, pick_stream(p, stream)));
  return TN_OK;
}

extern "C" tn_status tn_plan_export_table(tn_plan* p, int which, void* host_out) {
  if (!p || !host_out) return fail(TN_EINVAL, "tn_plan_export_table: NULL argument");
  if (p->omega_only && which != 2 && which != 3) return fail(TN_EUNSUPPORTED, "tn_plan_export_table: an omega-only plan has only the omega tables (2, 3)");
  if (p->general && (which == 4 || which == 5)) return fail(TN_EUNSUPPORTED, "tn_plan_export_table: only plans from tn_plan_create have the merged (bit-reversed) tables");
  const void* tabs[] = {p->d_psi_pow, p->d_psi_inv_ninv, p->d_omega_pow, p->d_omega_inv_pow, p->d_psi_brv, p->d_psi_inv_brv, p->d_psi_inv_pow};
  if (which < 0 || which > 6) return fail(TN_EINVAL, "tn_plan_export_table: unknown table");
  const size_t count = (which == 2 || which == 3) ? p->n / 2 : p->n;
  TN_ON_DEVICE(p);
  // device records are {w, w'} pairs (or, in a lazy 64-bit plan, split constants); only the constants w are
  // exported, as uint64
  std::vector<unsigned char> raw(count * 2 * (size_t)p->elem_bytes);
  TN_HIP(hipMemcpy(raw.data(), tabs[which], raw.size(), hipMemcpyDeviceToHost));
  uint64_t* out = (uint64_t*)host_out;
  const bool split = p->lazy && p->elem_bytes == 8;
  for (size_t i = 0; i < count; ++i) {
    if (p->elem_bytes == 8) out[i] = split ? h_split_value(((const Tw64*)raw.data())[i], p->k) : ((const Tw64*)raw.data())[i].w;
    else out[i] = ((const Tw32*)raw.data())[i].w;
  }
  return TN_OK;
}

static tn_status ntt_dev(tn_plan* p, int mode, const void* in, void* out, size_t batch, tn_variant variant, void* stream,
                         void* trace, const char* fn) {
  tn_status st = check_ptrs(p, in, in, out, batch, fn);
  if (st) return st;
  if (p->omega_only && mode == CG_TWIST_FWD) return fail(TN_EUNSUPPORTED, std::string(fn) + ": an omega-only plan has no psi to twist with");
  TN_ON_DEVICE(p);
  if (variant == TN_VARIANT_AUTO) variant = (p->has_fused && !trace) ? TN_VARIANT_FUSED : TN_VARIANT_CG;
  if (variant == TN_VARIANT_FUSED) {
    // register-tiled kernel: same results, no per-stage trace (its internal stages are not the CG stages)
    if (!p->has_fused) return fail(TN_EUNSUPPORTED, std::string(fn) + ": fused kernel not built for this n; use TN_VARIANT_CG");
    if (trace) return fail(TN_EUNSUPPORTED, std::string(fn) + ": per-stage traces need a CG variant");
    const int fmode = mode == CG_NTT_FWD ? FNTT_CYCLIC_FWD : (mode == CG_NTT_INV ? FNTT_CYCLIC_INV : FNTT_TWIST_FWD);
    TN_HIP(launch_ntt_fused(p, fmode, in, out, batch, pick_stream(p, stream)));
    return TN_OK;
  }
  CgSel sel;
  if (!cg_sel(variant, &sel)) return fail(TN_EINVAL, std::string(fn) + ": unknown variant");
  TN_HIP(launch_cg(p, mode, sel.group, sel.layout, in, nullptr, out, trace, batch, pick_stream(p, stream)));
  return TN_OK;
}

extern "C" tn_status tn_ntt_forward_dev(tn_plan* p, const void* in, void* out, size_t batch, tn_variant v, void* stream) {
  return ntt_dev(p, CG_NTT_FWD, in, out, batch, v, stream, nullptr, "tn_ntt_forward_dev");
}
extern "C" tn_status tn_ntt_inverse_dev(tn_plan* p, const void* in, void* out, size_t batch, tn_variant v, void* stream) {
  return ntt_dev(p, CG_NTT_INV, in, out, batch, v, stream, nullptr, "tn_ntt_inverse_dev");
}
extern "C" tn_status tn_twisted_ntt_forward_dev(tn_plan* p, const void* in, void* out, size_t batch, tn_variant v, void* stream) {
  return ntt_dev(p, CG_TWIST_FWD, in, out, batch, v, stream, nullptr, "tn_twisted_ntt_forward_dev");
}

// ---- host-buffer conveniences ------------------------------------------------
static tn_status ensure_scratch(tn_plan* p, size_t bytes) {
  if (bytes <= p->scratch_bytes) return TN_OK;
  if (p->d_scratch) { (void)hipFree(p->d_scratch); p->d_scratch = nullptr; p->scratch_bytes = 0; }
  hipError_t e = hipMalloc(&p->d_scratch, bytes);
  if (e != hipSuccess) return fail(TN_ENOMEM, std::string("device scratch allocation failed: ") + hipGetErrorString(e));
  p->scratch_bytes = bytes;
  return TN_OK;
}

// Host-buffer pipeline.  The batch is cut into chunks of `rows` rows; chunk i uses staging slot i % HOST_SLOTS
// (n_in input buffers + 1 output buffer on the device) and flows through three streams:
//   copy_in : H2D of the chunk's inputs      (waits until the kernel that last read the slot has finished)
//   stream  : the kernel                     (waits for the H2D and for the D2H that last read the slot's output)
//   copy_out: D2H of the chunk's output      (waits for the kernel)
// so with pinned host memory PCIe traffic in both directions overlaps the kernels.  With pageable memory the
// runtime's copies block the calling thread; the issue order below (inputs of chunk i+1 before the output of
// chunk i) still overlaps the next H2D with the current kernel.  Device staging is bounded by
// HOST_SLOTS * (n_in + 1) * chunk bytes whatever the batch.
static const size_t HOST_CHUNK_BYTES = (size_t)32 << 20;

static tn_status host_pipe_init(tn_plan* p) {
  if (p->copy_in) return TN_OK;
  TN_HIP(hipStreamCreateWithFlags(&p->copy_in, hipStreamNonBlocking));
  TN_HIP(hipStreamCreateWithFlags(&p->copy_out, hipStreamNonBlocking));
  for (int i = 0; i < tn_plan::HOST_SLOTS; ++i) {
    TN_HIP(hipEventCreateWithFlags(&p->ev_in[i], hipEventDisableTiming));
    TN_HIP(hipEventCreateWithFlags(&p->ev_k[i], hipEventDisableTiming));
    TN_HIP(hipEventCreateWithFlags(&p->ev_out[i], hipEventDisableTiming));
  }
  return TN_OK;
}

// launch(in0, in1, out, rows): enqueue the kernel for one chunk on p->stream (device pointers)
template <typename Launch>
static tn_status host_pipeline_run(tn_plan* p, int n_in, const void* const* in, void* out, size_t batch, Launch&& launch);

// Never returns with copies or kernels still in flight on the caller's buffers: on an error the three streams
// are drained before the status is handed back.
template <typename Launch>
static tn_status host_pipeline(tn_plan* p, int n_in, const void* const* in, void* out, size_t batch, Launch&& launch) {
  const tn_status st = host_pipeline_run(p, n_in, in, out, batch, launch);
  if (st != TN_OK) {
    const std::string msg = g_err;                   // keep the first error's message
    if (p->copy_in) (void)hipStreamSynchronize(p->copy_in);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->copy_out) (void)hipStreamSynchronize(p->copy_out);
    g_err = msg;
  }
  return st;
}

template <typename Launch>
static tn_status host_pipeline_run(tn_plan* p, int n_in, const void* const* in, void* out, size_t batch, Launch&& launch) {
  tn_status st;
  if ((st = host_pipe_init(p))) return st;
  const size_t row_bytes = (size_t)p->n * (size_t)p->elem_bytes;
  size_t rows = p->host_chunk_rows ? p->host_chunk_rows : (HOST_CHUNK_BYTES / row_bytes ? HOST_CHUNK_BYTES / row_bytes : 1);
  if (rows > batch) rows = batch;
  const size_t nchunks = (batch + rows - 1) / rows;
  const int slots = nchunks < (size_t)tn_plan::HOST_SLOTS ? (int)nchunks : tn_plan::HOST_SLOTS;
  const size_t chunk_bytes = rows * row_bytes;
  if ((st = ensure_scratch(p, (size_t)slots * (size_t)(n_in + 1) * chunk_bytes))) return st;
  char* base = (char*)p->d_scratch;
  auto slot_buf = [&](int slot, int j) { return base + ((size_t)slot * (size_t)(n_in + 1) + (size_t)j) * chunk_bytes; };
  auto rows_of = [&](size_t i) { return i + 1 < nchunks ? rows : batch - i * rows; };
  auto issue_in = [&](size_t i) -> tn_status {
    const int slot = (int)(i % (size_t)slots);
    const size_t off = i * chunk_bytes, bytes = rows_of(i) * row_bytes;
    if (i >= (size_t)slots) TN_HIP(hipStreamWaitEvent(p->copy_in, p->ev_k[slot], 0));
    for (int j = 0; j < n_in; ++j)
      TN_HIP(hipMemcpyAsync(slot_buf(slot, j), (const char*)in[j] + off, bytes, hipMemcpyHostToDevice, p->copy_in));
    TN_HIP(hipEventRecord(p->ev_in[slot], p->copy_in));
    return TN_OK;
  };
  if ((st = issue_in(0))) return st;
  for (size_t i = 0; i < nchunks; ++i) {
    const int slot = (int)(i % (size_t)slots);
    TN_HIP(hipStreamWaitEvent(p->stream, p->ev_in[slot], 0));
    if (i >= (size_t)slots) TN_HIP(hipStreamWaitEvent(p->stream, p->ev_out[slot], 0));
    if ((st = launch(slot_buf(slot, 0), n_in > 1 ? slot_buf(slot, 1) : nullptr, slot_buf(slot, n_in), rows_of(i)))) return st;
    TN_HIP(hipEventRecord(p->ev_k[slot], p->stream));
    if (i + 1 < nchunks && (st = issue_in(i + 1))) return st;
    TN_HIP(hipStreamWaitEvent(p->copy_out, p->ev_k[slot], 0));
    TN_HIP(hipMemcpyAsync((char*)out + i * chunk_bytes, slot_buf(slot, n_in), rows_of(i) * row_bytes, hipMemcpyDeviceToHost, p->copy_out));
    TN_HIP(hipEventRecord(p->ev_out[slot], p->copy_out));
  }
  TN_HIP(hipStreamSynchronize(p->copy_out));
  TN_HIP(hipStreamSynchronize(p->stream));
  TN_HIP(hipStreamSynchronize(p->copy_in));
  return TN_OK;
}

extern "C" tn_status tn_plan_set_host_chunk_rows(tn_plan* p, size_t rows) {
  if (!p) return fail(TN_EINVAL, "tn_plan_set_host_chunk_rows: plan is NULL");
  std::lock_guard<std::mutex> host_lock(p->host_mu);
  p->host_chunk_rows = rows;
  return TN_OK;
}

extern "C" tn_status tn_poly_mult_host(tn_plan* p, const void* a, const void* b, void* c, size_t batch, tn_variant variant) {
  tn_status st = check_ptrs(p, a, b, c, batch, "tn_poly_mult_host");
  if (st || batch == 0) return st;
  std::lock_guard<std::mutex> host_lock(p->host_mu);
  TN_ON_DEVICE(p);
  const void* in[2] = {a, b};
  return host_pipeline(p, 2, in, c, batch, [&](const void* da, const void* db, void* dc, size_t rows) {
    return tn_poly_mult_dev(p, da, db, dc, rows, variant, nullptr);
  });
}

static tn_status ntt_host(tn_plan* p, int mode, const void* in, void* out, void* trace, size_t batch, tn_variant v, const char* fn) {
  tn_status st = check_ptrs(p, in, in, out, batch, fn);
  if (st || batch == 0) return st;
  std::lock_guard<std::mutex> host_lock(p->host_mu);
  TN_ON_DEVICE(p);
  if (!trace) {
    const void* ins[1] = {in};
    return host_pipeline(p, 1, ins, out, batch, [&](const void* di, const void*, void* dout, size_t rows) {
      return ntt_dev(p, mode, di, dout, rows, v, nullptr, nullptr, fn);
    });
  }
  const size_t bytes = batch * p->n * (size_t)p->elem_bytes;          // traced transform: one row, one pass
  const size_t tbytes = bytes * p->logn;
  if ((st = ensure_scratch(p, 2 * bytes + tbytes))) return st;
  char* d = (char*)p->d_scratch;
  TN_HIP(hipMemcpyAsync(d, in, bytes, hipMemcpyHostToDevice, p->stream));
  if ((st = ntt_dev(p, mode, d, d + bytes, batch, v, nullptr, d + 2 * bytes, fn))) return st;
  TN_HIP(hipMemcpyAsync(out, d + bytes, bytes, hipMemcpyDeviceToHost, p->stream));
  TN_HIP(hipMemcpyAsync(trace, d + 2 * bytes, tbytes, hipMemcpyDeviceToHost, p->stream));
  TN_HIP(hipStreamSynchronize(p->stream));
  return TN_OK;
}

extern "C" tn_status tn_ntt_forward_host(tn_plan* p, const void* in, void* out, size_t batch, tn_variant v) {
  return ntt_host(p, CG_NTT_FWD, in, out, nullptr, batch, v, "tn_ntt_forward_host");
}
extern "C" tn_status tn_ntt_inverse_host(tn_plan* p, const void* in, void* out, size_t batch, tn_variant v) {
  return ntt_host(p, CG_NTT_INV, in, out, nullptr, batch, v, "tn_ntt_inverse_host");
}
extern "C" tn_status tn_twisted_ntt_forward_host(tn_plan* p, const void* in, void* out, size_t batch, tn_variant v) {
  return ntt_host(p, CG_TWIST_FWD, in, out, nullptr, batch, v, "tn_twisted_ntt_forward_host");
}
extern "C" tn_status tn_schoolbook_host(tn_plan* p, const void* a, const void* b, void* c, size_t batch) {
  tn_status st = check_ptrs(p, a, b, c, batch, "tn_schoolbook_host");
  if (st || batch == 0) return st;
  std::lock_guard<std::mutex> host_lock(p->host_mu);
  TN_ON_DEVICE(p);
  const void* in[2] = {a, b};
  return host_pipeline(p, 2, in, c, batch, [&](const void* da, const void* db, void* dc, size_t rows) {
    return tn_schoolbook_dev(p, da, db, dc, rows, nullptr);
  });
}
extern "C" tn_status tn_ntt_forward_trace_host(tn_plan* p, const void* in, void* out, void* trace, tn_variant v) {
  if (!trace) return fail(TN_EINVAL, "tn_ntt_forward_trace_host: trace is NULL");
  return ntt_host(p, CG_NTT_FWD, in, out, trace, 1, v, "tn_ntt_forward_trace_host");
}

extern "C" tn_status tn_fill_lcg_dev(tn_plan* p, void* dst, size_t batch, uint64_t seed0, uint64_t seed_stride, void* stream) {
  if (!p || (batch && !dst)) return fail(TN_EINVAL, "tn_fill_lcg_dev: NULL argument");
  if (batch > 0xffffffffull) return fail(TN_EINVAL, "tn_fill_lcg_dev: batch too large");
  TN_ON_DEVICE(p);
  TN_HIP(launch_fill_lcg(p, dst, batch, seed0, seed_stride, pick_stream(p, stream)));
  return TN_OK;
}

extern "C" tn_status tn_checksum_rows_dev(tn_plan* p, const void* src, uint64_t* out, size_t batch, void* stream) {
  if (!p || (batch && (!src || !out))) return fail(TN_EINVAL, "tn_checksum_rows_dev: NULL argument");
  if (batch > 0xffffffffull) return fail(TN_EINVAL, "tn_checksum_rows_dev: batch too large");
  TN_ON_DEVICE(p);
  TN_HIP(launch_checksum(p, src, out, batch, pick_stream(p, stream)));
  return TN_OK;
}

extern "C" tn_status tn_plan_synchronize(tn_plan* p) {
  if (!p) return fail(TN_EINVAL, "tn_plan_synchronize: plan is NULL");
  TN_ON_DEVICE(p);
  TN_HIP(hipStreamSynchronize(p->stream));
  return TN_OK;
}

extern "C" tn_status tn_time_poly_mult_dev(tn_plan* p, const void* a, const void* b, void* c, size_t batch, tn_variant variant,
                                           int iters, float* ms_per_launch) {
  if (!ms_per_launch || iters < 1) return fail(TN_EINVAL, "tn_time_poly_mult_dev: bad iters/ms pointer");
  tn_status st = check_ptrs(p, a, b, c, batch, "tn_time_poly_mult_dev");
  if (st) return st;
  std::lock_guard<std::mutex> host_lock(p->host_mu);
  TN_ON_DEVICE(p);
  TN_HIP(hipEventRecord(p->ev0, p->stream));
  for (int i = 0; i < iters; ++i)
    if ((st = tn_poly_mult_dev(p, a, b, c, batch, variant, nullptr))) return st;
  TN_HIP(hipEventRecord(p->ev1, p->stream));
  TN_HIP(hipEventSynchronize(p->ev1));
  float ms = 0.f;
  TN_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
  *ms_per_launch = ms / (float)iters;
  return TN_OK;
}

extern "C" const char* tn_kernel_name(const tn_plan* p, tn_variant variant) {
  if (!p) return "";
  if (variant == TN_VARIANT_AUTO) variant = p->has_fused ? TN_VARIANT_FUSED : TN_VARIANT_CG;
  if (variant == TN_VARIANT_FUSED) return fused_kernel_name(p);
  CgSel sel;
  if (cg_sel(variant, &sel)) return cg_kernel_name(p, sel.group, sel.layout);
  return "";
}
