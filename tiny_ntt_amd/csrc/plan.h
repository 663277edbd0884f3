// plan.h — internal plan object behind the opaque tn_plan handle, and the
// kernel launch entry points implemented in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <atomic>
#include <mutex>
#include "fused_core.h"

struct tn_plan {
  tn::u32 n = 0, logn = 0;
  tn::u64 q = 0, psi = 0, omega = 0;
  int device = 0;
  int num_cus = 256;
  tn::u32 flags = 0;
  int elem_bytes = 8;
  bool has_fused = false, lazy = false, cg_lazy = false, cg_sched = false;
  bool canonical_inputs = false;   // TN_PLAN_CANONICAL_INPUTS was given and the fused product kernel has a schedule for it (else the flag is ignored)
  bool omega_only = false;   // created by tn_plan_create_omega: no psi, only the constant-geometry transforms
  bool general = false;      // created by tn_plan_create_general: psi / q not validated, tables computed literally (cg_ntt.py:78-92 for ANY psi)
  int k = 0;            // bitlen(q)
  tn::Arith<tn::u64> ar64 = {};    // kernel-argument constants (h_make_arith); the one matching elem_bytes is used
  tn::Arith<tn::u32> ar32 = {};
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // host-buffer entry points: chunks flow H2D (copy_in) -> kernel (stream) -> D2H (copy_out) through HOST_SLOTS
  // device staging slots; created on first use
  static constexpr int HOST_SLOTS = 3;
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  hipEvent_t ev_in[HOST_SLOTS] = {}, ev_k[HOST_SLOTS] = {}, ev_out[HOST_SLOTS] = {};
  size_t host_chunk_rows = 0;      // rows per chunk; 0 = automatic (HOST_CHUNK_BYTES per operand)
  // device tables (Tw32[] or Tw64[] according to elem_bytes): split-constant records when the plan is lazy with 64-bit
  // lanes (h_make_fused_tw; the constant-geometry kernels then run their SPLIT instantiation), Shoup records
  // {w, floor(w 2^W / q)} otherwise.
  void* d_psi_brv = nullptr;       // [n]   psi^brv(i): merged forward twiddles (fused kernel)
  void* d_psi_inv_brv = nullptr;   // [n]   psi^-brv(i)
  void* d_omega_pow = nullptr;     // [n/2] omega^j   (cg_ntt.py:51,54 — pow(omega_s, i//k) = omega^(k*(i//k)))
  void* d_omega_inv_pow = nullptr; // [n/2] omega^-j
  void* d_psi_pow = nullptr;       // [n]   psi^i     (twist, cg_ntt.py:82-83)
  void* d_psi_inv_pow = nullptr;   // [n]   psi^-i
  void* d_cyc_brv = nullptr;       // [n]   merged twiddles of the cyclic transform (HostTables::cyc_brv): fused cg_ntt
  void* d_cyc_inv_brv = nullptr;   // [n]   their inverses: fused cg_intt
  void* d_psi_inv_ninv = nullptr;  // [n]   psi^-i * n^-1  (untwist :92 fused with the n^-1 of :74-75)
  // Host-side mutable state (staging scratch, the host pipeline's streams / events, ev0 / ev1 of the timing helper): the
  // *_host entry points and tn_time_poly_mult_dev take this lock, so two host threads may share one plan; *_dev entry
  // points touch none of it.
  mutable std::mutex host_mu;
  // dynamic row scheduling of the persistent fused kernel: SCHED_SLOTS pairs {next row, finished workgroups}, zeroed at
  // plan creation and re-armed by the kernel itself; consecutive launches take consecutive slots
  // a slot is handed out again only after the launch that used it last has retired (an event per slot, queried on reuse:
  // sched_acquire); while it is still busy the new launch falls back to the fixed stride, which is always correct
  static constexpr unsigned SCHED_SLOTS = 1024;     // launches that can be in flight with dynamic scheduling before the fallback applies
  tn::u32* d_sched = nullptr;
  mutable std::mutex sched_mu;
  mutable unsigned sched_seq = 0;
  mutable hipEvent_t sched_ev[SCHED_SLOTS] = {};
  mutable bool sched_used[SCHED_SLOTS] = {};
  void* d_scratch = nullptr;       // host-entry staging (grown on demand)
  size_t scratch_bytes = 0;
};

namespace tn {

enum FusedNttMode { FNTT_TWIST_FWD = 0, FNTT_CYCLIC_FWD = 1, FNTT_CYCLIC_INV = 2 };
enum CgMode { CG_NTT_FWD = 0, CG_NTT_INV = 1, CG_POLYMUL = 2, CG_TWIST_FWD = 3, CG_CYCLIC_POLYMUL = 4 };

template <typename E> struct PlanView {
  typedef typename TwOf<E>::type Tw;
  u32 n, logn;
  Arith<E> ar;
  const Tw* psi_brv;
  const Tw* psi_inv_brv;
  const Tw* omega_pow;
  const Tw* omega_inv_pow;
  const Tw* psi_pow;
  const Tw* psi_inv_ninv;
  const Tw* psi_inv_pow;
  const Tw* cyc_brv;
  const Tw* cyc_inv_brv;
};

template <typename E> inline const Arith<E>& plan_arith(const tn_plan* p);
template <> inline const Arith<u64>& plan_arith<u64>(const tn_plan* p) { return p->ar64; }
template <> inline const Arith<u32>& plan_arith<u32>(const tn_plan* p) { return p->ar32; }

template <typename E> inline PlanView<E> make_view(const tn_plan* p) {
  typedef typename TwOf<E>::type Tw;
  PlanView<E> v;
  v.n = p->n; v.logn = p->logn;
  v.ar = plan_arith<E>(p);
  v.psi_brv = (const Tw*)p->d_psi_brv; v.psi_inv_brv = (const Tw*)p->d_psi_inv_brv;
  v.omega_pow = (const Tw*)p->d_omega_pow; v.omega_inv_pow = (const Tw*)p->d_omega_inv_pow;
  v.psi_pow = (const Tw*)p->d_psi_pow; v.psi_inv_ninv = (const Tw*)p->d_psi_inv_ninv;
  v.psi_inv_pow = (const Tw*)p->d_psi_inv_pow;
  v.cyc_brv = (const Tw*)p->d_cyc_brv; v.cyc_inv_brv = (const Tw*)p->d_cyc_inv_brv;
  return v;
}

// Dynamic-row-scheduler slot for one launch on stream s, or nullptr (fixed stride, always correct) when
//   * the stream is being captured into a graph: a captured launch would bake the slot pointer into its kernel node while
//     the ring keeps advancing, so a later replay could share a counter pair with a live launch (rows skipped); and
//     hipEventQuery is not allowed while capturing;
//   * the ring's next slot is still in use by an earlier launch (possible across streams).
// A slot counts as used only once its event HAS BEEN RECORDED behind the launch (sched_release, under the same lock as the
// hand-out): sched_acquire returns with the lock held and sched_release drops it, so no other thread can see a slot that
// is handed out but whose event does not yet cover the launch.
struct SchedSlot { u32* ptr = nullptr; int index = -1; bool locked = false; };
inline SchedSlot sched_acquire(const tn_plan* p, hipStream_t stream) {
  SchedSlot r;
  if (!p->d_sched) return r;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return r;
  p->sched_mu.lock();
  const unsigned i = p->sched_seq % tn_plan::SCHED_SLOTS;
  bool ok = !(p->sched_used[i] && hipEventQuery(p->sched_ev[i]) != hipSuccess);       // still running (or errored): do not share it
  if (ok && !p->sched_ev[i]) ok = hipEventCreateWithFlags(&p->sched_ev[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { p->sched_mu.unlock(); return r; }
  r.ptr = p->d_sched + 2 * i; r.index = (int)i; r.locked = true;
  return r;
}
// launched: the kernel that uses the slot was enqueued (a failed launch leaves the slot free and the ring where it was)
inline void sched_release(const tn_plan* p, const SchedSlot& s, hipStream_t stream, bool launched = true) {
  if (!s.locked) return;
  if (launched) {
    if (hipEventRecord(p->sched_ev[s.index], stream) == hipSuccess) p->sched_used[s.index] = true;
    else (void)hipStreamSynchronize(stream);       // nothing can tell later when the kernel has finished: wait for it now, the slot stays free
    ++p->sched_seq;
  }
  p->sched_mu.unlock();
}

// kernels.hip
bool fused_supported(u32 logn, int elem_bytes);
const char* fused_kernel_name(const tn_plan* p);
const char* cg_kernel_name(const tn_plan* p, int group, int layout);
hipError_t launch_polymul_fused(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s, bool cyclic = false);
hipError_t launch_ntt_fused(const tn_plan* p, int mode, const void* in, void* out, size_t batch, hipStream_t s);
hipError_t launch_cg(const tn_plan* p, int mode, int group, int layout, const void* a, const void* b, void* out,
                     void* trace, size_t batch, hipStream_t s);
hipError_t launch_pointwise(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s);
hipError_t launch_schoolbook(const tn_plan* p, const void* a, const void* b, void* c, size_t batch, hipStream_t s);
hipError_t launch_fill_lcg(const tn_plan* p, void* dst, size_t batch, u64 seed0, u64 stride, hipStream_t s);
hipError_t launch_checksum(const tn_plan* p, const void* src, u64* out, size_t batch, hipStream_t s);

}  // namespace tn
