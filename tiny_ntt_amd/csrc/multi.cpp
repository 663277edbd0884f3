// multi.cpp — one host call sharded over several devices: polynomial pairs are independent, so the batch is cut
// into contiguous row blocks, one per device entry, each with its OWN plan and stream; no collective, no exchange.  Plain C ABI
// (include/tinyntt.h: tn_multi_*), so a C caller does not have to shard by hand.  The same device may be listed more than once
// (each entry still gets its own plan and stream): that is how CI exercises the multi-device path on a one-GPU box.
#include <hip/hip_runtime.h>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "../../include/tinyntt.h"

struct tn_multi {
  std::vector<tn_plan*> plans;
  std::vector<int> devices;
  uint32_t n = 0;
  uint32_t elem_bytes = 0;
};

namespace {
thread_local std::string g_multi_err;
tn_status mfail(tn_status s, const std::string& m) { g_multi_err = m; return s; }
}  // namespace

extern "C" const char* tn_multi_last_error(void) { return g_multi_err.c_str(); }

extern "C" tn_status tn_shard_rows(size_t batch, int parts, int index, size_t* first_row, size_t* rows) {
  if (parts < 1 || index < 0 || index >= parts || !first_row || !rows) return mfail(TN_EINVAL, "tn_shard_rows: bad argument");
  const size_t base = batch / (size_t)parts, extra = batch % (size_t)parts, i = (size_t)index;
  *first_row = i * base + (i < extra ? i : extra);          // contiguous blocks whose sizes differ by at most one row
  *rows = base + (i < extra ? 1 : 0);
  return TN_OK;
}

extern "C" tn_status tn_multi_create(tn_multi** out, uint32_t n, uint64_t q, uint64_t psi, const int* devices, int ndevices, uint32_t flags) {
  if (!out) return mfail(TN_EINVAL, "tn_multi_create: out is NULL");
  *out = nullptr;
  std::vector<int> devs;
  if (devices) {
    if (ndevices < 1) return mfail(TN_EINVAL, "tn_multi_create: ndevices must be >= 1 when a device list is given");
    devs.assign(devices, devices + ndevices);
  } else {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return mfail(TN_ENODEVICE, "no HIP device visible; libtinyntt has no CPU fallback");
    for (int d = 0; d < count; ++d) devs.push_back(d);
  }
  tn_multi* m = new (std::nothrow) tn_multi();
  if (!m) return mfail(TN_ENOMEM, "tn_multi_create: allocation failed");
  for (int d : devs) {
    tn_plan* p = nullptr;
    const tn_status st = tn_plan_create(&p, n, q, psi, d, flags);
    if (st != TN_OK) {
      const std::string msg = tn_last_error();
      tn_multi_destroy(m);
      return mfail(st, msg);
    }
    m->plans.push_back(p);
    m->devices.push_back(d);
  }
  m->n = n;
  m->elem_bytes = tn_plan_elem_bytes(m->plans[0]);
  *out = m;
  return TN_OK;
}

extern "C" tn_status tn_multi_destroy(tn_multi* m) {
  if (!m) return TN_OK;
  for (tn_plan* p : m->plans) (void)tn_plan_destroy(p);
  delete m;
  return TN_OK;
}

extern "C" int tn_multi_size(const tn_multi* m) { return m ? (int)m->plans.size() : 0; }
extern "C" tn_plan* tn_multi_plan(tn_multi* m, int index) { return (m && index >= 0 && index < (int)m->plans.size()) ? m->plans[(size_t)index] : nullptr; }
extern "C" int tn_multi_device(const tn_multi* m, int index) { return (m && index >= 0 && index < (int)m->devices.size()) ? m->devices[(size_t)index] : -1; }

// One host thread per entry: tn_poly_mult_host is synchronous (H2D -> kernel -> D2H pipeline on the plan's own streams), and HIP's
// current device is per thread, so the entries run concurrently without sharing any state.
extern "C" tn_status tn_multi_poly_mult_host(tn_multi* m, const void* a, const void* b, void* c, size_t batch, tn_variant variant) {
  if (!m || m->plans.empty()) return mfail(TN_EINVAL, "tn_multi_poly_mult_host: no plans");
  if (batch && (!a || !b || !c)) return mfail(TN_EINVAL, "tn_multi_poly_mult_host: NULL buffer");
  const int parts = (int)m->plans.size();
  const size_t row_bytes = (size_t)m->n * m->elem_bytes;
  std::vector<tn_status> st((size_t)parts, TN_OK);
  std::vector<std::string> msg((size_t)parts);
  std::vector<std::thread> workers;
  for (int i = 0; i < parts; ++i) {
    size_t first = 0, rows = 0;
    (void)tn_shard_rows(batch, parts, i, &first, &rows);
    if (!rows) continue;
    workers.emplace_back([=, &st, &msg]() {
      const size_t off = first * row_bytes;
      st[(size_t)i] = tn_poly_mult_host(m->plans[(size_t)i], (const char*)a + off, (const char*)b + off, (char*)c + off, rows, variant);
      if (st[(size_t)i] != TN_OK) msg[(size_t)i] = tn_last_error();        // (the message is per thread: carry it out)
    });
  }
  for (std::thread& w : workers) w.join();
  for (int i = 0; i < parts; ++i)
    if (st[(size_t)i] != TN_OK) return mfail(st[(size_t)i], "entry " + std::to_string(i) + " (device " + std::to_string(m->devices[(size_t)i]) + "): " + msg[(size_t)i]);
  return TN_OK;
}

// Device-resident form: entry i's operands already live on ITS device (a[i], b[i], c[i]: rows[i] rows each); every entry's launch is
// enqueued on its own plan's stream and the call returns; tn_multi_synchronize waits for all of them.
extern "C" tn_status tn_multi_poly_mult_dev(tn_multi* m, const void* const* a, const void* const* b, void* const* c, const size_t* rows,
                                            tn_variant variant) {
  if (!m || m->plans.empty() || !a || !b || !c || !rows) return mfail(TN_EINVAL, "tn_multi_poly_mult_dev: NULL argument");
  for (size_t i = 0; i < m->plans.size(); ++i) {
    if (!rows[i]) continue;
    const tn_status st = tn_poly_mult_dev(m->plans[i], a[i], b[i], c[i], rows[i], variant, nullptr);
    if (st != TN_OK) return mfail(st, "entry " + std::to_string(i) + ": " + tn_last_error());
  }
  return TN_OK;
}

extern "C" tn_status tn_multi_synchronize(tn_multi* m) {
  if (!m) return mfail(TN_EINVAL, "tn_multi_synchronize: NULL");
  for (size_t i = 0; i < m->plans.size(); ++i) {
    const tn_status st = tn_plan_synchronize(m->plans[i]);
    if (st != TN_OK) return mfail(st, "entry " + std::to_string(i) + ": " + tn_last_error());
  }
  return TN_OK;
}
