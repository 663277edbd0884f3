// cg_part.hip — one slice of the constant-geometry kernel instantiations (cg_kernel_impl.h); compiled once per
// -DTN_CG_PART=k by the Makefile so the slices build in parallel:
//   0  32-bit lanes, Shoup records                     1  64-bit lanes, Shoup records (canonical plans, omega-only / any-psi plans)
//   2  64-bit lanes, split records, canonical (traces)  3  64-bit lanes, split records, lazy, any n
//   4-6  64-bit lanes, split records, lazy with the static fold schedule, n = 4096 compiled in: GROUP {1, 2} / 4 / 8 x the three LDS layouts (BASELINE config 5)
#include "cg_kernel_impl.h"

namespace tn {

#ifndef TN_CG_PART
#error "compile with -DTN_CG_PART=0..6"
#endif

#if TN_CG_PART <= 3
#if TN_CG_PART == 0
typedef u32 PartE; constexpr int PART_AM = CGA_SHOUP;
#define TN_CG_PART_FN launch_cg_part0
#elif TN_CG_PART == 1
typedef u64 PartE; constexpr int PART_AM = CGA_SHOUP;
#define TN_CG_PART_FN launch_cg_part1
#elif TN_CG_PART == 2
typedef u64 PartE; constexpr int PART_AM = CGA_SPLIT_CANON;
#define TN_CG_PART_FN launch_cg_part2
#else
typedef u64 PartE; constexpr int PART_AM = CGA_SPLIT_LAZY;
#define TN_CG_PART_FN launch_cg_part3
#endif
// any n, linear image; big: the n = 8192 instantiations of GROUP 1 and 2
hipError_t TN_CG_PART_FN(const tn_plan* p, int mode, int group, int, bool big, const void* a, const void* b, void* out, void* trace,
                         size_t batch, hipStream_t s) {
  if (big) {
    if (group == 1) return launch_cg_t<PartE, 1, CG_LINEAR, PART_AM, true, 0>(p, mode, a, b, out, trace, batch, s);
    if (group == 2) return launch_cg_t<PartE, 2, CG_LINEAR, PART_AM, true, 0>(p, mode, a, b, out, trace, batch, s);
    return hipErrorInvalidValue;
  }
  switch (group) {
    case 1: return launch_cg_t<PartE, 1, CG_LINEAR, PART_AM, false, 0>(p, mode, a, b, out, trace, batch, s);
    case 2: return launch_cg_t<PartE, 2, CG_LINEAR, PART_AM, false, 0>(p, mode, a, b, out, trace, batch, s);
    case 4: return launch_cg_t<PartE, 4, CG_LINEAR, PART_AM, false, 0>(p, mode, a, b, out, trace, batch, s);
    case 8: return launch_cg_t<PartE, 8, CG_LINEAR, PART_AM, false, 0>(p, mode, a, b, out, trace, batch, s);
    default: return hipErrorInvalidValue;
  }
}
#else
template <int GROUP>
static hipError_t by_layout(const tn_plan* p, int mode, int layout, const void* a, const void* b, void* out, void* trace, size_t batch,
                            hipStream_t s) {
  switch (layout) {
    case CG_LINEAR: return launch_cg_t<u64, GROUP, CG_LINEAR, CGA_SPLIT_SCHED, false, 12>(p, mode, a, b, out, trace, batch, s);
    case CG_PADDED: return launch_cg_t<u64, GROUP, CG_PADDED, CGA_SPLIT_SCHED, false, 12>(p, mode, a, b, out, trace, batch, s);
    case CG_SWIZZLED: return launch_cg_t<u64, GROUP, CG_SWIZZLED, CGA_SPLIT_SCHED, false, 12>(p, mode, a, b, out, trace, batch, s);
    default: return hipErrorInvalidValue;
  }
}
#if TN_CG_PART == 4
hipError_t launch_cg_part4(const tn_plan* p, int mode, int group, int layout, bool, const void* a, const void* b, void* out, void* trace,
                           size_t batch, hipStream_t s) {
  return group == 1 ? by_layout<1>(p, mode, layout, a, b, out, trace, batch, s) : by_layout<2>(p, mode, layout, a, b, out, trace, batch, s);
}
#elif TN_CG_PART == 5
hipError_t launch_cg_part5(const tn_plan* p, int mode, int, int layout, bool, const void* a, const void* b, void* out, void* trace,
                           size_t batch, hipStream_t s) {
  return by_layout<4>(p, mode, layout, a, b, out, trace, batch, s);
}
#else
hipError_t launch_cg_part6(const tn_plan* p, int mode, int, int layout, bool, const void* a, const void* b, void* out, void* trace,
                           size_t batch, hipStream_t s) {
  return by_layout<8>(p, mode, layout, a, b, out, trace, batch, s);
}
#endif
#endif

}  // namespace tn

#if defined(TN_CG_STAMPS) && TN_CG_PART == 6
// diagnostic builds only (tools/gpu_cg_stamps.py): the stamps of the last constant-geometry product launch
extern "C" size_t tn_debug_cg_stamps(void* host, size_t max_bytes) {
  (void)hipDeviceSynchronize();
  const size_t nb = tn::tn_cg_stamp_size() < max_bytes ? tn::tn_cg_stamp_size() : max_bytes;
  if (nb) (void)hipMemcpy(host, tn::tn_cg_stamp_ptr(), nb, hipMemcpyDeviceToHost);
  return nb;
}
#endif
