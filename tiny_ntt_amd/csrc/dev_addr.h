// dev_addr.h — device-only addressing helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "modarith.h"

#ifndef TN_SADDR
#define TN_SADDR 1               // 1: operand rows are addressed as scalar base (+ register offset, scalar unit) + 32-bit thread offset
#endif

namespace tn {

// A global-memory pointer the compiler must keep in scalar registers (both halves through wave_uniform): the access it
// bases is then "scalar base + 32-bit thread offset" (global_load ... v_off, s[base]) and the base arithmetic stays on
// the scalar unit.  (The explicit address space keeps the access a global_* instruction after the integer round trip.)
#define TN_GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ TN_GLOBAL_AS T* uniform_ptr(T* p) {
#if TN_SADDR
  const unsigned long long v = (unsigned long long)p;
  return (TN_GLOBAL_AS T*)(((unsigned long long)wave_uniform((u32)(v >> 32)) << 32) | wave_uniform((u32)v));
#else
  return (TN_GLOBAL_AS T*)p;
#endif
}


// A (wave-uniform) pointer the compiler cannot see through at this point: everything derived from it is computed after this
// point — used at the top of a persistent row loop on the TABLE pointers, whose per-column bases (table + e * stride) are
// otherwise hoisted out of the loop, do not fit the scalar registers there and come back through v_readlane.
template <typename T> __device__ __forceinline__ T* opaque_sptr(T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// *p for a pointer in the global address space (the host pass of hipcc cannot copy a struct out of an address-space-qualified
// lvalue; it never runs this)
template <typename T> __device__ __forceinline__ T ld_global(const TN_GLOBAL_AS T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  (void)p;
  return T();
#endif
}

}  // namespace tn
