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

// A twiddle record as a plain vector value (one 16- / 8-byte global load; arrays of these stay in registers, arrays of the record
// structs did not) and back
typedef u64 tn_u64x2 __attribute__((ext_vector_type(2)));
typedef u32 tn_u32x2 __attribute__((ext_vector_type(2)));
template <typename E> struct TwRawOf;
template <> struct TwRawOf<u64> { typedef tn_u64x2 type; };
template <> struct TwRawOf<u32> { typedef tn_u32x2 type; };
__device__ __forceinline__ Tw64 tw_pack(tn_u64x2 v) { Tw64 t; t.w = v.x; t.wp = v.y; return t; }
__device__ __forceinline__ Tw32 tw_pack(tn_u32x2 v) { Tw32 t; t.w = v.x; t.wp = v.y; return t; }

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
