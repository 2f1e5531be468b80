// 8-byte {tag, value} granules: the cheapest way to hand a few KB from one workgroup to another INSIDE a launch on gfx950 (8 XCDs with
// private, mutually non-coherent L2s).  One naturally aligned agent-scope store publishes data and "ready" together (global_store
// ... sc1, written through to memory); a relaxed agent-scope load (global_load ... sc1) reads past the L1.  No fences, no flags: the
// consumer re-reads a granule until its tag is the one it expects (MI355X_MICROARCH.md, price list rows handoff-1to1 / allgather).
#pragma once
#include "dl_common.h"

namespace dl {

typedef unsigned long long u64_t;
typedef __attribute__((address_space(1))) u64_t gu64_t;
typedef __attribute__((address_space(1))) uint32_t gu32_t;

__device__ __forceinline__ void gr_store(u64_t* g, uint32_t tag, uint32_t val) {
  __hip_atomic_store((gu64_t*)(g), ((u64_t)tag << 32) | (u64_t)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_t gr_load(const u64_t* g) {
  return __hip_atomic_load((const gu64_t*)(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ctr_load(const uint32_t* g) {
  return __hip_atomic_load((const gu32_t*)(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace dl
