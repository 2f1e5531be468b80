// LDS-DMA building blocks shared by the kernels that run a loader wave next to consumer waves (decode_block.hip, gemm_smallm.hip variant 4).
#pragma once
#include "dl_common.h"

namespace dl {

#define DL_GLOBAL __attribute__((address_space(1)))
typedef DL_GLOBAL const uint16_t* bgc16_t;
typedef DL_GLOBAL uint16_t* bg16_t;
typedef uint32_t bu32x4_t __attribute__((ext_vector_type(4)));
// LDS through address-space-3 pointers ONLY: a generic (flat) access also waits on vmcnt -- behind a volatile control word the compiler
// emitted flat_load ... s_waitcnt vmcnt(0), which drained the loader's DMA queue on every poll
#define DL_LDS __attribute__((address_space(3)))
typedef DL_LDS volatile int* lvi_t;
typedef DL_LDS unsigned char* l8_t;
__device__ __forceinline__ uint4 lds_ld16(l8_t p) {
  const bu32x4_t r = *(DL_LDS const bu32x4_t*)p;
  return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void lds_st16(l8_t p, const uint4& v) {
  bu32x4_t r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  *(DL_LDS bu32x4_t*)p = r;
}
// control words are wave-uniform by construction: say so (scalar branches, SGPR loop state; m0 needs an SGPR)
__device__ __forceinline__ int lds_ld(lvi_t p) { return __builtin_amdgcn_readfirstlane(*p); }
__device__ __forceinline__ int lds_add(lvi_t p, int v) { return __hip_atomic_fetch_add((DL_LDS int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

}  // namespace dl
