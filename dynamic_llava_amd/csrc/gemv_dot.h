// Weight-streaming helpers shared by the launch-path GEMV (gemv.hip) and the persistent decode step (decode_persistent.hip): the two
// must produce bit-identical sums, so they use the same per-chunk dot product and the same load flavour.
#pragma once
#include "dl_common.h"

namespace dl {

// raw 16-byte chunk -> kVec floats
template <typename T>
__device__ __forceinline__ void unpack16(const uint4& r, float (&f)[Elem<T>::kVec]) {
  if constexpr (Elem<T>::kVec == 4) {
    f[0] = __uint_as_float(r.x);
    f[1] = __uint_as_float(r.y);
    f[2] = __uint_as_float(r.z);
    f[3] = __uint_as_float(r.w);
  } else {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = Elem<T>::to_f((uint16_t)(w[i] & 0xffffu));
      f[2 * i + 1] = Elem<T>::to_f((uint16_t)(w[i] >> 16));
    }
  }
}
// 8 (bf16/f16) or 4 (f32) products of one 16-byte weight chunk with one 16-byte x chunk, accumulated in fp32.  16-bit dtypes use
// v_dot2c_f32_{bf16,f16} on the packed words: no unpacking, 4 VALU ops per chunk instead of ~24 -- this is what keeps batches of
// 2..8 rows bound by the weight stream rather than by the vector ALU.
typedef __bf16 gv_bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 gv_f16x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ float dot16(const uint4& w, const uint4& x, float acc) {
  if constexpr (Elem<T>::kVec == 4) {
    acc = fmaf(__uint_as_float(w.x), __uint_as_float(x.x), acc);
    acc = fmaf(__uint_as_float(w.y), __uint_as_float(x.y), acc);
    acc = fmaf(__uint_as_float(w.z), __uint_as_float(x.z), acc);
    acc = fmaf(__uint_as_float(w.w), __uint_as_float(x.w), acc);
  } else if constexpr (Elem<T>::kBf16) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gv_bf16x2_t, w.x), __builtin_bit_cast(gv_bf16x2_t, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gv_bf16x2_t, w.y), __builtin_bit_cast(gv_bf16x2_t, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gv_bf16x2_t, w.z), __builtin_bit_cast(gv_bf16x2_t, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gv_bf16x2_t, w.w), __builtin_bit_cast(gv_bf16x2_t, x.w), acc, false);
  } else {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_f16x2_t, w.x), __builtin_bit_cast(gv_f16x2_t, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_f16x2_t, w.y), __builtin_bit_cast(gv_f16x2_t, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_f16x2_t, w.z), __builtin_bit_cast(gv_f16x2_t, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_f16x2_t, w.w), __builtin_bit_cast(gv_f16x2_t, x.w), acc, false);
  }
  return acc;
}

__device__ __forceinline__ uint4 ldg_nt(const void* p) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));  // global_load_dwordx4 ... nt
  return make_uint4(r.x, r.y, r.z, r.w);
}

}  // namespace dl
