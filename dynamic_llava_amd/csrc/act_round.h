// Epilogue arithmetic of the MFMA GEMMs: hardware fp32 -> bf16 conversion and activations whose 16-bit ROUNDED value equals the exact expression's at a fraction of
// its VALU cost.  An epilogue is owned by the four consumer waves of a workgroup: exact expf + an IEEE divide + software bf16 roundings cost dl_linear_tiles' fc1
// 8 us of an 18 us launch (round 6) and dl_linear_packed's gate|up 9 us of 59 (measured as the difference to its partial-sum epilogue).
#pragma once
#include "dl_common.h"

namespace dl {

// gfx950 converts fp32 -> bf16 in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even: the same bits as Elem<bf16_t>::from_f for every finite value --
// tests/test_linear_tiles_gpu.py holds the two against each other)
template <typename T>
__device__ __forceinline__ uint32_t hw_pack2(float a, float b) {
  if constexpr (Elem<T>::kBf16) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
  } else {
    return (uint32_t)Elem<T>::from_f(a) | ((uint32_t)Elem<T>::from_f(b) << 16);
  }
}
template <typename T>
__device__ __forceinline__ float hw_round(float a) {
  if constexpr (Elem<T>::kBf16)
    return __uint_as_float(hw_pack2<T>(a, 0.f) << 16);
  else
    return Elem<T>::round(a);
}

// distance (in fp32 ulps) of a value from the nearest rounding boundary of the 16-bit type is <= guard  (fp16: normal range only -- callers send small results
// to the exact path)
template <typename T>
__device__ __forceinline__ bool near_rounding_tie(float v, uint32_t guard) {
  constexpr int kDrop = Elem<T>::kBf16 ? 16 : 13;  // fp32 mantissa bits the 16-bit type drops
  const uint32_t low = __float_as_uint(v) & ((1u << kDrop) - 1u);
  const uint32_t half = 1u << (kDrop - 1);
  return (low > half ? low - half : half - low) <= guard;
}

// cast(silu(g)) = cast(g / (1 + expf(-g))): dl_silu_mul's first rounding (DML:328).  v_exp_f32 / v_rcp_f32 / one multiply are within a few fp32 ulps of the exact
// expression, so the ROUNDED value can only differ when the fp32 result sits within 64 ulps of a rounding boundary (2e-3 of the values for bf16), where the fast forms
// flush (|g| large), or -- fp16 -- where the result is subnormal in the type (another boundary grid): those lanes evaluate the exact expression.
template <typename T>
__device__ __forceinline__ float silu_rounded(float g) {
  const float fast = g * __builtin_amdgcn_rcpf(1.0f + __expf(-g));
  const bool in_range = fabsf(g) < (Elem<T>::kBf16 ? 16.0f : 8.0f) && (Elem<T>::kBf16 || fabsf(fast) > 1.0e-4f);  // (v_exp_f32's argument -g log2(e) is rounded: |g| 2^-24 relative in the result)
  float s = fast;
  if (near_rounding_tie<T>(fast, 64) || !in_range) s = g / (1.0f + expf(-g));
  return hw_round<T>(s);
}

// cast(sigmoid(t)) = cast(1 / (1 + expf(-t))): dl_quick_gelu's second rounding, same scheme (dl_linear_tiles' QuickGELU epilogue holds its own copy of this function)
template <typename T>
__device__ __forceinline__ float sigmoid_rounded(float t) {
  const float fast = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
  const bool in_range = fabsf(t) < (Elem<T>::kBf16 ? 16.0f : 8.0f);  // fp16: sigmoid below 2^-14 is subnormal in the type (another boundary grid)
  float s = fast;
  if (near_rounding_tie<T>(fast, 64) || !in_range) s = 1.0f / (1.0f + expf(-t));
  return hw_round<T>(s);
}

}  // namespace dl
