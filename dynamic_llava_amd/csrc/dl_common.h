// Shared device/host helpers for libdynllava_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dynllava.h"

namespace dl {

// ---------------------------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define DL_REQUIRE(cond, ...)             \
  do {                                    \
    if (!(cond)) {                        \
      dl::set_error(__VA_ARGS__);         \
      return DL_ERR_ARG;                  \
    }                                     \
  } while (0)

#define DL_CHECK_LAUNCH(name)                                                  \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      dl::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
      return DL_ERR_LAUNCH;                                                    \
    }                                                                          \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------------------------
// element types: storage is raw bits; arithmetic is fp32; rounding is round-to-nearest-even,
// i.e. what torch does when an eager op writes a bf16 / fp16 tensor.
// ---------------------------------------------------------------------------------------------
struct f32_t {
  float v;
};
struct f16_t {
  uint16_t v;
};
struct bf16_t {
  uint16_t v;
};

__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_float(uint16_t b) {
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}
__device__ __forceinline__ uint16_t float_to_f16_bits(float f) {
  // The value must exist as an fp32 number before it is rounded to fp16, as in the eager reference (fp32 op, then a cast).
  // Without the barrier the compiler folds `cast(a * b)` into v_fma_mixlo_f16, which rounds the exact product ONCE: a different
  // result whenever the fp32 product lands on an fp16 tie (e.g. 1.702f * x in QuickGELU: 4e-4 of all elements).
  asm("" : "+v"(f));
  _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE
  uint16_t b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}

template <typename T>
struct Elem;
template <>
struct Elem<f32_t> {
  static constexpr int kBytes = 4;
  static constexpr bool kBf16 = false;
  static constexpr int kVec = 4;  // elements per 16-byte access
  using storage = float;
  __device__ static __forceinline__ float to_f(float s) { return s; }
  __device__ static __forceinline__ float from_f(float f) { return f; }
  __device__ static __forceinline__ float round(float f) { return f; }
};
template <>
struct Elem<f16_t> {
  static constexpr int kBytes = 2;
  static constexpr bool kBf16 = false;
  static constexpr int kVec = 8;
  using storage = uint16_t;
  __device__ static __forceinline__ float to_f(uint16_t s) { return f16_bits_to_float(s); }
  __device__ static __forceinline__ uint16_t from_f(float f) { return float_to_f16_bits(f); }
  __device__ static __forceinline__ float round(float f) { return f16_bits_to_float(float_to_f16_bits(f)); }
};
template <>
struct Elem<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr bool kBf16 = true;
  static constexpr int kVec = 8;
  using storage = uint16_t;
  __device__ static __forceinline__ float to_f(uint16_t s) { return bf16_bits_to_float(s); }
  __device__ static __forceinline__ uint16_t from_f(float f) { return float_to_bf16_bits(f); }
  __device__ static __forceinline__ float round(float f) { return bf16_bits_to_float(float_to_bf16_bits(f)); }
};

// 16-byte vector of elements <-> kVec floats
template <typename T>
struct Vec16 {
  uint4 raw;
};

template <typename T>
__device__ __forceinline__ void load16(const void* p, float (&f)[Elem<T>::kVec]) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
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

template <typename T>
__device__ __forceinline__ uint4 pack16(const float (&f)[Elem<T>::kVec]) {
  uint4 r;
  if constexpr (Elem<T>::kVec == 4) {
    r.x = __float_as_uint(f[0]);
    r.y = __float_as_uint(f[1]);
    r.z = __float_as_uint(f[2]);
    r.w = __float_as_uint(f[3]);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)Elem<T>::from_f(f[2 * i]) | ((uint32_t)Elem<T>::from_f(f[2 * i + 1]) << 16);
    r.x = w[0];
    r.y = w[1];
    r.z = w[2];
    r.w = w[3];
  }
  return r;
}

template <typename T>
__device__ __forceinline__ void store16(void* p, const float (&f)[Elem<T>::kVec]) {
  *reinterpret_cast<uint4*>(p) = pack16<T>(f);
}

template <typename T>
__device__ __forceinline__ float load1(const void* base, int64_t idx) {
  return Elem<T>::to_f(reinterpret_cast<const typename Elem<T>::storage*>(base)[idx]);
}
template <typename T>
__device__ __forceinline__ void store1(void* base, int64_t idx, float f) {
  reinterpret_cast<typename Elem<T>::storage*>(base)[idx] = Elem<T>::from_f(f);
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64)
// ---------------------------------------------------------------------------------------------
// Cross-lane reductions inside an aligned group of 16 lanes (one DPP "row") as VALU DPP modifiers -- a few cycles per step.
// __shfl_xor compiles to ds_bpermute_b32 (the LDS crossbar: ~100 cycles of latency per step and a slot in the LDS pipe), which made
// the 4-step reductions of the attention kernels (per-key dot products, softmax row max / sum) their longest dependency chains.
// Steps: xor 1 and xor 2 as quad permutes, then the two mirrors: after the quad steps every lane of a quad holds the quad's value, so
// mirroring within 8 and within 16 lanes pairs each quad / half with its partner.  All 16 lanes end up with bit-identical results
// (fp add / max are commutative; the association differs from the xor butterfly by rounding only).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror
constexpr int kDppMirror = 0x140;       // row_mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<kDppXor1>(v);
  v += dpp_f32<kDppXor2>(v);
  v += dpp_f32<kDppHalfMirror>(v);
  v += dpp_f32<kDppMirror>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<kDppXor1>(v));
  v = fmaxf(v, dpp_f32<kDppXor2>(v));
  v = fmaxf(v, dpp_f32<kDppHalfMirror>(v));
  v = fmaxf(v, dpp_f32<kDppMirror>(v));
  return v;
}
__device__ __forceinline__ float row8_sum(float v) {  // aligned groups of 8 lanes
  v += dpp_f32<kDppXor1>(v);
  v += dpp_f32<kDppXor2>(v);
  v += dpp_f32<kDppHalfMirror>(v);
  return v;
}
// whole wave (all 64 lanes active): DPP inside the four 16-lane rows, then the four row results through v_readlane (scalar
// registers, wave-uniform) -- no LDS crossbar at all.  Every lane gets the same value.
// "These values must be in registers NOW": an empty asm that claims to modify them.  Without it the compiler is free to sink a load below a later
// polling loop (nothing orders a plain load against relaxed atomic loads), which turns a prefetch into a cold round trip on the critical path.
__device__ __forceinline__ void pin_reg(uint4& x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
__device__ __forceinline__ void pin_reg(float& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ float wave_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (wave_lane(v, 0) + wave_lane(v, 16)) + (wave_lane(v, 32) + wave_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(wave_lane(v, 0), wave_lane(v, 16)), fmaxf(wave_lane(v, 32), wave_lane(v, 48)));
}
// sum over a block of NW waves; `red` is LDS scratch of >= NW floats; result broadcast to all threads.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// dl_linear_packed's activation order (include/dynllava.h, dl_pack_x_tiles): element offset of the 16-byte chunk k8 = k / 8 of `row` when the
// matrix has n_tiles 16-row tiles -- Xp[step = k / 64][tile][k half][lane = 16 ((k % 32) / 8) + row % 16][8].  2-byte element types only.
__device__ __forceinline__ int64_t lp_x_chunk_offset(int64_t row, int k8, int n_tiles) {
  const int step = k8 >> 3, half = (k8 >> 2) & 1, lg = k8 & 3;
  return ((((int64_t)step * n_tiles + (row >> 4)) * 2 + half) * 64 + lg * 16 + (row & 15)) * 8;
}
static inline int lp_x_tiles(int64_t rows) { return (int)((((rows + 15) / 16 + 3) / 4) * 4); }

// dtype dispatch on the host
#define DL_DISPATCH_DTYPE(dtype, T, ...)                 \
  switch (dtype) {                                       \
    case DL_F32: {                                       \
      using T = dl::f32_t;                               \
      __VA_ARGS__;                                       \
    } break;                                             \
    case DL_F16: {                                       \
      using T = dl::f16_t;                               \
      __VA_ARGS__;                                       \
    } break;                                             \
    case DL_BF16: {                                      \
      using T = dl::bf16_t;                              \
      __VA_ARGS__;                                       \
    } break;                                             \
    default:                                             \
      dl::set_error("unsupported dtype %d", (int)dtype); \
      return DL_ERR_ARG;                                 \
  }

// ---- split-KV partials of the decode attention: entry (row b, head h, split s) = [m, l, -, -, o[0..D)] floats (D + 4: the
// o vector is 16-byte aligned).  attn_split_merge: the merged, normalised output for NV consecutive head dims starting at d0, in
// split order, 8 splits per round trip. ----
constexpr int kAttnPartPad = 4;
template <int NV>
__device__ __forceinline__ void attn_split_merge(const float* __restrict__ p, int n_splits, int D, int d0, float (&out)[NV]) {
  const int stride = D + kAttnPartPad;
  float M = -INFINITY, L = 0.f, O[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) O[e] = 0.f;
  for (int s0 = 0; s0 < n_splits; s0 += 8) {
    float m8[8], l8[8], o8[8][NV];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u < n_splits ? s0 + u : n_splits - 1;
      const float* q = p + (int64_t)s * stride;
      m8[u] = q[0];
      l8[u] = q[1];
      if constexpr (NV == 4) {
        const float4 v = *reinterpret_cast<const float4*>(q + kAttnPartPad + d0);
        o8[u][0] = v.x;
        o8[u][1] = v.y;
        o8[u][2] = v.z;
        o8[u][3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < NV; ++e) o8[u][e] = q[kAttnPartPad + d0 + e];
      }
    }
    float mc = M;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < n_splits) mc = fmaxf(mc, m8[u]);
    if (mc > -INFINITY) {
      const float a = __expf(M - mc);  // M = -inf -> 0
      L *= a;
#pragma unroll
      for (int e = 0; e < NV; ++e) O[e] *= a;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < n_splits) {
          const float w = __expf(m8[u] - mc);  // empty split: exp(-inf) = 0
          L += l8[u] * w;
#pragma unroll
          for (int e = 0; e < NV; ++e) O[e] += o8[u][e] * w;
        }
      M = mc;
    }
  }
#pragma unroll
  for (int e = 0; e < NV; ++e) out[e] = L > 0.f ? O[e] / L : 0.f;
}

}  // namespace dl
