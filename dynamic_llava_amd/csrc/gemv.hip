// Decode-time weight streaming: y[b,:] = W @ x[b,:] for B <= 8 rows (the batch-1 decode layer reads 405 MB of
// weights per layer and does ~1 flop/byte: pure HBM streaming, no MFMA).  Replaces, for small B, the
// torch/hipBLASLt GEMMs of DML:1011-1013 (q/k/v_proj), DML:1127 (o_proj), DML:328 (gate/up/down_proj) and
// DML:2709 (lm_head) together with the element-wise ops around them, which become PROLOGUES of the stream:
//   DL_GEMV_ADDNORM : x = rmsnorm(h += delta) * w   (DML:1289/1295 residual add + DML:134-139), h written back
//   DL_GEMV_SILUMUL : x = cast(cast(silu(g)) * u)    (DML:328)
//   DL_GEMV_PLAIN   : x as given
// so a decode layer is 5 weight-streaming launches + attention instead of 10 launches.
//
// Mapping: x (B rows, model dtype, after the prologue) sits in LDS; a wave owns R=2 output neurons at a time and its
// 64 lanes stride the K dimension in 16-byte chunks (one wave-instruction = 1 KiB of one weight row, fully
// coalesced, non-temporal: every weight byte is used exactly once per step).  fp32 accumulate, 6-step wave
// reduction, one rounding to the model dtype -- the same contract as the GEMM it replaces.
#include "dl_common.h"

namespace dl {

constexpr int kGemvThreads = 256;
constexpr int kGemvR = 2;       // neurons per wave per pass
constexpr int kGemvMaxB = 8;

template <typename T>
__device__ __forceinline__ void load16_nt(const void* p, float (&f)[Elem<T>::kVec]) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));  // global_load_dwordx4 ... nt
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

template <typename T, int B, int MODE>
__global__ __launch_bounds__(kGemvThreads) void gemv_kernel(const void* __restrict__ W_, int N, int K, const void* x_, int64_t x_rs,
                                                            const void* __restrict__ h_, void* __restrict__ h_out_,
                                                            const void* __restrict__ delta_, const void* __restrict__ nw_,
                                                            float eps, void* __restrict__ y_, int64_t y_rs) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* xs = reinterpret_cast<S*>(smem);  // [B][K] in the model dtype
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nvec = K / V;

  // ---- prologue: build x in LDS ----
  if constexpr (MODE == 1) {  // ADDNORM
    const S* h = reinterpret_cast<const S*>(h_);
    S* h_out = reinterpret_cast<S*>(h_out_);
    const S* dl_ = reinterpret_cast<const S*>(delta_);
    const S* nw = reinterpret_cast<const S*>(nw_);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float ss = 0.f;
      for (int v = tid; v < nvec; v += kGemvThreads) {
        float a[V];
        load16<T>(h + (int64_t)b * K + v * V, a);
        if (dl_) {
          float d[V];
          load16<T>(dl_ + (int64_t)b * K + v * V, d);
#pragma unroll
          for (int e = 0; e < V; ++e) a[e] = Elem<T>::round(a[e] + d[e]);
          // updated residual stream: written once, to a DIFFERENT buffer (other workgroups are still reading h_in)
          if (blockIdx.x == 0) store16<T>(h_out + (int64_t)b * K + v * V, a);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) ss += a[e] * a[e];
        store16<T>(xs + b * K + v * V, a);
      }
      const float rstd = rsqrtf(block_sum<4>(ss, red) / (float)K + eps);
      for (int v = tid; v < nvec; v += kGemvThreads) {
        float a[V], w[V];
        load16<T>(xs + b * K + v * V, a);
        load16<T>(nw + v * V, w);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] = w[e] * Elem<T>::round(a[e] * rstd);
        store16<T>(xs + b * K + v * V, a);
      }
    }
  } else if constexpr (MODE == 2) {  // SILUMUL: x_ = gate_up [B, 2K]
    const S* gu = reinterpret_cast<const S*>(x_);
#pragma unroll
    for (int b = 0; b < B; ++b)
      for (int v = tid; v < nvec; v += kGemvThreads) {
        float g[V], u[V];
        load16<T>(gu + (int64_t)b * x_rs + v * V, g);
        load16<T>(gu + (int64_t)b * x_rs + K + v * V, u);
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = Elem<T>::round(g[e] / (1.0f + expf(-g[e]))) * u[e];
        store16<T>(xs + b * K + v * V, g);
      }
  } else {
    const S* x = reinterpret_cast<const S*>(x_);
#pragma unroll
    for (int b = 0; b < B; ++b)
      for (int v = tid; v < nvec; v += kGemvThreads)
        *reinterpret_cast<uint4*>(xs + b * K + v * V) = *reinterpret_cast<const uint4*>(x + (int64_t)b * x_rs + v * V);
  }
  __syncthreads();

  // ---- stream the weights ----
  const S* W = reinterpret_cast<const S*>(W_);
  const int groups = (N + 4 * kGemvR - 1) / (4 * kGemvR);
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int n0 = grp * 4 * kGemvR + wid * kGemvR;
    float acc[kGemvR][B];
#pragma unroll
    for (int r = 0; r < kGemvR; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
    const S* w0 = W + (int64_t)(n0 < N ? n0 : N - 1) * K;
    const S* w1 = W + (int64_t)(n0 + 1 < N ? n0 + 1 : N - 1) * K;
    int v = lane;
    for (; v + 192 < nvec; v += 256) {  // 4 chunks x 2 rows = 8 independent 16-byte loads in flight per lane
      float wa[4][V], wb[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load16_nt<T>(w0 + (v + 64 * u) * V, wa[u]);
        load16_nt<T>(w1 + (v + 64 * u) * V, wb[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int b = 0; b < B; ++b) {
          float xv[V];
          load16<T>(xs + b * K + (v + 64 * u) * V, xv);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            acc[0][b] = fmaf(wa[u][e], xv[e], acc[0][b]);
            acc[1][b] = fmaf(wb[u][e], xv[e], acc[1][b]);
          }
        }
    }
    for (; v < nvec; v += 64) {
      float wa[V], wb[V];
      load16_nt<T>(w0 + v * V, wa);
      load16_nt<T>(w1 + v * V, wb);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float xv[V];
        load16<T>(xs + b * K + v * V, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          acc[0][b] = fmaf(wa[e], xv[e], acc[0][b]);
          acc[1][b] = fmaf(wb[e], xv[e], acc[1][b]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kGemvR; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float a = wave_sum(acc[r][b]);
        if (lane == 0 && n0 + r < N) store1<T>(y_, (int64_t)b * y_rs + n0 + r, a);
      }
  }
}

template <typename T, int B>
static int gemv_launch(int mode, const void* W, int N, int K, const void* x, int64_t x_rs, const void* h, void* h_out, const void* delta, const void* nw,
                       float eps, void* y, int64_t y_rs, hipStream_t st) {
  const size_t smem = (size_t)B * K * Elem<T>::kBytes;
  const int groups = (N + 4 * kGemvR - 1) / (4 * kGemvR);
  const int grid = groups < 2048 ? groups : 2048;
#define DL_GEMV_GO(MODE)                                                                                                       \
  {                                                                                                                            \
    auto kfn = gemv_kernel<T, B, MODE>;                                                                                        \
    if (smem > 64 * 1024) {                                                                                                    \
      static bool attr_set = false;                                                                                            \
      if (!attr_set) {                                                                                                         \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) { \
          (void)hipGetLastError(); /* do not leave a sticky error behind */                                                    \
          set_error("dl_gemv: cannot raise the dynamic LDS limit to 152 KiB");                                                 \
          return DL_ERR_LAUNCH;                                                                                                \
        }                                                                                                                      \
        attr_set = true;                                                                                                       \
      }                                                                                                                        \
    }                                                                                                                          \
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kGemvThreads), smem, st, W, N, K, x, x_rs, h, h_out, delta, nw, eps, y, y_rs); \
  }
  if (mode == DL_GEMV_ADDNORM) DL_GEMV_GO(1)
  else if (mode == DL_GEMV_SILUMUL) DL_GEMV_GO(2)
  else DL_GEMV_GO(0)
#undef DL_GEMV_GO
  return DL_OK;
}

}  // namespace dl

using namespace dl;

extern "C" int dl_gemv_max_batch(int K, int dtype) {
  const int es = dtype == DL_F32 ? 4 : 2;
  int b = (int)((size_t)(150 * 1024) / ((size_t)K * es));
  return b > kGemvMaxB ? kGemvMaxB : b;
}

extern "C" int dl_gemv(int mode, const void* W, int N, int K, const void* x, int64_t x_row_stride, const void* h_in, void* h_out,
                       const void* delta, const void* norm_w, float eps, void* y, int64_t y_row_stride, int B, int dtype, void* stream) {
  const void* h = h_in;
  DL_REQUIRE(W && y, "dl_gemv: NULL pointer");
  DL_REQUIRE(N > 0 && K > 0 && B > 0, "dl_gemv: bad shape");
  DL_REQUIRE(mode == DL_GEMV_PLAIN || mode == DL_GEMV_ADDNORM || mode == DL_GEMV_SILUMUL, "dl_gemv: bad mode %d", mode);
  DL_REQUIRE(mode == DL_GEMV_ADDNORM ? (h && norm_w) : (x != nullptr), "dl_gemv: missing operand for mode %d", mode);
  DL_REQUIRE(!(mode == DL_GEMV_ADDNORM && delta) || (h_out && h_out != h_in), "dl_gemv: h_out must be a distinct buffer when delta is given");
  DL_REQUIRE(B <= dl_gemv_max_batch(K, dtype), "dl_gemv: B=%d rows of K=%d do not fit in LDS (max %d)", B, K, dl_gemv_max_batch(K, dtype));
  hipStream_t st = as_stream(stream);
  int rc = DL_OK;
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(K % Elem<T>::kVec == 0 && (mode == DL_GEMV_ADDNORM || x_row_stride % Elem<T>::kVec == 0), "dl_gemv: K / strides must be multiples of %d", Elem<T>::kVec);
    switch (B) {
      case 1: rc = gemv_launch<T, 1>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      case 2: rc = gemv_launch<T, 2>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      case 3: rc = gemv_launch<T, 3>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      case 4: rc = gemv_launch<T, 4>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      case 5: rc = gemv_launch<T, 5>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      case 6: rc = gemv_launch<T, 6>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      case 7: rc = gemv_launch<T, 7>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
      default: rc = gemv_launch<T, 8>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, st); break;
    }
  });
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_gemv");
  return DL_OK;
}
