// dl_linear: C = epilogue(A @ W^T + bias) for the predictor networks (VisionPredictor DML:1324-1346,
// CTL:107-123,146-180).  nn.Linear layout: A [M,K] and W [N,K] are both K-contiguous, which is exactly the
// MFMA A/B fragment shape (8 consecutive k per lane) -> both operands are 16-byte LDS reads.
//
// f16/bf16: 64x64 output tile per 256-thread workgroup, 4 waves as 2x2, each wave 32x32 = 2x2 tiles of
// v_mfma_f32_16x16x32; BK = 64 staged through padded LDS.  The epilogue reproduces the eager reference's
// rounding points: round after (A W^T + b), after GELU, after the residual add.
// f32: plain FMA tile kernel (parity/debug path).
#include "dl_common.h"

namespace dl {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ f32x4_t mfma16_lin(const uint4& a, const uint4& b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma16_lin<bf16_t>(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma16_lin<f16_t>(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

constexpr int kTM = 64, kTN = 64;

template <typename T>
__device__ __forceinline__ float epilogue(float acc, float bias, const void* R, int64_t ridx, int flags) {
  float v = Elem<T>::round(acc + bias);
  if (flags & DL_EPI_GELU) v = Elem<T>::round(gelu_erf(v));
  if (flags & DL_EPI_RESIDUAL) v = Elem<T>::round(load1<T>(R, ridx) + v);
  return v;
}

// TK: K slab per step (64; 128 for K >= 1024 -- half the barriers of the long serial K loops of in_conv / the FF down projection)
template <typename T, int kTK>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const void* __restrict__ A_, int64_t lda, const void* __restrict__ W_,
                                                           const void* __restrict__ bias_, void* C_, int64_t ldc, const void* R_,
                                                           int64_t ldr, int M, int N, int K, int flags) {
  using S = uint16_t;
  constexpr int kLd = kTK + 8, CPR = kTK / 8, IT = (kTM * CPR) / 256;
  __shared__ __attribute__((aligned(16))) S As[kTM * kLd];
  __shared__ __attribute__((aligned(16))) S Ws[kTN * kLd];
  const S* A = reinterpret_cast<const S*>(A_);
  const S* W = reinterpret_cast<const S*>(W_);
  const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // software-pipelined over K: the next 64-wide slab of A and W is in flight in registers while the MFMAs of the current one run
  // (the launches here are small -- 36 to 288 workgroups, up to 64 K steps -- so an exposed global round trip per step was most of
  // their time: in_conv [576,4096]x[512,4096] 100 -> see DESIGN)
  uint4 a4[IT], w4[IT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {  // 64 rows x CPR chunks per operand
      const int idx = it * 256 + tid;
      const int r = idx / CPR, ch = (idx % CPR) * 8;
      a4[it] = make_uint4(0, 0, 0, 0);
      w4[it] = make_uint4(0, 0, 0, 0);
      if (k0 + ch < K) {
        if (m0 + r < M) a4[it] = *reinterpret_cast<const uint4*>(A + (int64_t)(m0 + r) * lda + k0 + ch);
        if (n0 + r < N) w4[it] = *reinterpret_cast<const uint4*>(W + (int64_t)(n0 + r) * K + k0 + ch);
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += kTK) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = it * 256 + tid;
      const int r = idx / CPR, ch = (idx % CPR) * 8;
      *reinterpret_cast<uint4*>(As + r * kLd + ch) = a4[it];
      *reinterpret_cast<uint4*>(Ws + r * kLd + ch) = w4[it];
    }
    __syncthreads();
    if (k0 + kTK < K) fetch(k0 + kTK);
#pragma unroll
    for (int ks = 0; ks < kTK / 32; ++ks) {
      uint4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const uint4*>(As + (wr * 32 + i * 16 + lr) * kLd + ks * 32 + lg * 8);
        bf[i] = *reinterpret_cast<const uint4*>(Ws + (wc * 32 + i * 16 + lr) * kLd + ks * 32 + lg * 8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_lin<T>(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }
  // C layout of a 16x16 tile: row = lg*4 + r, col = lr
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 32 + j * 16 + lr;
      if (col >= N) continue;
      const float bv = bias_ ? load1<T>(bias_, col) : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + i * 16 + lg * 4 + r;
        if (row < M) store1<T>(C_, (int64_t)row * ldc + col, epilogue<T>(acc[i][j][r], bv, R_, (int64_t)row * ldr + col, flags));
      }
    }
}

// ---- f32 reference-grade tile kernel: 64x64 tile, each thread a 4x4 micro-tile ----
template <typename T>
__global__ __launch_bounds__(256) void linear_simple_kernel(const void* __restrict__ A_, int64_t lda, const void* __restrict__ W_,
                                                             const void* __restrict__ bias_, void* C_, int64_t ldc, const void* R_,
                                                             int64_t ldr, int M, int N, int K, int flags) {
  constexpr int BK = 16;
  __shared__ float As[64][BK + 1];
  __shared__ float Ws[64][BK + 1];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += BK) {
    for (int idx = tid; idx < 64 * BK; idx += 256) {
      const int r = idx / BK, c = idx % BK;
      As[r][c] = (m0 + r < M && k0 + c < K) ? load1<T>(A_, (int64_t)(m0 + r) * lda + k0 + c) : 0.f;
      Ws[r][c] = (n0 + r < N && k0 + c < K) ? load1<T>(W_, (int64_t)(n0 + r) * K + k0 + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < BK; ++c) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = As[ty * 4 + i][c];
        b[i] = Ws[tx * 4 + i][c];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + ty * 4 + i;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + tx * 4 + j;
      if (col >= N) continue;
      const float bv = bias_ ? load1<T>(bias_, col) : 0.f;
      store1<T>(C_, (int64_t)row * ldc + col, epilogue<T>(acc[i][j], bv, R_, (int64_t)row * ldr + col, flags));
    }
  }
}

int linear_launch(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, const void* R, int64_t ldr, int M,
                  int N, int K, int flags, int dtype, hipStream_t st) {
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
  if (dtype == DL_F32) {
    hipLaunchKernelGGL((linear_simple_kernel<f32_t>), grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags);
  } else if (dtype == DL_BF16) {
    if (K >= 1024) hipLaunchKernelGGL((linear_mfma_kernel<bf16_t, 128>), grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags);
    else hipLaunchKernelGGL((linear_mfma_kernel<bf16_t, 64>), grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags);
  } else if (dtype == DL_F16) {
    if (K >= 1024) hipLaunchKernelGGL((linear_mfma_kernel<f16_t, 128>), grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags);
    else hipLaunchKernelGGL((linear_mfma_kernel<f16_t, 64>), grid, dim3(256), 0, st, A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags);
  } else {
    return DL_ERR_ARG;
  }
  return DL_OK;
}

}  // namespace dl

using namespace dl;

extern "C" int dl_linear(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, const void* R, int64_t ldr,
                         int M, int N, int K, int flags, int dtype, void* stream) {
  DL_REQUIRE(A && W && C, "dl_linear: NULL pointer");
  DL_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0, "dl_linear: bad shape M=%d N=%d K=%d lda=%lld", M, N, K, (long long)lda);
  DL_REQUIRE(!(flags & DL_EPI_RESIDUAL) || R, "dl_linear: residual requested but R is NULL");
  DL_REQUIRE(dtype == DL_F32 || dtype == DL_F16 || dtype == DL_BF16, "dl_linear: unsupported dtype %d", dtype);
  if (M == 0) return DL_OK;
  linear_launch(A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags, dtype, as_stream(stream));
  DL_CHECK_LAUNCH("dl_linear");
  return DL_OK;
}
