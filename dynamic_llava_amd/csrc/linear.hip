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

// weights of the split-K kernels are read once, by one workgroup: non-temporal like every other weight stream here
__device__ __forceinline__ uint4 lin_ldg_nt(const void* p) {
  typedef uint32_t lin_u32x4_t __attribute__((ext_vector_type(4)));
  const lin_u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const lin_u32x4_t*>(p));
  return make_uint4(r.x, r.y, r.z, r.w);
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

// ---- split-K variant for the prefill's narrow projections (o_proj [H,H], down_proj [H,I] at M = 100..256 rows): N / 64 x M / 64 tiles
// alone leave most CUs idle behind long serial K loops (the library runs them at 1.5-1.9 TB/s of weight streaming), so K is cut into
// gridDim.z slices and every slice writes an fp32 partial tile; the consumer (dl_add_rmsnorm_parts: residual add + RMSNorm) adds the
// slices in order -- deterministic, one rounding, no reduce launch.  Operands swapped (A = W rows, B = X rows) so that a lane ends up
// with 4 consecutive output columns of one row: 16-byte partial stores.
template <typename T, int kTK>
__global__ __launch_bounds__(256) void linear_splitk_kernel(const void* __restrict__ A_, int64_t lda, const void* __restrict__ W_, float* __restrict__ part,
                                                             int M, int N, int K, int n_slices) {
  using S = uint16_t;
  constexpr int kLd = kTK + 8, CPR = kTK / 8, IT = (kTM * CPR) / 256;
  __shared__ __attribute__((aligned(16))) S As[kTM * kLd];
  __shared__ __attribute__((aligned(16))) S Ws[kTN * kLd];
  const S* A = reinterpret_cast<const S*>(A_);
  const S* W = reinterpret_cast<const S*>(W_);
  const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN, slice = blockIdx.z;
  const int steps = (K + kTK - 1) / kTK;
  const int s0 = (int)((int64_t)steps * slice / n_slices), s1 = (int)((int64_t)steps * (slice + 1) / n_slices);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;  // wave tile: 32 output columns (wr) x 32 rows (wc)
  const int lr = lane & 15, lg = lane >> 4;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 a4[IT], w4[IT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
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
  if (s0 < s1) fetch(s0 * kTK);
  for (int st = s0; st < s1; ++st) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = it * 256 + tid;
      const int r = idx / CPR, ch = (idx % CPR) * 8;
      *reinterpret_cast<uint4*>(As + r * kLd + ch) = a4[it];
      *reinterpret_cast<uint4*>(Ws + r * kLd + ch) = w4[it];
    }
    __syncthreads();
    if (st + 1 < s1) fetch((st + 1) * kTK);
#pragma unroll
    for (int ks = 0; ks < kTK / 32; ++ks) {
      uint4 wf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wf[i] = *reinterpret_cast<const uint4*>(Ws + (wr * 32 + i * 16 + lr) * kLd + ks * 32 + lg * 8);
        xf[i] = *reinterpret_cast<const uint4*>(As + (wc * 32 + i * 16 + lr) * kLd + ks * 32 + lg * 8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_lin<T>(wf[i], xf[j], acc[i][j]);  // D[n][m]
    }
    __syncthreads();
  }
  // D tile: row (n) = lg*4 + r, col (m) = lr  ->  part[slice][m][n .. n+3]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wr * 32 + i * 16 + lg * 4;
      const int m = m0 + wc * 32 + j * 16 + lr;
      if (m < M && n < N) *reinterpret_cast<float4*>(part + ((int64_t)slice * M + m) * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

// ---- the same with ALL rows in one tile (M <= 32 * MTW): every weight byte is read once, by one workgroup per (64 columns, K slice).
// Waves 2 (columns) x 2 (rows): a wave owns 32 columns x 16*MTW rows, so a K step reads 2 W + MTW X fragments from LDS for 2*MTW MFMAs
// (the 64x64 tiling above reads 1 + 1 per MFMA pair and streams W three times at M = 170).
template <typename T, int kTK, int MTW>
__global__ __launch_bounds__(256) void linear_splitk_wide_kernel(const void* __restrict__ A_, int64_t lda, const void* __restrict__ W_, float* __restrict__ part,
                                                                  void* __restrict__ C_, int64_t ldc, int M, int N, int K, int n_slices) {
  using S = uint16_t;
  constexpr int kRows = 32 * MTW;  // rows of the tile
  constexpr int kLd = kTK + 8, CPR = kTK / 8, ITA = (kRows * CPR) / 256, ITW = (kTN * CPR) / 256;
  static_assert((kRows * CPR) % 256 == 0 && (kTN * CPR) % 256 == 0, "staging must divide evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char lin_smem[];
  S* As = reinterpret_cast<S*>(lin_smem);  // [kRows][kLd]
  S* Ws = As + kRows * kLd;                // [64][kLd]
  const S* A = reinterpret_cast<const S*>(A_);
  const S* W = reinterpret_cast<const S*>(W_);
  const int n0 = blockIdx.x * kTN, slice = blockIdx.y;
  const int steps = (K + kTK - 1) / kTK;
  const int s0 = (int)((int64_t)steps * slice / n_slices), s1 = (int)((int64_t)steps * (slice + 1) / n_slices);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w >> 1, wm = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  f32x4_t acc[2][MTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 a4[ITA], w4[ITW];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
      const int idx = it * 256 + tid;
      const int r = idx / CPR, ch = (idx % CPR) * 8;
      a4[it] = make_uint4(0, 0, 0, 0);
      if (r < M && k0 + ch < K) a4[it] = *reinterpret_cast<const uint4*>(A + (int64_t)r * lda + k0 + ch);
    }
#pragma unroll
    for (int it = 0; it < ITW; ++it) {
      const int idx = it * 256 + tid;
      const int r = idx / CPR, ch = (idx % CPR) * 8;
      w4[it] = make_uint4(0, 0, 0, 0);
      if (n0 + r < N && k0 + ch < K) w4[it] = lin_ldg_nt(W + (int64_t)(n0 + r) * K + k0 + ch);
    }
  };
  if (s0 < s1) fetch(s0 * kTK);
  for (int st = s0; st < s1; ++st) {
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
      const int idx = it * 256 + tid;
      *reinterpret_cast<uint4*>(As + (idx / CPR) * kLd + (idx % CPR) * 8) = a4[it];
    }
#pragma unroll
    for (int it = 0; it < ITW; ++it) {
      const int idx = it * 256 + tid;
      *reinterpret_cast<uint4*>(Ws + (idx / CPR) * kLd + (idx % CPR) * 8) = w4[it];
    }
    __syncthreads();
    if (st + 1 < s1) fetch((st + 1) * kTK);
#pragma unroll
    for (int ks = 0; ks < kTK / 32; ++ks) {
      uint4 wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const uint4*>(Ws + (wn * 32 + i * 16 + lr) * kLd + ks * 32 + lg * 8);
#pragma unroll
      for (int j = 0; j < MTW; ++j) {
        const uint4 xf = *reinterpret_cast<const uint4*>(As + (wm * 16 * MTW + j * 16 + lr) * kLd + ks * 32 + lg * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = mfma16_lin<T>(wf[i], xf, acc[i][j]);  // D[n][m]
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const int n = n0 + wn * 32 + i * 16 + lg * 4;
      const int m = wm * 16 * MTW + j * 16 + lr;
      if (m < M && n < N) {
        if (part) {
          *reinterpret_cast<float4*>(part + ((int64_t)slice * M + m) * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {  // single slice: the rounded result directly (8-byte store)
          uint2 o;
          o.x = (uint32_t)Elem<T>::from_f(acc[i][j][0]) | ((uint32_t)Elem<T>::from_f(acc[i][j][1]) << 16);
          o.y = (uint32_t)Elem<T>::from_f(acc[i][j][2]) | ((uint32_t)Elem<T>::from_f(acc[i][j][3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<S*>(C_) + (int64_t)m * ldc + n) = o;
        }
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

extern "C" int dl_linear_splitk(const void* A, int64_t lda, const void* W, float* parts, int M, int N, int K, int n_slices, int dtype, void* stream) {
  DL_REQUIRE(A && W && parts, "dl_linear_splitk: NULL pointer");
  DL_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0 && N % 4 == 0, "dl_linear_splitk: bad shape M=%d N=%d K=%d", M, N, K);
  DL_REQUIRE(n_slices >= 1 && n_slices <= 64 && n_slices <= (K + 127) / 128, "dl_linear_splitk: n_slices=%d out of range (1..min(64, K/128))", n_slices);
  DL_REQUIRE(dtype == DL_F16 || dtype == DL_BF16, "dl_linear_splitk: bf16 / f16 only");
  hipStream_t st = as_stream(stream);
  // tools/bench_linear_splitk.py: up to 192 rows the all-rows tile with 64-wide K slabs wins where the library is weakest -- down_proj
  // [4096,11008] at M=170, 8 slices: 35.7 us against 48.6 (o_proj ties, q|k|v and gate|up lose: every 64-column workgroup re-reads the
  // whole X slab from L2, N/64 x |X| in total, which is what bounds these launches) -- beyond that the 64x64 tiling.
  if (M <= 192) {
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)n_slices);
#define DL_WIDE(TT, MTWV)                                                                                                        \
  {                                                                                                                              \
    auto kfn = linear_splitk_wide_kernel<TT, 64, MTWV>;                                                                          \
    const size_t smem = (size_t)(32 * MTWV + 64) * (64 + 8) * 2;                                                                 \
    hipLaunchKernelGGL(kfn, grid, dim3(256), smem, st, A, lda, W, parts, (void*)nullptr, (int64_t)0, M, N, K, n_slices);          \
  }
    if (dtype == DL_BF16) {
      if (M <= 128) DL_WIDE(bf16_t, 4) else DL_WIDE(bf16_t, 6)
    } else {
      if (M <= 128) DL_WIDE(f16_t, 4) else DL_WIDE(f16_t, 6)
    }
#undef DL_WIDE
    DL_CHECK_LAUNCH("dl_linear_splitk");
    return DL_OK;
  }
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)n_slices);
  if (dtype == DL_BF16) hipLaunchKernelGGL((linear_splitk_kernel<bf16_t, 128>), grid, dim3(256), 0, st, A, lda, W, parts, M, N, K, n_slices);
  else hipLaunchKernelGGL((linear_splitk_kernel<f16_t, 128>), grid, dim3(256), 0, st, A, lda, W, parts, M, N, K, n_slices);
  DL_CHECK_LAUNCH("dl_linear_splitk");
  return DL_OK;
}

extern "C" int dl_linear(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, const void* R, int64_t ldr,
                         int M, int N, int K, int flags, int dtype, void* stream) {
  DL_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0, "dl_linear: bad shape M=%d N=%d K=%d lda=%lld", M, N, K, (long long)lda);
  DL_REQUIRE(dtype == DL_F32 || dtype == DL_F16 || dtype == DL_BF16, "dl_linear: unsupported dtype %d", dtype);
  if (M == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(A && W && C, "dl_linear: NULL pointer");
  DL_REQUIRE(!(flags & DL_EPI_RESIDUAL) || R, "dl_linear: residual requested but R is NULL");
  linear_launch(A, lda, W, bias, C, ldc, R, ldr, M, N, K, flags, dtype, as_stream(stream));
  DL_CHECK_LAUNCH("dl_linear");
  return DL_OK;
}
