// dl_gemm_smallm: Y[M,N] = X[M,K] @ W[N,K]^T for decode batches M <= 32 -- the nn.Linear calls of DML:1011-1013 (q/k/v_proj),
// DML:1127 (o_proj), DML:328 (gate/up/down_proj) and DML:2709 (lm_head) when 5..32 rows are decoded together.  Like dl_gemv this is
// pure weight streaming (M flop per weight byte), but the vector ALU cannot keep up beyond ~4 rows (4 v_dot2c per row and 16-byte
// chunk), so the products go to the matrix cores:
//
//  * X (one K slice of it) is RESIDENT in LDS for the whole kernel ([16*NB rows][Ks], padded rows): no staging pipeline, one
//    barrier.  K is split over gridDim.y slices when M*K*2 exceeds LDS or when N alone gives too few waves; partial sums go to an
//    fp32 workspace and are added in slice order by a second launch (deterministic).
//  * a WAVE owns 16 neurons at a time.  Weights go straight from HBM into the MFMA A-operand: lane l reads 16 bytes of row
//    (l & 15) at k-group (l >> 4), i.e. one instruction = 16 rows x 64 contiguous bytes, through an 8-deep register ring
//    (8 KiB per wave, 64 KiB per CU in flight).  B-operand = 16 batch rows x 32 k from LDS (conflict-free 16-byte reads).
//    v_mfma_f32_16x16x32: D[neuron][batch]; every lane ends up with 4 consecutive neurons of one batch row: stored directly, no
//    cross-lane reduction.  Two accumulators per batch tile break the MFMA dependency chain.
#include "dl_common.h"

namespace dl {

typedef __bf16 sm_bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 sm_f16x8_t __attribute__((ext_vector_type(8)));
typedef float sm_f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ sm_f32x4_t sm_mfma(const uint4& a, const uint4& b, sm_f32x4_t c);
template <>
__device__ __forceinline__ sm_f32x4_t sm_mfma<bf16_t>(const uint4& a, const uint4& b, sm_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sm_bf16x8_t, a), __builtin_bit_cast(sm_bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ sm_f32x4_t sm_mfma<f16_t>(const uint4& a, const uint4& b, sm_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sm_f16x8_t, a), __builtin_bit_cast(sm_f16x8_t, b), c, 0, 0, 0);
}

constexpr int kSmU = 8;          // weight ring depth: K steps of 32 in flight per wave
constexpr int kSmKUnit = 256;    // K slices are multiples of kSmU * 32
constexpr int kSmMaxM = 32;
constexpr int kSmLdsBytes = 150 * 1024;

// NB: batch tiles of 16 rows (M <= 16*NB).  NW: waves per workgroup.
template <typename T, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_smallm_kernel(const void* __restrict__ X_, int64_t ldx, const void* __restrict__ W_,
                                                               void* __restrict__ Y_, int64_t ldy, float* __restrict__ part, int M, int N,
                                                               int K, int n_slices, int direct) {
  using S = uint16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_smem[];
  S* xs = reinterpret_cast<S*>(sm_smem);
  const S* X = reinterpret_cast<const S*>(X_);
  const S* W = reinterpret_cast<const S*>(W_);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // this workgroup's K slice, in units of 256
  const int units = K / kSmKUnit;
  const int slice = blockIdx.y;
  const int u0 = (int)((int64_t)units * slice / n_slices), u1 = (int)((int64_t)units * (slice + 1) / n_slices);
  const int k0 = u0 * kSmKUnit, Ks = (u1 - u0) * kSmKUnit;
  const int ld = Ks + 8;  // LDS row stride (elements): +16 bytes -> rows shift by 4 banks
  const int steps = Ks / 32;

  // this wave's neuron tiles: t = first, first + stride, ...
  const int n_tiles = (N + 15) / 16;
  const int first = blockIdx.x * NW + wid, stride = gridDim.x * NW;
  const int my_tiles = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
  const int total = my_tiles * steps;  // linear (tile, step) stream of this wave

  auto wbase = [&](int tile_i) -> const S* {  // lane's weight row of this wave's tile_i-th tile, at the slice start
    int n = (first + tile_i * stride) * 16 + lr;
    n = n < N ? n : N - 1;
    return W + (int64_t)n * K + k0 + lg * 8;
  };
  auto wptr = [&](int t) -> const S* {  // weight fragment address of linear step t
    const int tile_i = t / steps;
    return wbase(tile_i) + (t - tile_i * steps) * 32;
  };

  // ---- start the weight stream before X is staged ----
  uint4 wr[kSmU];
#pragma unroll
  for (int u = 0; u < kSmU; ++u)
    if (u < total) wr[u] = *reinterpret_cast<const uint4*>(wptr(u));

  // ---- X slice -> LDS (rows >= M are zero) ----
  {
    const int chunks_per_row = Ks / 8;
    const int n_chunks = NB * 16 * chunks_per_row;
    for (int c = tid; c < n_chunks; c += NW * 64) {
      const int row = c / chunks_per_row, col = (c - row * chunks_per_row) * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row < M) v = *reinterpret_cast<const uint4*>(X + (int64_t)row * ldx + k0 + col);
      *reinterpret_cast<uint4*>(xs + row * ld + col) = v;
    }
  }
  __syncthreads();

  sm_f32x4_t acc[NB][2];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb][0] = acc[nb][1] = sm_f32x4_t{0.f, 0.f, 0.f, 0.f};

  const S* xb = xs + lr * ld + lg * 8;
  int st = 0, ti = 0;  // step within the tile, tile index
  for (int t0 = 0; t0 < total; t0 += kSmU) {  // steps is a multiple of kSmU: a ring round never straddles a tile
    const bool more = t0 + kSmU < total;  // the next round exists (whole rounds only)
    const int st_n = st + kSmU == steps ? 0 : st + kSmU;
    const S* wn = more ? wbase(st_n == 0 ? ti + 1 : ti) + st_n * 32 : nullptr;
#pragma unroll
    for (int u = 0; u < kSmU; ++u) {
      const uint4 a = wr[u];
      if (more) wr[u] = *reinterpret_cast<const uint4*>(wn + u * 32);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint4 b = *reinterpret_cast<const uint4*>(xb + nb * 16 * ld + (st + u) * 32);
        acc[nb][u & 1] = sm_mfma<T>(a, b, acc[nb][u & 1]);
      }
    }
    st += kSmU;
    if (st == steps) {  // tile finished: D[neuron = lg*4 + e][batch = lr]
      const int n = (first + ti * stride) * 16 + lg * 4;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int m = nb * 16 + lr;
        const sm_f32x4_t v = acc[nb][0] + acc[nb][1];
        if (m < M && n < N) {  // N % 4 == 0: a lane's 4 neurons are all inside or all outside
          if (direct) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store1<T>(Y_, (int64_t)m * ldy + n + e, v[e]);
          } else {  // one 16-byte store per lane: 16 rows x 64 contiguous bytes per instruction (was four 4-byte stores behind branches)
            *reinterpret_cast<float4*>(part + ((int64_t)slice * M + m) * N + n) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
        acc[nb][0] = acc[nb][1] = sm_f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      st = 0;
      ++ti;
    }
  }
}

// ---- variant 2: coalesced weight loads, transposed through wave-private LDS ----
// The direct-fragment loads above touch 16 rows x 64 B per instruction (half a cache line per row) and top out near 3.5-4 TB/s.
// Here a wave loads its [16 neurons x 256 k] chunk as dl_gemv does -- each instruction two rows x 512 contiguous bytes, non-temporal,
// two chunks (16 KiB) in flight per wave -- parks it in its own 8 KiB of LDS (no barrier: only the wave itself reads it back) and
// reads the MFMA A-fragments from there.  8 waves per workgroup; X slice resident as before.
constexpr int kSmStagePerWaveBytes(int kc) { return 16 * (kc + 8) * 2; }

__device__ __forceinline__ uint4 sm_ldg_nt(const void* p) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(r.x, r.y, r.z, r.w);
}

// KC: k per chunk (256: one instruction = 2 rows x 512 B; 128: 4 rows x 256 B).  NW: waves per workgroup (staging = NW x 16 x (KC+8) x 2 B).
template <typename T, int NB, int KC, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_smallm_staged_kernel(const void* __restrict__ X_, int64_t ldx,
                                                                             const void* __restrict__ W_, void* __restrict__ Y_, int64_t ldy,
                                                                             float* __restrict__ part, int M, int N, int K, int n_slices, int direct) {
  using S = uint16_t;
  constexpr int kSmKC = KC, kSmStLd = KC + 8;
  constexpr int LPR = KC / 8, RPI = 64 / LPR, NI = 16 / RPI;  // lanes per row, rows per load instruction, instructions per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_smem[];
  const S* X = reinterpret_cast<const S*>(X_);
  const S* W = reinterpret_cast<const S*>(W_);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int units = K / kSmKUnit;
  const int slice = blockIdx.y;
  const int u0 = (int)((int64_t)units * slice / n_slices), u1 = (int)((int64_t)units * (slice + 1) / n_slices);
  const int k0 = u0 * kSmKUnit, Ks = (u1 - u0) * kSmKUnit;
  const int ld = Ks + 8;
  const int cpt = Ks / kSmKC;  // chunks per tile
  S* stg = reinterpret_cast<S*>(sm_smem) + wid * 16 * kSmStLd;         // wave-private staging [16][kSmStLd]
  S* xs = reinterpret_cast<S*>(sm_smem) + NW * 16 * kSmStLd;           // X slice [NB*16][ld]

  const int n_tiles = (N + 15) / 16;
  const int first = blockIdx.x * NW + wid, stride = gridDim.x * NW;
  const int my_tiles = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
  const int total = my_tiles * cpt;

  const int srow = lane / LPR, scol = (lane % LPR) * 8;  // staging-load role of this lane: row RPI*j + srow, 16 bytes at scol
  auto issue = [&](int t, uint4(&r)[NI]) {
    const int tile_i = t / cpt, c = t - tile_i * cpt;
    const int n0 = (first + tile_i * stride) * 16;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int n = n0 + RPI * j + srow;
      n = n < N ? n : N - 1;
      r[j] = sm_ldg_nt(W + (int64_t)n * K + k0 + c * kSmKC + scol);
    }
  };

  uint4 ra[NI], rb[NI];
  if (total > 0) issue(0, ra);
  if (total > 1) issue(1, rb);

  {  // X slice -> LDS (rows >= M are zero)
    const int chunks_per_row = Ks / 8;
    const int n_chunks = NB * 16 * chunks_per_row;
    for (int c = tid; c < n_chunks; c += NW * 64) {
      const int row = c / chunks_per_row, col = (c - row * chunks_per_row) * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row < M) v = *reinterpret_cast<const uint4*>(X + (int64_t)row * ldx + k0 + col);
      *reinterpret_cast<uint4*>(xs + row * ld + col) = v;
    }
  }
  __syncthreads();

  sm_f32x4_t acc[NB][2];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb][0] = acc[nb][1] = sm_f32x4_t{0.f, 0.f, 0.f, 0.f};
  const S* xb = xs + lr * ld + lg * 8;
  const S* ab = stg + lr * kSmStLd + lg * 8;
  int ci = 0, ti = 0;  // chunk within the tile, tile index

  auto consume = [&](uint4(&r)[NI], int t_next) {
#pragma unroll
    for (int j = 0; j < NI; ++j) *reinterpret_cast<uint4*>(stg + (RPI * j + srow) * kSmStLd + scol) = r[j];
    if (t_next < total) issue(t_next, r);
#pragma unroll
    for (int ks = 0; ks < kSmKC / 32; ++ks) {
      const uint4 a = *reinterpret_cast<const uint4*>(ab + ks * 32);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint4 b = *reinterpret_cast<const uint4*>(xb + nb * 16 * ld + ci * kSmKC + ks * 32);
        acc[nb][ks & 1] = sm_mfma<T>(a, b, acc[nb][ks & 1]);
      }
    }
    if (++ci == cpt) {  // tile finished: D[neuron = lg*4 + e][batch = lr]
      const int n = (first + ti * stride) * 16 + lg * 4;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int m = nb * 16 + lr;
        const sm_f32x4_t v = acc[nb][0] + acc[nb][1];
        if (m < M && n < N) {  // N % 4 == 0: a lane's 4 neurons are all inside or all outside
          if (direct) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store1<T>(Y_, (int64_t)m * ldy + n + e, v[e]);
          } else {  // one 16-byte store per lane: 16 rows x 64 contiguous bytes per instruction (was four 4-byte stores behind branches)
            *reinterpret_cast<float4*>(part + ((int64_t)slice * M + m) * N + n) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
        acc[nb][0] = acc[nb][1] = sm_f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      ci = 0;
      ++ti;
    }
  };

  for (int t = 0; t < total; t += 2) {
    consume(ra, t + 2);
    if (t + 1 < total) consume(rb, t + 3);
  }
}

// Y[m,n] = cast(sum_s part[s,m,n]) in slice order.  N % 4 == 0.
template <typename T>
__global__ __launch_bounds__(256) void gemm_smallm_reduce_kernel(const float* __restrict__ part, int n_slices, int M, int N,
                                                                  void* __restrict__ Y_, int64_t ldy) {
  const int64_t nq = (int64_t)M * (N / 4);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nq; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / (N / 4);
    const int n = (int)(idx - m * (N / 4)) * 4;
    float4 s = *reinterpret_cast<const float4*>(part + m * N + n);
    for (int k0 = 1; k0 < n_slices; k0 += 8) {  // the loads of up to eight slices are in flight together; the additions stay in slice order
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(part + ((int64_t)(k0 + j < n_slices ? k0 + j : k0) * M + m) * N + n);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < n_slices) {
          s.x += v[j].x;
          s.y += v[j].y;
          s.z += v[j].z;
          s.w += v[j].w;
        }
    }
    store1<T>(Y_, m * ldy + n, s.x);
    store1<T>(Y_, m * ldy + n + 1, s.y);
    store1<T>(Y_, m * ldy + n + 2, s.z);
    store1<T>(Y_, m * ldy + n + 3, s.w);
  }
}

// slices: enough to (a) fit the X slice in LDS and (b) give the 256 CUs ~8 waves each.
static int sm_slices(int M, int N, int K, int want, int variant) {
  const int units = K / kSmKUnit;
  const int rows = M <= 16 ? 16 : 32;
  const int budget = kSmLdsBytes - (variant == 2 ? 8 * kSmStagePerWaveBytes(256) : (variant == 3 ? 16 * kSmStagePerWaveBytes(128) : 0));
  const int max_ks = (budget / (rows * 2) - 8) / kSmKUnit * kSmKUnit;
  int s_lds = 1;
  while ((units + s_lds - 1) / s_lds * kSmKUnit > max_ks) ++s_lds;
  int s = want;
  if (s <= 0) {
    const int n_tiles = (N + 15) / 16;
    // tools/bench_gemm_smallm.py: many short weight streams beat few long ones.  Direct fragments: qkv / o / down 8, gate|up 5;
    // LDS-staged (2048 waves in one round): gate|up / lm_head 4, qkv / o / down 8.
    s = variant >= 2 ? (n_tiles >= 1024 ? 4 : 8) : (6144 + n_tiles - 1) / n_tiles;
    if (s > 8) s = 8;
  }
  if (s < s_lds) s = s_lds;
  if (s > units) s = units;
  return s;
}

template <typename T, int NB, int NW>
static int sm_go(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, float* part, int M, int N, int K, int n_slices,
                 int direct, hipStream_t st) {
  const int units = K / kSmKUnit;
  const int max_ks = (units + n_slices - 1) / n_slices * kSmKUnit;
  const size_t smem = (size_t)NB * 16 * (max_ks + 8) * 2;
  auto kfn = gemm_smallm_kernel<T, NB, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      set_error("dl_gemm_smallm: cannot raise the dynamic LDS limit to 152 KiB");
      return DL_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int n_tiles = (N + 15) / 16;
  int gx = (n_tiles + NW - 1) / NW;
  const int cap = 8192 / NW / n_slices > 0 ? 8192 / NW / n_slices : 1;  // further tiles are looped over
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(kfn, dim3((unsigned)gx, (unsigned)n_slices), dim3(NW * 64), smem, st, X, ldx, W, Y, ldy, part, M, N, K, n_slices, direct);
  return DL_OK;
}

template <typename T>
static void sm_reduce(float* part, void* Y, int64_t ldy, int M, int N, int n_slices, hipStream_t st) {
  if (n_slices > 1) {
    const int64_t nq = (int64_t)M * (N / 4);
    const int64_t blocks = (nq + 255) / 256;
    hipLaunchKernelGGL((gemm_smallm_reduce_kernel<T>), dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, part, n_slices, M, N, Y,
                       ldy);
  }
}

template <typename T, int NB, int KC, int NW>
static int sm_go_staged(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, float* part, int M, int N, int K, int n_slices,
                        int direct, hipStream_t st) {
  const int units = K / kSmKUnit;
  const int max_ks = (units + n_slices - 1) / n_slices * kSmKUnit;
  const size_t smem = (size_t)NW * kSmStagePerWaveBytes(KC) + (size_t)NB * 16 * (max_ks + 8) * 2;
  auto kfn = gemm_smallm_staged_kernel<T, NB, KC, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      set_error("dl_gemm_smallm: cannot raise the dynamic LDS limit to 152 KiB");
      return DL_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int n_tiles = (N + 15) / 16;
  int gx = (n_tiles + NW - 1) / NW;
  const int cap = (256 * NW) / NW / n_slices > 0 ? (256 * NW) / NW / n_slices : 1;  // one workgroup per CU in a single round; tiles beyond that are looped over
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(kfn, dim3((unsigned)gx, (unsigned)n_slices), dim3(NW * 64), smem, st, X, ldx, W, Y, ldy, part, M, N, K, n_slices, direct);
  return DL_OK;
}

}  // namespace dl

using namespace dl;

extern "C" int dl_gemm_smallm_max_m(void) { return kSmMaxM; }

extern "C" int64_t dl_gemm_smallm_workspace_bytes(int M, int N, int K, int n_slices, int variant) {
  if (M <= 0 || N <= 0 || K <= 0 || K % kSmKUnit) return 0;
  const int s = sm_slices(M, N, K, n_slices, variant == 0 ? 2 : variant);
  return (int64_t)s * M * N * (int64_t)sizeof(float);  // also covers defer_reduce with a single slice
}

extern "C" int dl_gemm_smallm_slices(int M, int N, int K, int n_slices, int variant) {
  if (M <= 0 || N <= 0 || K <= 0 || K % kSmKUnit) return 0;
  return sm_slices(M, N, K, n_slices, variant == 0 ? 2 : variant);
}

extern "C" int dl_gemm_smallm(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, void* workspace, int M, int N, int K,
                              int n_slices, int wg_waves, int variant, int defer_reduce, int dtype, void* stream) {
  DL_REQUIRE(X && W && (Y || defer_reduce), "dl_gemm_smallm: NULL pointer");
  DL_REQUIRE(M > 0 && M <= kSmMaxM && N > 0 && K > 0, "dl_gemm_smallm: bad shape M=%d (max %d) N=%d K=%d", M, kSmMaxM, N, K);
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_gemm_smallm: bf16 / f16 only (MFMA path)");
  DL_REQUIRE(K % kSmKUnit == 0 && N % 4 == 0 && ldx % 8 == 0, "dl_gemm_smallm: K %% 256, N %% 4 and ldx %% 8 must be 0");
  DL_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0, "dl_gemm_smallm: X and W must be 16-byte aligned");
  DL_REQUIRE(n_slices >= 0 && n_slices <= 64, "dl_gemm_smallm: n_slices must be in [0, 64]");
  DL_REQUIRE(wg_waves == 0 || wg_waves == 4 || wg_waves == 8, "dl_gemm_smallm: wg_waves must be 0 (auto), 4 or 8");
  DL_REQUIRE(variant >= 0 && variant <= 3, "dl_gemm_smallm: variant must be 0 (auto), 1 (direct fragments), 2 (LDS-staged, 8 waves x 256 k) or 3 (16 waves x 128 k)");
  if (variant == 0) variant = 2;
  const int s = sm_slices(M, N, K, n_slices, variant);
  const int direct = (s == 1 && !defer_reduce) ? 1 : 0;
  DL_REQUIRE(direct || workspace, "dl_gemm_smallm: workspace required (dl_gemm_smallm_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  const bool w8 = wg_waves != 4;
  int rc = DL_OK;
#define DL_SM_ARGS X, ldx, W, Y, ldy, part, M, N, K, s, direct, st
  if (variant == 2) {
    if (dtype == DL_BF16) rc = M <= 16 ? sm_go_staged<bf16_t, 1, 256, 8>(DL_SM_ARGS) : sm_go_staged<bf16_t, 2, 256, 8>(DL_SM_ARGS);
    else rc = M <= 16 ? sm_go_staged<f16_t, 1, 256, 8>(DL_SM_ARGS) : sm_go_staged<f16_t, 2, 256, 8>(DL_SM_ARGS);
  } else if (variant == 3) {
    if (dtype == DL_BF16) rc = M <= 16 ? sm_go_staged<bf16_t, 1, 128, 16>(DL_SM_ARGS) : sm_go_staged<bf16_t, 2, 128, 16>(DL_SM_ARGS);
    else rc = M <= 16 ? sm_go_staged<f16_t, 1, 128, 16>(DL_SM_ARGS) : sm_go_staged<f16_t, 2, 128, 16>(DL_SM_ARGS);
  } else if (dtype == DL_BF16) {
    if (M <= 16) rc = w8 ? sm_go<bf16_t, 1, 8>(DL_SM_ARGS) : sm_go<bf16_t, 1, 4>(DL_SM_ARGS);
    else rc = w8 ? sm_go<bf16_t, 2, 8>(DL_SM_ARGS) : sm_go<bf16_t, 2, 4>(DL_SM_ARGS);
  } else {
    if (M <= 16) rc = w8 ? sm_go<f16_t, 1, 8>(DL_SM_ARGS) : sm_go<f16_t, 1, 4>(DL_SM_ARGS);
    else rc = w8 ? sm_go<f16_t, 2, 8>(DL_SM_ARGS) : sm_go<f16_t, 2, 4>(DL_SM_ARGS);
  }
#undef DL_SM_ARGS
  if (rc != DL_OK) return rc;
  if (!defer_reduce) {
    if (dtype == DL_BF16) sm_reduce<bf16_t>(part, Y, ldy, M, N, s, st); else sm_reduce<f16_t>(part, Y, ldy, M, N, s, st);
  }
  DL_CHECK_LAUNCH("dl_gemm_smallm");
  return DL_OK;
}
