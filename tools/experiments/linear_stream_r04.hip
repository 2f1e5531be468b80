// dl_linear_stream: Y[M <= 192, N] = X[M, K] W[N, K]^T for the decoder GEMMs of the post-compaction prefill layers (DML:1011-1013 q|k|v,
// DML:1127 o_proj, DML:328 gate / up / down at M = N' = 117..192 packed rows: 30 of the 32 layers of a B=1 request).
//
// At these row counts the GEMM is a WEIGHT STREAM with a matrix-core consumer: 404.8 MB of weights per layer against 68.8 GFLOP (27 us of
// MFMA at peak, 64 us of HBM at 6.3 TB/s).  What bounded the round-2/3 kernels (linear_splitk_wide_kernel, the library at these shapes) was
// not the X re-reads from L2 but the bytes of W in flight: one 8 KiB slab per workgroup per ~2 us memory round trip is 2 TB/s.  Here every
// wave keeps D 64-wide K slabs of ITS OWN 32 weight rows in flight in registers (non-temporal 16-byte loads in MFMA operand layout: the
// weights never touch LDS), X (L2-resident, <= 1.5 MB per K = 4096) is staged two slabs ahead through double-buffered LDS shared by the
// workgroup's waves (one barrier per slab), and all M rows sit in one tile: every weight byte is read exactly once.
//
//   tile     : all rows (128 or 192) x 32 NW columns, a wave owns 32 columns x all rows
//   split-K  : n_slices > 1 -> fp32 partial tiles [slice][M][N], added in slice order by the consumer (dl_add_rmsnorm_parts): deterministic
//   n_slices = 1 -> the rounded result directly; DL_STREAM_SILU_PAIR: W = [gate; up] ([2 I, K]) and a wave's columns are 16 gate
//              neurons + their 16 up neurons, so the epilogue writes cast(cast(silu(g)) * u) -> [M, I] (DML:328; the separate SiLU*up
//              launch and its [M, 2 I] round trip disappear)
#include <mutex>
#include <type_traits>

#include "dl_common.h"
#include "dynllava.h"
#include "experiments_abi.h"

namespace dl {

typedef __bf16 ls_bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 ls_f16x8_t __attribute__((ext_vector_type(8)));
typedef float ls_f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ ls_f32x4_t ls_mfma(const uint4& a, const uint4& b, ls_f32x4_t c);
template <>
__device__ __forceinline__ ls_f32x4_t ls_mfma<bf16_t>(const uint4& a, const uint4& b, ls_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ls_bf16x8_t, a), __builtin_bit_cast(ls_bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ ls_f32x4_t ls_mfma<f16_t>(const uint4& a, const uint4& b, ls_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ls_f16x8_t, a), __builtin_bit_cast(ls_f16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ uint4 ls_ldg_nt(const void* p) {
  typedef uint32_t ls_u32x4_t __attribute__((ext_vector_type(4)));
  const ls_u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const ls_u32x4_t*>(p));
  return make_uint4(r.x, r.y, r.z, r.w);
}

// compile-time loop: the register rings below must be indexed by CONSTANTS (a `#pragma unroll` loop that the optimiser declines to unroll
// turns them into scratch memory)
template <int I, int N, typename F>
__device__ __forceinline__ void ls_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ls_static_for<I + 1, N>(f);
  }
}

// (a `uint4 = *ptr` struct assignment becomes a memcpy into private memory that keeps the whole ring out of registers: load as a vector)
__device__ __forceinline__ uint4 ls_ldg(const void* p) {
  typedef uint32_t ls_u32x4_t __attribute__((ext_vector_type(4)));
  const ls_u32x4_t r = *reinterpret_cast<const ls_u32x4_t*>(p);
  return make_uint4(r.x, r.y, r.z, r.w);
}

constexpr int kLsTK = 64;          // K slab per step
constexpr int kLsLd = kLsTK + 8;   // padded LDS row (elements): 144 bytes, 16-byte aligned, conflict-free for the 16-byte fragment reads

// NW waves, each owning 32 output columns (two 16-column MFMA tiles) x ALL rows (16 MT: 128 or 192); D weight slabs in flight per wave.
//   W : global -> registers, already in MFMA operand layout (lane (lr, lg) = row n0 + lr, 8 consecutive k at lg * 8: a wave instruction reads
//       16 rows x 64 contiguous bytes, the two k-halves of a slab back to back so that every 128-byte line is fetched once); a wave's columns
//       are its own, so no weight byte ever passes through LDS or is read twice
//   X : global (L2) -> registers -> double-buffered LDS, shared by the NW waves; one barrier per slab
//   per 32 k: 2 W fragments (registers) x MT X fragments (LDS) -> 2 MT MFMAs: 0.5 KiB of LDS reads per MFMA (the first version of this
//       kernel -- wave tile 32 x 96, W through LDS too -- needed 0.9 and was LDS-bound at a third of the MFMA rate)
template <typename T, int NW, int MT, int D, bool PAIR>
__global__ __launch_bounds__(NW * 64) void linear_stream_kernel(const void* __restrict__ A_, int64_t lda, const void* __restrict__ W_, float* __restrict__ part,
                                                                 void* __restrict__ C_, int64_t ldc, int M, int N, int K, int n_slices) {
  using S = uint16_t;
  constexpr int kThreads = NW * 64;
  constexpr int kRows = 16 * MT;
  constexpr int CPR = kLsTK / 8;                                   // 16-byte chunks per slab row
  constexpr int kChunks = kRows * CPR;
  constexpr int ITA = (kChunks + kThreads - 1) / kThreads;         // X chunks per lane per slab
  constexpr int DX = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char ls_smem[];
  constexpr int kBufElems = kRows * kLsLd;
  S* lds = reinterpret_cast<S*>(ls_smem);  // [2][kRows][kLsLd]
  const S* A = reinterpret_cast<const S*>(A_);
  const S* W = reinterpret_cast<const S*>(W_);
  const int slice = blockIdx.y;
  const int steps = K / kLsTK;
  const int s0 = (int)((int64_t)steps * slice / n_slices), s1 = (int)((int64_t)steps * (slice + 1) / n_slices);
  const int ns = s1 - s0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // PAIR: N = 2 I rows of [gate; up]; the wave's first tile = 16 gate neurons, its second tile = the matching 16 up neurons, so a lane holds g and u
  // of the same (row, neuron) in acc[0][j] / acc[1][j]
  const int n_out = PAIR ? N / 2 : N;
  const int per_wave = PAIR ? 16 : 32;
  const int nb = (blockIdx.x * NW + w) * per_wave;  // first output neuron of this wave

  const S* wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int n;
    if constexpr (PAIR) {
      const int neuron = nb + lr;
      n = i * n_out + (neuron < n_out ? neuron : n_out - 1);
    } else {
      n = nb + i * 16 + lr;
      n = n < N ? n : N - 1;
    }
    wsrc[i] = W + (int64_t)n * K + lg * 8;
  }
  const S* asrc[ITA];
  int adst[ITA];
#pragma unroll
  for (int it = 0; it < ITA; ++it) {
    int idx = it * kThreads + tid;
    idx = idx < kChunks ? idx : kChunks - 1;  // (a partial last round re-stages the last chunk: same value, same place)
    const int r = idx / CPR, ch = (idx % CPR) * 8;
    asrc[it] = A + (int64_t)(r < M ? r : M - 1) * lda + ch;  // rows past M compute garbage that is never stored
    adst[it] = r * kLsLd + ch;
  }

  ls_f32x4_t acc[2][MT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = ls_f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 wr[D][2][2], xr[DX][ITA];  // wr[slab][k half][tile]
  // every prefetch is unconditional (a conditional load collapses hipcc's counted vmcnt waits to full drains): steps past the end re-read
  // the last slab, and nothing consumes them
  auto kof = [&](int st) { return (int64_t)(s0 + (st < ns ? st : ns - 1)) * kLsTK; };
#pragma unroll
  for (int d = 0; d < DX; ++d) {
    const int64_t k0 = kof(d);
#pragma unroll
    for (int it = 0; it < ITA; ++it) xr[d][it] = ls_ldg(asrc[it] + k0);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int64_t k0 = kof(d);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) wr[d][ks][i] = ls_ldg_nt(wsrc[i] + k0 + ks * 32);
  }

  static_assert(D % DX == 0, "the X ring must divide the W ring");
  // Software pipeline (one wave per SIMD: nothing else hides the LDS latency).  The X fragments of the NEXT 32 k are requested from LDS while
  // the MFMAs of the current 32 k run: inside a slab from the same buffer, across slabs right after the barrier that publishes the next buffer.
  //   iteration st:  stage slab st+1 into the other buffer; issue the global prefetches;
  //                  [reads k-half 1 of slab st || MFMAs k-half 0];  barrier;  [reads k-half 0 of slab st+1 || MFMAs k-half 1]
  auto xfrag = [&](const S* bufp, int ks, int j) { return *reinterpret_cast<const uint4*>(bufp + (j * 16 + lr) * kLsLd + ks * 32 + lg * 8); };
  if (ns > 0) {
#pragma unroll
    for (int it = 0; it < ITA; ++it) *reinterpret_cast<uint4*>(lds + adst[it]) = xr[0][it];
    const int64_t kx = kof(DX);
#pragma unroll
    for (int it = 0; it < ITA; ++it) xr[0][it] = ls_ldg(asrc[it] + kx);
  }
  __syncthreads();
  uint4 xa[MT], xb[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) xa[j] = xfrag(lds, 0, j);
  int buf = 0;
  for (int base = 0; base < ns; base += D) {
    ls_static_for<0, D>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int st = base + u;
      constexpr int dw = u, dxn = (u + 1) % DX;  // ring slot of slab st + 1 (slab 0 used slot 0 in the prologue)
      const S* cur = lds + buf * kBufElems;
      S* nxt = lds + (buf ^ 1) * kBufElems;
      uint4 wf[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[ks][i] = wr[dw][ks][i];
      if (st + 1 < ns) {
#pragma unroll
        for (int it = 0; it < ITA; ++it) *reinterpret_cast<uint4*>(nxt + adst[it]) = xr[dxn][it];
      }
      {
        const int64_t kx = kof(st + 1 + DX), kw = kof(st + D);
#pragma unroll
        for (int it = 0; it < ITA; ++it) xr[dxn][it] = ls_ldg(asrc[it] + kx);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) wr[dw][ks][i] = ls_ldg_nt(wsrc[i] + kw + ks * 32);
      }
      if (st < ns) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          xb[j] = xfrag(cur, 1, j);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = ls_mfma<T>(wf[0][i], xa[j], acc[i][j]);  // D[n][m]: row (n) = lg*4 + r, col (m) = lr
        }
      }
      __syncthreads();
      if (st < ns) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          xa[j] = xfrag(nxt, 0, j);  // slab st + 1 (garbage after the last slab: never consumed)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = ls_mfma<T>(wf[1][i], xb[j], acc[i][j]);
        }
      }
      buf ^= 1;
    });
  }

  // ---- epilogue ----
  if constexpr (PAIR) {
    // acc[0][j][r] = gate, acc[1][j][r] = up of neuron nb + lg*4 + r, row j*16 + lr
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = j * 16 + lr;
      const int n = nb + lg * 4;
      if (m < M && n < n_out) {
        uint32_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = Elem<T>::round(acc[0][j][r]), u = Elem<T>::round(acc[1][j][r]);
          o[r] = (uint32_t)Elem<T>::from_f(Elem<T>::round(g / (1.0f + expf(-g))) * u);
        }
        S* dst = reinterpret_cast<S*>(C_) + (int64_t)m * ldc + n;
        if (n + 3 < n_out) {
          *reinterpret_cast<uint2*>(dst) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
        } else {
          for (int r = 0; r < 4 && n + r < n_out; ++r) dst[r] = (S)o[r];
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int n = nb + i * 16 + lg * 4;
        const int m = j * 16 + lr;
        if (m < M && n < N) {  // N % 4 == 0 (host): the 4 columns of a lane are all inside
          if (part) {
            *reinterpret_cast<float4*>(part + ((int64_t)slice * M + m) * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          } else {
            uint2 o;
            o.x = (uint32_t)Elem<T>::from_f(acc[i][j][0]) | ((uint32_t)Elem<T>::from_f(acc[i][j][1]) << 16);
            o.y = (uint32_t)Elem<T>::from_f(acc[i][j][2]) | ((uint32_t)Elem<T>::from_f(acc[i][j][3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<S*>(C_) + (int64_t)m * ldc + n) = o;
          }
        }
      }
  }
}

template <typename T, int NW, int MT, int D, bool PAIR>
static int ls_go(const void* A, int64_t lda, const void* W, float* parts, void* C, int64_t ldc, int M, int N, int K, int n_slices, hipStream_t st) {
  auto kfn = linear_stream_kernel<T, NW, MT, D, PAIR>;
  const size_t smem = (size_t)2 * (16 * MT) * kLsLd * 2;
  const int n_out = PAIR ? N / 2 : N;
  const int per = NW * (PAIR ? 16 : 32);
  const dim3 grid((unsigned)((n_out + per - 1) / per), (unsigned)n_slices);
  hipLaunchKernelGGL(kfn, grid, dim3(NW * 64), smem, st, A, lda, W, parts, C, ldc, M, N, K, n_slices);
  return DL_OK;
}

}  // namespace dl

using namespace dl;

extern "C" int dl_linear_stream(const void* A, int64_t lda, const void* W, void* out, int64_t ldc, int M, int N, int K, int n_slices, int flags, int variant,
                                int dtype, void* stream) {
  DL_REQUIRE(A && W && out, "dl_linear_stream: NULL pointer");
  DL_REQUIRE(M > 0 && M <= 192 && N > 0 && N % 4 == 0 && K > 0 && K % kLsTK == 0 && lda % 8 == 0, "dl_linear_stream: bad shape M=%d N=%d K=%d (M <= 192, K %% 64 == 0)", M, N, K);
  DL_REQUIRE(n_slices >= 1 && n_slices <= 64 && n_slices <= K / kLsTK, "dl_linear_stream: n_slices=%d out of range", n_slices);
  DL_REQUIRE(dtype == DL_F16 || dtype == DL_BF16, "dl_linear_stream: bf16 / f16 only");
  const bool pair = (flags & DL_STREAM_SILU_PAIR) != 0;
  DL_REQUIRE(!pair || (n_slices == 1 && N % 8 == 0), "dl_linear_stream: SILU_PAIR needs n_slices == 1 and N % 8 == 0");
  DL_REQUIRE(n_slices > 1 || (ldc % 4 == 0 && ldc >= (pair ? N / 2 : N)), "dl_linear_stream: bad ldc");
  DL_REQUIRE(variant >= 0 && variant <= 3, "dl_linear_stream: variant must be 0..3");
  hipStream_t st = as_stream(stream);
  float* parts = n_slices > 1 ? reinterpret_cast<float*>(out) : nullptr;
  void* C = n_slices > 1 ? nullptr : out;
  int rc = DL_OK;
  // variant: waves per workgroup (32 columns each) x weight slabs in flight per wave.  0: 4 x 6, 1: 3 x 6, 2: 8 x 4, 3: 6 x 4
#define DL_LS(TT, NWV, MTV, DV)                                                                                             \
  rc = pair ? ls_go<TT, NWV, MTV, DV, true>(A, lda, W, parts, C, ldc, M, N, K, n_slices, st)                                \
            : ls_go<TT, NWV, MTV, DV, false>(A, lda, W, parts, C, ldc, M, N, K, n_slices, st)
#define DL_LS_M(TT, NWV, DV)                                                                                                \
  {                                                                                                                         \
    if (M <= 128) { DL_LS(TT, NWV, 8, DV); } else { DL_LS(TT, NWV, 12, DV); }                                                \
  }
#define DL_LS_V(TT)                                                                                                         \
  {                                                                                                                         \
    if (variant == 0) DL_LS_M(TT, 4, 6) else if (variant == 1) DL_LS_M(TT, 3, 6) else if (variant == 2) DL_LS_M(TT, 8, 4) else DL_LS_M(TT, 6, 4) \
  }
  if (dtype == DL_BF16) DL_LS_V(bf16_t) else DL_LS_V(f16_t)
#undef DL_LS_V
#undef DL_LS_M
#undef DL_LS
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_linear_stream");
  return DL_OK;
}
