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
#include <mutex>

#include "gemv_dot.h"
#include "granule.h"
#include "attn_decode_body.h"
#include "tp_body.h"
#include "../../include/dynllava.h"

namespace dl {

// tools/gemv_timing.hip compiles this file with -DDL_GEMV_TIMING to stamp the phases of workgroup 0 (100 MHz wall clock).
#ifdef DL_GEMV_TIMING
__device__ long long g_gemv_stamps[8];
#define DL_GSTAMP(i)                                                                                \
  do {                                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_gemv_stamps[i] = wall_clock64();                     \
  } while (0)
#else
#define DL_GSTAMP(i)
#endif

constexpr int kGemvThreads = 256;
constexpr int kGemvR = 2;       // neurons per wave per pass
constexpr int kGemvMaxB = 8;

// MODE: prologue (0 plain, 1 add+rmsnorm, 2 silu*up).  PAIR: the wave's R=2 neurons are (n, n + N/2) and the output is
// cast(cast(silu(y_n)) * y_{n+N/2}) -> y [B, N/2]  (gate|up fused weight: DML:328 computed in the epilogue).
// R neurons per wave per pass, U 16-byte chunks per neuron in flight  =>  R*U independent loads per lane.
// the stored value's bit pattern (what store1<T> writes), for the in-launch granules
template <typename T>
__device__ __forceinline__ uint32_t gemv_bits(float f) {
  if constexpr (Elem<T>::kBytes == 4) return __float_as_uint(f);
  else return (uint32_t)Elem<T>::from_f(f);
}

// The body is a device function so that dl_gemv_qkv_attn can run it as part of a wider grid: `bid` / `nblk` are this workgroup's index and the
// number of workgroups that share the rows.  gran != nullptr: every output is ALSO published as an 8-byte {gtag, value bits} granule
// (granule.h) for consumers inside the same launch.
template <typename T, int B, int MODE, bool PAIR, int R, int U>
__device__ __forceinline__ void gemv_body(const void* __restrict__ W_, int N, int K, const void* x_, int64_t x_rs, const void* __restrict__ h_,
                                          void* __restrict__ h_out_, const void* __restrict__ delta_, const void* __restrict__ nw_, float eps,
                                          void* __restrict__ y_, int64_t y_rs, const int bid, const int nblk, u64_t* gran, uint32_t gtag) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* xs = reinterpret_cast<S*>(smem);  // [B][K] in the model dtype
  __shared__ float red[4];
  DL_GSTAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nvec = K / V;

  // ---- start the weight stream BEFORE the prologue: the first R*U chunks of this workgroup's first neurons are in
  // flight while x is being built (a dependent HBM round trip costs ~1.5 us here; the prologue has two of them) ----
  const S* W = reinterpret_cast<const S*>(W_);
  const int n_out = PAIR ? N / 2 : N;                 // neurons indexed by the wave
  constexpr int RW = PAIR ? 1 : R;                    // wave-owned output neurons per pass (PAIR: one act = two rows)
  const int groups = (n_out + 4 * RW - 1) / (4 * RW);
  uint4 pre[R][U];
  const bool have_pre = bid < groups && lane + 64 * (U - 1) < nvec;
  if (have_pre) {
    const int n0 = bid * 4 * RW + wid * RW;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int n = PAIR ? (n0 + r * n_out) : (n0 + r);
      n = n < N ? n : N - 1;
      if (PAIR && n0 >= n_out) n = r * n_out;
#pragma unroll
      for (int u = 0; u < U; ++u) pre[r][u] = ldg_nt(W + (int64_t)n * K + (lane + 64 * u) * V);
    }
  }

  // ---- prologue: build x in LDS ----
  if constexpr (MODE == 1) {  // ADDNORM
    const S* h = reinterpret_cast<const S*>(h_);
    S* h_out = reinterpret_cast<S*>(h_out_);
    const S* dl_ = reinterpret_cast<const S*>(delta_);
    const S* nw = reinterpret_cast<const S*>(nw_);
    constexpr int MAXC = 4;  // 16-byte chunks per thread held in registers: K <= 256 * 4 * kVec (8192 for the 16-bit dtypes)
    if (nvec <= kGemvThreads * MAXC) {
      // every global load of the prologue (residual row, delta row, norm weight) is requested before anything is used: the
      // prologue is then ONE L2 round trip + the reduction, instead of load -> reduce -> load
      uint4 wr[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int v = tid + c * kGemvThreads;
        if (v < nvec) wr[c] = *reinterpret_cast<const uint4*>(nw + v * V);
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        uint4 hr[MAXC], dr[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * kGemvThreads;
          if (v < nvec) {
            hr[c] = *reinterpret_cast<const uint4*>(h + (int64_t)b * K + v * V);
            if (dl_) dr[c] = *reinterpret_cast<const uint4*>(dl_ + (int64_t)b * K + v * V);
          }
        }
        float a[MAXC][V];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * kGemvThreads;
          if (v < nvec) {
            unpack16<T>(hr[c], a[c]);
            if (dl_) {
              float d[V];
              unpack16<T>(dr[c], d);
#pragma unroll
              for (int e = 0; e < V; ++e) a[c][e] = Elem<T>::round(a[c][e] + d[e]);
              // updated residual stream: written once, to a DIFFERENT buffer (other workgroups are still reading h_in)
              if (bid == 0) store16<T>(h_out + (int64_t)b * K + v * V, a[c]);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) ss += a[c][e] * a[c][e];
          }
        }
        const float rstd = rsqrtf(block_sum<4>(ss, red) / (float)K + eps);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * kGemvThreads;
          if (v < nvec) {
            float w[V];
            unpack16<T>(wr[c], w);
#pragma unroll
            for (int e = 0; e < V; ++e) a[c][e] = w[e] * Elem<T>::round(a[c][e] * rstd);
            store16<T>(xs + b * K + v * V, a[c]);
          }
        }
      }
    } else {
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
            if (bid == 0) store16<T>(h_out + (int64_t)b * K + v * V, a);
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
  DL_GSTAMP(1);  // x is in LDS

  // ---- stream the weights ----
  bool first = have_pre;
  for (int grp = bid; grp < groups; grp += nblk) {
    const int n0 = grp * 4 * RW + wid * RW;
    float acc[R][B];
    const S* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
      int n = PAIR ? (n0 + r * n_out) : (n0 + r);
      n = n < N ? n : N - 1;
      if (PAIR && n0 >= n_out) n = r * n_out;
      wp[r] = W + (int64_t)n * K;
    }
    int v = lane;
    for (; v + 64 * (U - 1) < nvec; v += 64 * U) {
      uint4 raw[R][U];
      if (first) {  // already in flight since before the prologue
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) raw[r][u] = pre[r][u];
        first = false;
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) raw[r][u] = ldg_nt(wp[r] + (v + 64 * u) * V);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const uint4 xv = *reinterpret_cast<const uint4*>(xs + b * K + (v + 64 * u) * V);
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][b] = dot16<T>(raw[r][u], xv, acc[r][b]);
        }
      }
    }
    for (; v < nvec; v += 64) {
      uint4 raw[R];
#pragma unroll
      for (int r = 0; r < R; ++r) raw[r] = ldg_nt(wp[r] + v * V);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xs + b * K + v * V);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][b] = dot16<T>(raw[r], xv, acc[r][b]);
      }
    }
    if (grp == bid) DL_GSTAMP(2);  // first neuron group streamed
#ifdef DL_QA_TIMING
    if (grp == bid && threadIdx.x == 0 && blockIdx.x < 1200) g_qa_stamps[blockIdx.x][2] = wall_clock64();
#endif
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);
    if (lane == 0) {
      if constexpr (PAIR) {
        if (n0 < n_out) {
#pragma unroll
          for (int b = 0; b < B; ++b) {
            const float g = Elem<T>::round(acc[0][b]), u = Elem<T>::round(acc[1][b]);
            store1<T>(y_, (int64_t)b * y_rs + n0, Elem<T>::round(g / (1.0f + expf(-g))) * u);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int b = 0; b < B; ++b)
            if (n0 + r < N) {
              store1<T>(y_, (int64_t)b * y_rs + n0 + r, acc[r][b]);
              if (gran) gr_store(gran + (int64_t)b * N + n0 + r, gtag, gemv_bits<T>(acc[r][b]));
            }
      }
    }
  }
  DL_GSTAMP(3);
}

template <typename T, int B, int MODE, bool PAIR, int R, int U>
__global__ __launch_bounds__(kGemvThreads) void gemv_kernel(const void* __restrict__ W_, int N, int K, const void* x_, int64_t x_rs,
                                                            const void* __restrict__ h_, void* __restrict__ h_out_,
                                                            const void* __restrict__ delta_, const void* __restrict__ nw_,
                                                            float eps, void* __restrict__ y_, int64_t y_rs) {
  gemv_body<T, B, MODE, PAIR, R, U>(W_, N, K, x_, x_rs, h_, h_out_, delta_, nw_, eps, y_, y_rs, (int)blockIdx.x, (int)gridDim.x, nullptr, 0u);
}

// ---- batch 1, plain prologue (o_proj, down_proj): x lives in REGISTERS ----
// Every row a wave processes needs the same x chunks (lane + 64 c), so each lane loads its XB*8 chunks of x once, together with the first
// weight chunks: no LDS staging, no barrier -- the prologue of the generic kernel (copy x to LDS, 3 us cold) disappears into the same
// round trip as the first weights.  One row per wave, all of its chunks requested before the first dot product.
template <typename T, int XB>
__global__ __launch_bounds__(kGemvThreads) void gemv_b1_plain_kernel(const void* __restrict__ W_, int N, int K, const void* __restrict__ x_,
                                                                     void* __restrict__ y_) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nvec = K / V;
  const S* W = reinterpret_cast<const S*>(W_);
  const S* x = reinterpret_cast<const S*>(x_);
  const int groups = (N + 3) / 4;
  uint4 xr[XB * 8];
#pragma unroll
  for (int c = 0; c < XB * 8; ++c) {
    const int v = lane + 64 * c;
    xr[c] = v < nvec ? *reinterpret_cast<const uint4*>(x + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
  }
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    int n = grp * 4 + wid;
    const bool live = n < N;
    n = live ? n : N - 1;
    const S* wp = W + (int64_t)n * K;
    uint4 w[XB * 8];
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) {
      const int v = lane + 64 * c;
      w[c] = v < nvec ? ldg_nt(wp + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) acc = dot16<T>(w[c], xr[c], acc);
    acc = wave_sum(acc);
    if (lane == 0 && live) store1<T>(y_, n, acc);
  }
}

// The same for long rows (8192 < K <= 16384: down_proj, K = 11008 / 13824): a row is shared by the two waves of a pair, each
// keeping ITS half of x in registers (C chunks per lane) and streaming its half of the row; the pair's partial sums meet in LDS and
// the even wave adds them low half first (fixed order).  Two rows per workgroup pass.
template <typename T, int C>
__global__ __launch_bounds__(kGemvThreads) void gemv_b1_plain_halves_kernel(const void* __restrict__ W_, int N, int K, const void* __restrict__ x_,
                                                                            void* __restrict__ y_) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ float part[kGemvThreads / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int half = wid & 1, pr = wid >> 1;
  const int nvec = K / V;
  const int hv = ((nvec + 1) / 2 + 63) / 64 * 64;  // chunks of the low half (whole lane sets)
  const int v0 = half * hv, v1 = half ? nvec : (hv < nvec ? hv : nvec);
  const S* W = reinterpret_cast<const S*>(W_);
  const S* x = reinterpret_cast<const S*>(x_);
  constexpr int RPW = kGemvThreads / 128;  // rows per workgroup pass
  const int groups = (N + RPW - 1) / RPW;
  uint4 xr[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int v = v0 + lane + 64 * c;
    xr[c] = v < v1 ? *reinterpret_cast<const uint4*>(x + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
  }
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    int n = grp * RPW + pr;
    const bool live = n < N;
    n = live ? n : N - 1;
    const S* wp = W + (int64_t)n * K;
    uint4 w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int v = v0 + lane + 64 * c;
      w[c] = v < v1 ? ldg_nt(wp + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) acc = dot16<T>(w[c], xr[c], acc);
    acc = wave_sum(acc);
    if (lane == 0 && half) part[wid] = acc;
    __syncthreads();
    if (lane == 0 && !half && live) store1<T>(y_, n, acc + part[wid + 1]);
    __syncthreads();
  }
}

// workgroup cap (per call: `grid_cap` of dl_gemv, 0 = this default; no process-global state)
// tools/bench_gemv.py sweep (after the prologue became one round trip): 4 workgroups per CU beat 2 on the add+norm shapes (qkv 19.6 ->
// 17.3 us, gate|up 31.6 -> 29.2 us, vocabulary projection 45.2 -> 40.9 us); o / down have only 512 neuron groups
constexpr int kGemvGridCap = 1024;

template <typename T, int B, int MODE, bool PAIR, int R, int U>
static int gemv_go(const void* W, int N, int K, const void* x, int64_t x_rs, const void* h, void* h_out, const void* delta,
                   const void* nw, float eps, void* y, int64_t y_rs, int grid_cap, hipStream_t st) {
  const size_t smem = (size_t)B * K * Elem<T>::kBytes;
  const int n_out = PAIR ? N / 2 : N;
  const int per = 4 * (PAIR ? 1 : R);
  const int groups = (n_out + per - 1) / per;
  // B >= 2: the extra workgroups only add x-staging and VALU pressure (B=2: 3.23 -> 3.49 ms/step with the B=1 cap): half the cap
  const int cap = B == 1 ? grid_cap : (grid_cap / 2 > 0 ? grid_cap / 2 : 1);
  const int grid = groups < cap ? groups : cap;
  auto kfn = gemv_kernel<T, B, MODE, PAIR, R, U>;
  if (smem > 64 * 1024) {
    static std::once_flag once;  // one per kernel instantiation; concurrent host threads are fine
    static bool attr_ok = false;
    std::call_once(once, [&] {
      attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) == hipSuccess;
      if (!attr_ok) (void)hipGetLastError();  // do not leave a sticky error behind
    });
    if (!attr_ok) {
      set_error("dl_gemv: cannot raise the dynamic LDS limit to 152 KiB");
      return DL_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kGemvThreads), smem, st, W, N, K, x, x_rs, h, h_out, delta, nw, eps, y, y_rs);
  return DL_OK;
}

template <typename T, int B, int MODE>
static int gemv_variant(bool pair, const void* W, int N, int K, const void* x, int64_t x_rs, const void* h, void* h_out, const void* delta,
                        const void* nw, float eps, void* y, int64_t y_rs, int grid_cap, hipStream_t st) {
#define DL_ARGS W, N, K, x, x_rs, h, h_out, delta, nw, eps, y, y_rs, grid_cap, st
  // one load schedule (2 neurons x 4 chunks per wave in flight): the tools/bench_gemv.py sweep over (4x2), (2x8), (1x8), (4x4) found
  // nothing faster on any decode shape, and every extra schedule costs 72 kernel instantiations of compile time
  if (pair) return gemv_go<T, B, MODE, true, 2, 4>(DL_ARGS);
  // batch 1, plain prologue (o_proj / down_proj), 16-bit dtypes: x in registers, no LDS / barrier (down 18.3 -> 17.1 us, decode step
  // 2.709 -> 2.675 ms).  Otherwise one row x 8 chunks per wave, which doubles the neuron groups of these 4096-row projections so that
  // they also reach 4 workgroups per CU (o 7.70 -> 7.42 us, down 19.0 -> 18.3 us).
  if constexpr (B == 1 && MODE == 0 && Elem<T>::kVec == 8) {
    if (K / 8 <= 64 * 16) {  // x fits the register file: 8 / 16 chunks per lane (K <= 8192)
      const int groups = (N + 3) / 4;
      const int cap = grid_cap / 2 > 0 ? grid_cap / 2 : 1;  // two rows per wave reuse the x registers
      const int grid = groups < cap ? groups : cap;
      if (K / 8 <= 512) hipLaunchKernelGGL((gemv_b1_plain_kernel<T, 1>), dim3((unsigned)grid), dim3(kGemvThreads), 0, st, W, N, K, x, y);
      else hipLaunchKernelGGL((gemv_b1_plain_kernel<T, 2>), dim3((unsigned)grid), dim3(kGemvThreads), 0, st, W, N, K, x, y);
      return DL_OK;
    }
    if (K / 8 <= 2 * 64 * 16) {
      // 8192 < K <= 16384 (down_proj: K = 11008 / 13824): a row per wave PAIR, each wave half of x in registers (12 / 16 chunks per lane), whole
      // passes over a grid of up to 5/4 of the cap.  tools/bench_gemv.py: 7B down 17.5 (one wave per row, 24 chunks) -> 16.7 us, 13B down (which
      // did not fit one wave's registers and ran the generic LDS kernel) -> 23.0 us = 6.16 TB/s (2560 row pairs = 2 passes of 1280 workgroups)
      const int groups = (N + 1) / 2;
      const int wide = grid_cap + grid_cap / 4;
      const int passes = (groups + wide - 1) / wide;
      const int grid = (groups + passes - 1) / passes;
      if (((K / 8 + 1) / 2 + 63) / 64 <= 12) hipLaunchKernelGGL((gemv_b1_plain_halves_kernel<T, 12>), dim3((unsigned)grid), dim3(kGemvThreads), 0, st, W, N, K, x, y);
      else hipLaunchKernelGGL((gemv_b1_plain_halves_kernel<T, 16>), dim3((unsigned)grid), dim3(kGemvThreads), 0, st, W, N, K, x, y);
      return DL_OK;
    }
  }
  if constexpr (B == 1 && MODE == 0) return gemv_go<T, B, MODE, false, 1, 8>(DL_ARGS);
  return gemv_go<T, B, MODE, false, 2, 4>(DL_ARGS);
#undef DL_ARGS
}

template <typename T, int B>
static int gemv_launch(int mode, const void* W, int N, int K, const void* x, int64_t x_rs, const void* h, void* h_out, const void* delta, const void* nw,
                       float eps, void* y, int64_t y_rs, int grid_cap, hipStream_t st) {
  const bool pair = (mode & DL_GEMV_OUT_SILU_PAIR) != 0;
  const int pro = mode & 3;
  if (pro == DL_GEMV_ADDNORM) return gemv_variant<T, B, 1>(pair, W, N, K, x, x_rs, h, h_out, delta, nw, eps, y, y_rs, grid_cap, st);
  if (pro == DL_GEMV_SILUMUL) return gemv_variant<T, B, 2>(pair, W, N, K, x, x_rs, h, h_out, delta, nw, eps, y, y_rs, grid_cap, st);
  return gemv_variant<T, B, 0>(pair, W, N, K, x, x_rs, h, h_out, delta, nw, eps, y, y_rs, grid_cap, st);
}

// ---- dl_gemv_qkv_attn: the q|k|v projection of a batch-1 decode layer and the attention that consumes it, in ONE launch ----
// The decode attention at batch 1 is a 9 us latency chain (launch, first K/V bytes, softmax trips, merge) that moves 3 MB: 11 % of the step.
// Here its workgroups (one per head) request their K/V rows as soon as they start -- those do not depend on the projection -- wait for
// THEIR head_dim q outputs (the projection's first third; the streaming workgroups publish every output as a granule next to the ordinary
// store), run scores / softmax / P.V over the slab keys while the weights still stream, and only then wait for the new token's k / v (the
// projection's last outputs): after the stream ends the attention has one key and the merge left.  25.4 vs 26.5 us per layer, decode 2.617 ->
// 2.564 ms/token.  Timeline of one launch (wall-clock stamps, T = 200): streaming workgroups start 0-0.6 us, the attention workgroups 4.4 us (they
// are the last blocks); q arrives 12.4 us; slab part done 18.4 us; the stream's last outputs (v) arrive 19.0 us; attention done 21.9 us.
// No producer ever waits, so the launch cannot deadlock; the consumers' wait is bounded and poisons the output (NaN + error word) on give-up.
// Both halves are the shared bodies (gemv_body, attn_decode_body.h): the results are bit-identical to dl_gemv + dl_attn_decode_rope.
// neurons per wave per pass x 16-byte chunks in flight per neuron of the fused launch's projection.  ONE neuron per wave: a pass of the grid then covers
// ~4 x 992 neurons, i.e. the q rows (the first third of the projection) are complete after the FIRST pass -- the attention workgroups can start on the
// slab keys at about a third of the stream instead of two thirds (tools/qa_timing.hip: with 2 x 4 the last head received its q at 16.3 us of an
// 18.2 us stream, and the slab keys' 5.4 us then ran past the stream's end)
#ifndef DL_QA_R
#define DL_QA_R 1
#endif
constexpr int kQaR = DL_QA_R, kQaU = 8 / DL_QA_R;

struct QkvAttnArgs {
  // projection
  const void* W; const void* h; void* h_out; const void* delta; const void* nw; void* y;
  int N, K; float eps;
  // attention
  const void* cos_tab; const void* sin_tab; const int32_t* pos_base; const int32_t* kv_len; void* k_slab; void* v_slab; void* out;
  int64_t stride_b, stride_h;
  int n_pos, T_cap, n_heads, n_kv_heads, call_tag;
  u64_t* gran; int32_t* err;
  int n_splits, chunk_keys;  // attention workgroups per head, keys per workgroup (the last one takes the rest)
};

constexpr int kQaMaxSplits = 4;

// four workgroups per CU or the grid (projection + attention workgroups) is not resident at once: hold the kernel to 128 VGPRs
// MULTI: several attention workgroups per head (run-time n_splits).  Two instantiations because the general form needs ~20 VGPRs more than the
// one-workgroup-per-head form (126), and the grid is only resident at once with four workgroups per CU = 128 VGPRs: the general form is held to
// that by the attribute (a few dwords of spill in the attention path, measured faster than three workgroups per CU by far).
template <typename T, int D, bool MULTI>
__global__ __launch_bounds__(kGemvThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemv_qkv_attn_kernel(QkvAttnArgs a_) {
  QkvAttnArgs a = a_;
  if constexpr (!MULTI) a.n_splits = 1;
  using S = typename Elem<T>::storage;
  const uint32_t tag = ((((uint32_t)a.pos_base[0] & 0x7fffffu) << 8) | ((uint32_t)a.call_tag & 0xffu)) + 1u;
  // the attention workgroups are the LAST blocks of the grid (and the grid stays within what the device holds at once): as first blocks they
  // displaced 32 streaming workgroups into a late second round (26.9 vs 25.4 us per launch)
  const int n_gemv = (int)gridDim.x - a.n_heads * a.n_splits;
  DL_QSTAMP(0);
  if ((int)blockIdx.x < n_gemv) {
    gemv_body<T, 1, 1, false, kQaR, kQaU>(a.W, a.N, a.K, nullptr, 0, a.h, a.h_out, a.delta, a.nw, a.eps, a.y, a.N, (int)blockIdx.x, n_gemv, a.gran, tag);
    DL_QSTAMP(1);
    return;
  }
  constexpr int NW = 4, U = 4;
  using St = AttnSplitState<T, D, NW, U>;
  constexpr int NG = St::NG;
  __shared__ float sm_m[NG], sm_l[NG];
  __shared__ float sm_o[NG * D];
  __shared__ __attribute__((aligned(16))) S rows[D];
  // round 4: `n_splits` workgroups per head, each with `chunk_keys` (= two register trips) of the head's slab keys -- the whole key range is then in
  // registers BEFORE q arrives, so what follows q is arithmetic + one exchange of partials instead of a chain of cold K/V round trips
  const int aw = (int)blockIdx.x - n_gemv, tid = threadIdx.x;
  const int h = aw / a.n_splits, split = aw % a.n_splits;
  const int n_rep = a.n_heads / a.n_kv_heads, kvh = h / n_rep;
  const int T_old = a.kv_len[0];
  St st;
  // (constant chunk sizes: with a run-time chunk the compiler keeps the non-speculative request path alive too, ~20 VGPRs the kernel does not have)
  if (a.n_splits == 1) attn_split_issue<T, D, NW, true, U>(st, tid, a.k_slab, a.v_slab, a.stride_b, a.stride_h, T_old, 1, 0, kvh, 0, 1, a.T_cap, 256);
  else attn_split_issue<T, D, NW, true, U>(st, tid, a.k_slab, a.v_slab, a.stride_b, a.stride_h, T_old, 1, 0, kvh, split, a.n_splits, a.T_cap, 128);
  attn_split_prefetch2<T, D, NW, U>(st);  // two trips in flight while the projection produces q
  AttnRopeRow<T> rope;  // the RoPE table row of the new token's position: requested now, not after q has arrived
  attn_newlast_preload<T, D, NW, U>(st, tid, a.cos_tab, a.sin_tab, a.n_pos, a.pos_base[0], rope);
  attn_split_pin_prefetched<T, D, NW, U>(st);  // the two trips of K/V really are in registers before the wait for q begins (this workgroup is idle until then anyway)
  // q first (the projection's first third): the slab keys need nothing else; k / v of the new token (its last third) only at the very end
  bool bad = false;
  auto value_of = [](u64_t v) -> S {
    if constexpr (Elem<T>::kBytes == 4) return __uint_as_float((uint32_t)v);
    else return (S)(uint32_t)v;
  };
  {  // rows[0, D) <- the q granules of this head; one lane watches the head's last neuron (produced last) with long naps first
    if (tid == 0) {
      for (int spins = 0;; ++spins) {
        if ((uint32_t)(gr_load(a.gran + h * D + D - 1) >> 32) == tag) break;
        if (spins > (1 << 20)) {
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(16);
      }
    }
    __syncthreads();
    for (int i = tid; i < D; i += kGemvThreads) {
      u64_t v = 0;
      for (int spins = 0;; ++spins) {
        v = gr_load(a.gran + h * D + i);
        if ((uint32_t)(v >> 32) == tag) break;
        if (spins > (1 << 20)) {
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      rows[i] = value_of(v);
    }
    __syncthreads();
    DL_QSTAMP(1);  // q has arrived
  }
  // the slab keys' partials are merged while the projection still streams; the new token is folded in by the D finishing threads, each of which
  // receives its three values (k[d], k[d +- D/2], v[d]: the projection's last outputs) straight from the granules: attn_split_finish_newlast
  __shared__ float red[NW];
  __shared__ __attribute__((aligned(16))) S q_rot_lds[D];
  float o_head;
  const u64_t* gk = a.gran + (int64_t)(a.n_heads + kvh) * D;
  const u64_t* gv = a.gran + (int64_t)(a.n_heads + a.n_kv_heads + kvh) * D;
  attn_split_finish_newlast<T, D, NW, U>(st, tid, rows, rope, q_rot_lds, 1.0f / sqrtf((float)D), h % n_rep == 0, a.T_cap, sm_m, sm_l,
                                         sm_o, red, o_head, [&](int d, int dpar, S& k_own, S& k_par, S& v_new) {
                                           u64_t g0 = 0, g1 = 0, g2 = 0;
                                           for (int spins = 0;; ++spins) {  // the three requests travel together; repeated until all carry this step's tag
                                             g0 = gr_load(gk + d);
                                             g1 = gr_load(gk + dpar);
                                             g2 = gr_load(gv + d);
                                             if ((uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag && (uint32_t)(g2 >> 32) == tag) break;
                                             if (spins > (1 << 20)) {
                                               bad = true;
                                               break;
                                             }
                                             __builtin_amdgcn_s_sleep(1);
                                           }
                                           k_own = value_of(g0);
                                           k_par = value_of(g1);
                                           v_new = value_of(g2);
                                         },
                                         [&](float& M, float& L, float& O) -> bool {
                                           if (a.n_splits == 1) return true;
                                           // partials of a head: granules [head][split - 1][M, L, O[D]] behind the projection's N granules
                                           constexpr int PG = D + 2;
                                           u64_t* pg = a.gran + a.N + ((int64_t)h * (kQaMaxSplits - 1)) * PG;
                                           if (split != 0) {
                                             if (tid < D) {
                                               u64_t* mine = pg + (int64_t)(split - 1) * PG;
                                               gr_store(mine + 2 + tid, tag, __float_as_uint(O));
                                               if (tid == 0) {
                                                 gr_store(mine, tag, __float_as_uint(M));
                                                 gr_store(mine + 1, tag, __float_as_uint(L));
                                               }
                                             }
                                             return false;
                                           }
                                           if (tid < D) {  // the primary: fold the other parts in, one after the other in split order (few live registers)
                                             for (int j = 0; j < a.n_splits - 1; ++j) {
                                               const u64_t* src = pg + (int64_t)j * PG;
                                               u64_t gm = 0, gl = 0, go = 0;
                                               for (int spins = 0;; ++spins) {  // the three requests travel together; repeated until all carry this step's tag
                                                 gm = gr_load(src);
                                                 gl = gr_load(src + 1);
                                                 go = gr_load(src + 2 + tid);
                                                 if ((uint32_t)(gm >> 32) == tag && (uint32_t)(gl >> 32) == tag && (uint32_t)(go >> 32) == tag) break;
                                                 if (spins > (1 << 20)) {
                                                   bad = true;
                                                   break;
                                                 }
                                                 __builtin_amdgcn_s_sleep(1);
                                               }
                                               const float Mj = __uint_as_float((uint32_t)gm), Lj = __uint_as_float((uint32_t)gl), Oj = __uint_as_float((uint32_t)go);
                                               const float Mt = fmaxf(M, Mj);
                                               if (Mt > -INFINITY) {
                                                 const float w0 = __expf(M - Mt), w1 = __expf(Mj - Mt);  // an empty part: exp(-inf) = 0
                                                 L = L * w0 + Lj * w1;
                                                 O = O * w0 + Oj * w1;
                                                 M = Mt;
                                               }
                                             }
                                           }
                                           return true;
                                         });
  if (split != 0) {  // (uniform per workgroup: the secondaries have published their part)
    if (__syncthreads_or(bad ? 1 : 0) && tid == 0 && a.err) atomicOr(a.err, 1);
    return;
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  DL_QSTAMP(3);
  if (tid < D) store1<T>(a.out, (int64_t)h * D + tid, any_bad ? __uint_as_float(0x7fc00000u) : o_head);
  if (any_bad && tid == 0 && a.err) atomicOr(a.err, 1);
}

// ---- dl_gemv_gu_tp: the gate|up projection of layer `sparse_layer` and the text predictor, in ONE launch ----
// The predictor (three small latency-bound launches, 22 us of the batch-1 step: LN + Linear(H -> d), Linear(d -> d/2), the d/2 -> d/4 -> 2 tail)
// reads the residual stream entering the layer -- the h_in of this very launch -- and only the end-of-step length advance needs its decision.
// Its stages run as the FIRST workgroups of the grid (resident from the start, gone after a few microseconds; stage k+1 receives stage k's
// outputs as granules), the projection streams next to them.  Shared stage bodies (tp_body.h): bit-identical to dl_text_predictor_decide.
struct GuTpArgs {
  const void* W; const void* h; void* h_out; const void* delta; const void* nw; void* y;
  int N, K; float eps;
  dl_tp_weights w;
  float* tp_ws; float* logits; int32_t* decision;
  const int32_t* pos_base;
  u64_t* gran; int32_t* err;
  int D, call_tag;
};

// MAXC: 16-byte chunks per lane of a stage-1 weight row held in registers (8: H <= 4096, 10: H <= 5120).  With 8 and two passes of stage 2b
// in flight the kernel needs 121 VGPRs -- four workgroups per CU like the plain projection (135 / 237 VGPRs held three / two).
template <typename T, int MAXC>
__global__ __launch_bounds__(kGemvThreads) void gemv_gu_tp_kernel(GuTpArgs a) {
  const int D = a.D, n1 = (D + 7) / 8, n2 = (D / 2 + 7) / 8, side = n1 + n2 + 1;
  const int bid = blockIdx.x, tid = threadIdx.x;
  if (bid >= side) {
    gemv_body<T, 1, 1, true, 2, 4>(a.W, a.N, a.K, nullptr, 0, a.h, a.h_out, a.delta, a.nw, a.eps, a.y, a.N / 2, bid - side, (int)gridDim.x - side, nullptr, 0u);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float gt_dyn[];  // >= max(H, 2 D) floats (host)
  const uint32_t tag = ((((uint32_t)a.pos_base[0] & 0x7fffffu) << 8) | ((uint32_t)a.call_tag & 0xffu)) + 1u;
  float* h1 = a.tp_ws;
  float* a1 = h1 + D;
  u64_t* g1 = a.gran;
  u64_t* g2 = a.gran + D;
  if (bid < n1) {  // stage 1: needs nothing from this launch
    tp_stage1_body<T, MAXC>(a.h, a.K, a.w.ln_w, a.w.ln_b, a.w.l1_w, a.w.l1_b, h1, a.K, D, bid, 0, g1, tag);
    return;
  }
  bool bad = false;
  auto fetch = [&](const u64_t* g, int n, float* dst) {  // dst[0, n) <- granules of the previous stage (produced almost together: watch the last one first)
    if (tid == 0) {
      for (int spins = 0;; ++spins) {
        if ((uint32_t)(gr_load(g + n - 1) >> 32) == tag) break;
        if (spins > (1 << 20)) {
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    // a thread's granules are requested TOGETHER and re-requested until all carry the tag (one at a time, each poll was a ~2 us round trip
    // beside the weight stream: 6 us from the last stage-1 store to stage 2a's first instruction)
    constexpr int GB = 4;
    for (int i0 = tid; i0 < n; i0 += GB * kGemvThreads) {
      u64_t v[GB];
      for (int spins = 0;; ++spins) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < GB; ++j) {
          const int i = i0 + j * kGemvThreads;
          v[j] = gr_load(g + (i < n ? i : i0));
          all = all && (uint32_t)(v[j] >> 32) == tag;
        }
        if (all) break;
        if (spins > (1 << 20)) {
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        const int i = i0 + j * kGemvThreads;
        if (i < n) dst[i] = __uint_as_float((uint32_t)v[j]);
      }
    }
  };
  if (bid < n1 + n2) {  // stage 2a
    fetch(g1, D, gt_dyn);
    if (__syncthreads_or(bad ? 1 : 0)) {
      if (tid == 0 && a.err) atomicOr(a.err, 2);
      return;  // its granules never appear: stage 2b gives up too and reports
    }
    tp_stage2a_body<T>(gt_dyn, a.w.l3_w, a.w.l3_b, a1, D, bid - n1, 0, g2, tag);
    return;
  }
  // stage 2b: its weights are requested first, then it waits for stage 2a's outputs (inside the body)
  float* a1s = gt_dyn + D;  // granule values; the body copies them into its own LDS area at gt_dyn
  tp_stage2b_body<T>(a1s, a.w.l5_w, a.w.l5_b, a.w.l7_w, a.w.l7_b, a.logits, a.decision, D, 0, gt_dyn, [&]() {
    fetch(g2, D / 2, a1s);
    bad = __syncthreads_or(bad ? 1 : 0) != 0;  // give-up: a1s holds whatever arrived; flag it, the decision below is then overwritten
    if (bad && tid == 0 && a.err) atomicOr(a.err, 2);
  });
  if (bad && tid == 0) a.decision[0] = 1;  // keep the token: the conservative outcome
}

}  // namespace dl

using namespace dl;

extern "C" int64_t dl_gemv_gu_tp_workspace_bytes(int d_model) { return d_model > 0 ? (int64_t)(d_model + d_model / 2) * (int64_t)sizeof(u64_t) : 0; }

extern "C" int dl_gemv_gu_tp(const void* W, int N, int K, const void* h_in, void* h_out, const void* delta, const void* norm_w, float eps, void* y,
                             const dl_tp_weights* tp, int d_model, void* tp_workspace, float* logits_out, int32_t* decision, const int32_t* pos_base,
                             void* granules, int call_tag, int32_t* err_flag, int dtype, int grid_cap, void* stream) {
  DL_REQUIRE(W && h_in && norm_w && y && tp && tp_workspace && decision && pos_base && granules, "dl_gemv_gu_tp: NULL pointer");
  DL_REQUIRE(N > 0 && N % 2 == 0 && K > 0 && d_model > 0 && d_model % 32 == 0 && K % 8 == 0 && K <= 5120, "dl_gemv_gu_tp: bad shape N=%d K=%d d_model=%d", N, K, d_model);
  DL_REQUIRE(!delta || (h_out && h_out != h_in), "dl_gemv_gu_tp: h_out must be a distinct buffer when delta is given");
  DL_REQUIRE(call_tag >= 0 && grid_cap >= 0, "dl_gemv_gu_tp: call_tag / grid_cap must be >= 0");
  if (grid_cap == 0) grid_cap = kGemvGridCap;
  hipStream_t st = as_stream(stream);
  int rc = DL_OK;
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(K % Elem<T>::kVec == 0, "dl_gemv_gu_tp: K must be a multiple of %d", Elem<T>::kVec);
    GuTpArgs a;
    a.W = W; a.h = h_in; a.h_out = h_out; a.delta = delta; a.nw = norm_w; a.y = y; a.N = N; a.K = K; a.eps = eps;
    a.w = *tp; a.tp_ws = reinterpret_cast<float*>(tp_workspace); a.logits = logits_out; a.decision = decision; a.pos_base = pos_base;
    a.gran = reinterpret_cast<u64_t*>(granules); a.err = err_flag; a.D = d_model; a.call_tag = call_tag;
    const int side = (d_model + 7) / 8 + (d_model / 2 + 7) / 8 + 1;
    const int groups = (N / 2 + 3) / 4;
    if (grid_cap > 2 * side) grid_cap -= side;  // projection + predictor workgroups together stay within what is resident at once (as dl_gemv_qkv_attn): with
                                                // 1024 + 97 the last 97 streaming workgroups ran as a second round behind the others (+9 us on the launch)
    const int grid = (groups < grid_cap ? groups : grid_cap) + side;
    size_t smem = (size_t)K * sizeof(float);  // stage 1 stages the row in fp32; the projection needs K elements of the model dtype
    if (smem < (size_t)2 * d_model * sizeof(float)) smem = (size_t)2 * d_model * sizeof(float);
    if (K / Elem<T>::kVec <= 64 * 8) hipLaunchKernelGGL((gemv_gu_tp_kernel<T, 8>), dim3((unsigned)grid), dim3(kGemvThreads), smem, st, a);
    else hipLaunchKernelGGL((gemv_gu_tp_kernel<T, 10>), dim3((unsigned)grid), dim3(kGemvThreads), smem, st, a);
  });
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_gemv_gu_tp");
  return DL_OK;
}

extern "C" int64_t dl_gemv_qkv_attn_workspace_bytes(int n_heads, int n_kv_heads, int head_dim) {
  // the projection's N granules + the partials of the secondary attention workgroups: [head][kQaMaxSplits - 1][M, L, O[head_dim]]
  return ((int64_t)(n_heads + 2 * n_kv_heads) * head_dim + (int64_t)n_heads * (kQaMaxSplits - 1) * (head_dim + 2)) * (int64_t)sizeof(u64_t);
}

extern "C" int dl_gemv_qkv_attn(const void* W, int K, const void* h_in, void* h_out, const void* delta, const void* norm_w, float eps, void* qkv,
                                const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* pos_base, const int32_t* kv_len, void* k_slab,
                                void* v_slab, int64_t slab_stride_b, int64_t slab_stride_h, int T_cap, void* out, void* granules, int call_tag,
                                int32_t* err_flag, int n_splits, int n_heads, int n_kv_heads, int head_dim, int dtype, int grid_cap, void* stream) {
  DL_REQUIRE(W && h_in && norm_w && qkv && cos_tab && sin_tab && pos_base && kv_len && k_slab && v_slab && out && granules, "dl_gemv_qkv_attn: NULL pointer");
  DL_REQUIRE(n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && (head_dim == 128 || head_dim == 64) && K > 0 && n_pos > 0 && T_cap > 0,
             "dl_gemv_qkv_attn: bad shape");
  DL_REQUIRE(!delta || (h_out && h_out != h_in), "dl_gemv_qkv_attn: h_out must be a distinct buffer when delta is given");
  DL_REQUIRE(call_tag >= 0 && grid_cap >= 0, "dl_gemv_qkv_attn: call_tag / grid_cap must be >= 0");
  DL_REQUIRE(n_splits >= 1 && n_splits <= kQaMaxSplits, "dl_gemv_qkv_attn: n_splits=%d must be in [1, %d]", n_splits, kQaMaxSplits);
  if (grid_cap == 0) grid_cap = kGemvGridCap;
  const int N = (n_heads + 2 * n_kv_heads) * head_dim;
  hipStream_t st = as_stream(stream);
  int rc = DL_OK;
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(K % Elem<T>::kVec == 0 && (size_t)K * Elem<T>::kBytes <= 48 * 1024, "dl_gemv_qkv_attn: K=%d unsupported", K);
    QkvAttnArgs a;
    a.W = W; a.h = h_in; a.h_out = h_out; a.delta = delta; a.nw = norm_w; a.y = qkv; a.N = N; a.K = K; a.eps = eps;
    a.cos_tab = cos_tab; a.sin_tab = sin_tab; a.pos_base = pos_base; a.kv_len = kv_len; a.k_slab = k_slab; a.v_slab = v_slab; a.out = out;
    a.stride_b = slab_stride_b; a.stride_h = slab_stride_h; a.n_pos = n_pos; a.T_cap = T_cap; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads;
    a.call_tag = call_tag; a.gran = reinterpret_cast<u64_t*>(granules); a.err = err_flag;

    // one workgroup per head: the whole row (speculative first request of 256 keys, as the stand-alone single-split launch); several: 128 keys
    // each = the two trips a workgroup holds in registers while it waits for q, the last one takes what is left
    a.n_splits = n_splits;
    a.chunk_keys = n_splits == 1 ? 256 : 128;
    const int n_attn = n_heads * n_splits;
    const int groups = (N + 4 * kQaR - 1) / (4 * kQaR);
    if (grid_cap > 2 * n_attn) grid_cap -= n_attn;  // projection + attention workgroups together stay within what is resident at once
    const int grid = (groups < grid_cap ? groups : grid_cap) + n_attn;
    const size_t smem = (size_t)K * Elem<T>::kBytes;
    if (n_splits == 1) {
      if (head_dim == 128) hipLaunchKernelGGL((gemv_qkv_attn_kernel<T, 128, false>), dim3((unsigned)grid), dim3(kGemvThreads), smem, st, a);
      else hipLaunchKernelGGL((gemv_qkv_attn_kernel<T, 64, false>), dim3((unsigned)grid), dim3(kGemvThreads), smem, st, a);
    } else {
      if (head_dim == 128) hipLaunchKernelGGL((gemv_qkv_attn_kernel<T, 128, true>), dim3((unsigned)grid), dim3(kGemvThreads), smem, st, a);
      else hipLaunchKernelGGL((gemv_qkv_attn_kernel<T, 64, true>), dim3((unsigned)grid), dim3(kGemvThreads), smem, st, a);
    }
  });
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_gemv_qkv_attn");
  return DL_OK;
}


extern "C" int dl_gemv_max_batch(int K, int dtype) {
  const int es = dtype == DL_F32 ? 4 : 2;
  int b = (int)((size_t)(150 * 1024) / ((size_t)K * es));
  return b > kGemvMaxB ? kGemvMaxB : b;
}

extern "C" int dl_gemv(int mode, const void* W, int N, int K, const void* x, int64_t x_row_stride, const void* h_in, void* h_out,
                       const void* delta, const void* norm_w, float eps, void* y, int64_t y_row_stride, int B, int dtype, int grid_cap,
                       void* stream) {
  const void* h = h_in;
  DL_REQUIRE(W && y, "dl_gemv: NULL pointer");
  DL_REQUIRE(N > 0 && K > 0 && B > 0 && grid_cap >= 0, "dl_gemv: bad shape");
  if (grid_cap == 0) grid_cap = kGemvGridCap;
  const int pro = mode & 3;
  DL_REQUIRE((mode & ~(3 | DL_GEMV_OUT_SILU_PAIR)) == 0 && pro <= DL_GEMV_SILUMUL, "dl_gemv: bad mode %d", mode);
  DL_REQUIRE(!(mode & DL_GEMV_OUT_SILU_PAIR) || N % 2 == 0, "dl_gemv: SILU_PAIR needs an even N");
  DL_REQUIRE(pro == DL_GEMV_ADDNORM ? (h && norm_w) : (x != nullptr), "dl_gemv: missing operand for mode %d", mode);
  DL_REQUIRE(!(pro == DL_GEMV_ADDNORM && delta) || (h_out && h_out != h_in), "dl_gemv: h_out must be a distinct buffer when delta is given");
  DL_REQUIRE(B <= dl_gemv_max_batch(K, dtype), "dl_gemv: B=%d rows of K=%d do not fit in LDS (max %d)", B, K, dl_gemv_max_batch(K, dtype));
  hipStream_t st = as_stream(stream);
  int rc = DL_OK;
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(K % Elem<T>::kVec == 0 && (pro == DL_GEMV_ADDNORM || x_row_stride % Elem<T>::kVec == 0), "dl_gemv: K / strides must be multiples of %d", Elem<T>::kVec);
    switch (B) {
      case 1: rc = gemv_launch<T, 1>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      case 2: rc = gemv_launch<T, 2>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      case 3: rc = gemv_launch<T, 3>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      case 4: rc = gemv_launch<T, 4>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      case 5: rc = gemv_launch<T, 5>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      case 6: rc = gemv_launch<T, 6>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      case 7: rc = gemv_launch<T, 7>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
      default: rc = gemv_launch<T, 8>(mode, W, N, K, x, x_row_stride, h, h_out, delta, norm_w, eps, y, y_row_stride, grid_cap, st); break;
    }
  });
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_gemv");
  return DL_OK;
}
