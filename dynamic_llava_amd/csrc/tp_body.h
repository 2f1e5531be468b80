// Text predictor (TextPredictor.forward DML:1385-1387 + the decision DML:2388-2391) as device functions shared by predictors.hip (three
// stand-alone launches) and gemv.hip (dl_gemv_gu_tp: the same stages as extra workgroups of the gate|up projection launch).
#pragma once
#include "dl_common.h"
#include "granule.h"

namespace dl {

// raw 16-byte chunk -> kVec floats (weights prefetched into registers stay packed until they are used)
template <typename T>
__device__ __forceinline__ void tp_unpack(const uint4& r, float (&f)[Elem<T>::kVec]) {
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

// ---- text predictor, stage 1: LN(H) + Linear(H -> D) + GELU.  grid (ceil(D/8), B): a wave owns 2 neurons ----
// Single-instance latency kernel on the decode step's critical path: the weight rows (cold HBM) are requested before the
// LayerNorm chain starts, x / ln_w / ln_b are 16-byte loads issued together -- one HBM round trip instead of ~5.
constexpr int kTp1MaxChunks = 10;   // 16-byte chunks per lane per weight row: H <= 64 * 8 * 10 = 5120
// The stages are device functions (bx / by = the block coordinates of the stand-alone grids) so that dl_gemv_gu_tp can run them as extra
// workgroups of a projection launch; `gran` != nullptr: every output is also published as a granule {gtag, float bits} for a consumer in
// the same launch.
template <typename T, int MAXC = kTp1MaxChunks>
__device__ __forceinline__ void tp_stage1_body(const void* __restrict__ x_, int64_t x_rs, const void* __restrict__ ln_w,
                                               const void* __restrict__ ln_b, const void* __restrict__ w1, const void* __restrict__ b1,
                                               float* __restrict__ h1, int H, int D, const int bx, const int by, u64_t* gran, uint32_t gtag) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  extern __shared__ float xs[];  // [H]
  __shared__ float red[4];
  const int b = by;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nvec = H / V;
  const S* W = reinterpret_cast<const S*>(w1);
  const int nb = bx * 8 + wid * 2;  // this wave's 2 neurons
  uint4 wv[2][MAXC];
  const bool pre = nvec <= 64 * MAXC;  // else (fp32 at full width): plain streaming loop below
  if (pre) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nb + j < D ? nb + j : D - 1;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (lane + 64 * c < nvec) wv[j][c] = *reinterpret_cast<const uint4*>(W + (int64_t)n * H + (lane + 64 * c) * V);
    }
  }
  // LayerNorm over the row (every workgroup redoes it: 8 KB from L2)
  const S* xr = reinterpret_cast<const S*>(x_) + (int64_t)b * x_rs;
  float s = 0.f;
  for (int v = tid; v < nvec; v += 256) {
    float a[V];
    load16<T>(xr + v * V, a);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      xs[v * V + e] = a[e];
      s += a[e];
    }
  }
  const float mean = block_sum<4>(s, red) / (float)H;
  float q = 0.f;
  for (int v = tid; v < nvec; v += 256)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float d = xs[v * V + e] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(block_sum<4>(q, red) / (float)H + 1e-5f);
  for (int v = tid; v < nvec; v += 256) {
    float g[V], be[V];
    load16<T>(reinterpret_cast<const S*>(ln_w) + v * V, g);
    load16<T>(reinterpret_cast<const S*>(ln_b) + v * V, be);
#pragma unroll
    for (int e = 0; e < V; ++e) xs[v * V + e] = Elem<T>::round((xs[v * V + e] - mean) * rstd * g[e] + be[e]);
  }
  __syncthreads();
  float acc[2] = {0.f, 0.f};
  if (!pre) {
    for (int v = lane; v < nvec; v += 64)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float wf[V];
        load16<T>(W + (int64_t)(nb + j < D ? nb + j : D - 1) * H + v * V, wf);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[j] = fmaf(wf[e], xs[v * V + e], acc[j]);
      }
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (pre && lane + 64 * c < nvec) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float wf[V];
        tp_unpack<T>(wv[j][c], wf);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[j] = fmaf(wf[e], xs[(lane + 64 * c) * V + e], acc[j]);
      }
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a = wave_sum(acc[j]);
    const int n = nb + j;
    if (lane == 0 && n < D) {
      const float y = Elem<T>::round(gelu_erf(Elem<T>::round(a + load1<T>(b1, n))));
      h1[(int64_t)b * D + n] = y;
      if (gran) gr_store(gran + (int64_t)b * D + n, gtag, __float_as_uint(y));
    }
  }
}

template <typename T>
__device__ __forceinline__ void tp_dense(const float* in, float* out, const void* w_, const void* b_, int K, int N, bool gelu) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  const S* W = reinterpret_cast<const S*>(w_);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nvec = K / V;
  for (int n0 = wid * 4; n0 < N; n0 += nw * 4) {  // 4 output neurons per wave per pass: 4 weight rows in flight
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int v = lane; v < nvec; v += 64) {
      float wv[4][V];
#pragma unroll
      for (int j = 0; j < 4; ++j) load16<T>(W + (int64_t)(n0 + j < N ? n0 + j : N - 1) * K + v * V, wv[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[j] = fmaf(wv[j][e], in[v * V + e], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = wave_sum(acc[j]);
      if (lane == 0 && n0 + j < N) {
        const float y = Elem<T>::round(a + load1<T>(b_, n0 + j));
        out[n0 + j] = gelu ? Elem<T>::round(gelu_erf(y)) : y;
      }
    }
  }
  __syncthreads();
}

// ---- text predictor, stage 2a: Linear(D -> D/2) + GELU spread over the chip.  grid (ceil(D/16), B): a wave owns 2 neurons ----
// (as one workgroup per row this layer had to pull its 262 KB of weights through a single CU: 18.8 us for all of stage 2)
template <typename T>
__device__ __forceinline__ void tp_stage2a_body(const float* __restrict__ h1, const void* __restrict__ w3, const void* __restrict__ b3,
                                                float* __restrict__ a1, int D, const int bx, const int by, u64_t* gran, uint32_t gtag) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  const int b = by, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int N3 = D / 2, nvec = D / V;
  const int nb = bx * 8 + wid * 2;
  float acc[2] = {0.f, 0.f};
  for (int v = lane; v < nvec; v += 64) {
    float wv[2][V], xv[V];
#pragma unroll
    for (int j = 0; j < 2; ++j) load16<T>(reinterpret_cast<const S*>(w3) + (int64_t)(nb + j < N3 ? nb + j : N3 - 1) * D + v * V, wv[j]);
#pragma unroll
    for (int e = 0; e < V; ++e) xv[e] = h1[(int64_t)b * D + v * V + e];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[j] = fmaf(wv[j][e], xv[e], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a = wave_sum(acc[j]);
    const int n = nb + j;
    if (lane == 0 && n < N3) {
      const float y = Elem<T>::round(gelu_erf(Elem<T>::round(a + load1<T>(b3, n))));
      a1[(int64_t)b * N3 + n] = y;
      if (gran) gr_store(gran + (int64_t)b * N3 + n, gtag, __float_as_uint(y));
    }
  }
}

// ---- text predictor, stage 2b: D/2 -> D/4 -> 2 and the keep/evict decision.  grid (B) ----
// `wait_input` (default: nothing) is called by every thread after the weights have been requested and before the input row is read: inside
// dl_gemv_gu_tp the stage waits there for stage 2a's granules.  When a lane's share of the D/2 -> D/4 layer fits (<= 8 passes of 4 neurons per
// wave, one 16-byte chunk per row) ALL its weight rows and biases are requested up front instead of pass by pass: beside a weight stream
// every dependent round trip costs ~2 us (8 passes: 13 us, which pushed the fused launch 9 us past the end of its projection).  Same chunk ->
// lane dealing, same accumulation order as tp_dense: same bits.
struct TpNoWait {
  __device__ __forceinline__ void operator()() const {}
};
template <typename T, typename WaitInput = TpNoWait>
__device__ __forceinline__ void tp_stage2b_body(const float* __restrict__ a1g, const void* w5, const void* b5, const void* w7, const void* b7,
                                                float* __restrict__ logits, int32_t* __restrict__ decision, int D, const int bx, float* sm,
                                                WaitInput wait_input = WaitInput()) {
  // sm: [D/2] + [D/4] + [2] floats of LDS
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  float* a1 = sm;
  float* a2 = a1 + D / 2;
  float* a3 = a2 + D / 4;
  const int b = bx;
  const int K5 = D / 2, N5 = D / 4;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int passes = (N5 + nw * 4 - 1) / (nw * 4);
  const bool pre = K5 % V == 0 && K5 / V <= 64 && passes <= 8;
  constexpr int PB = 2;  // passes whose weight rows are in flight together: 32 VGPRs (eight at once made dl_gemv_gu_tp a 237-VGPR kernel: two workgroups per CU)
  uint4 wq[PB][4];
  float bq[PB][4];
  auto request = [&](int p0) {
    const S* W = reinterpret_cast<const S*>(w5);
#pragma unroll
    for (int ps = 0; ps < PB; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = wid * 4 + (p0 + ps) * nw * 4 + j;
        n = n < N5 ? n : N5 - 1;
        if (lane < K5 / V) wq[ps][j] = *reinterpret_cast<const uint4*>(W + (int64_t)n * K5 + lane * V);
        bq[ps][j] = load1<T>(b5, n);
      }
  };
  if (pre) {
    request(0);
    // really requested NOW (the compiler may otherwise sink these loads below wait_input()'s polling loop -- what "requests its weights before it
    // waits" was written to avoid; found with the fused attention launch, tools/qa_timing.hip)
#pragma unroll
    for (int ps = 0; ps < PB; ++ps)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pin_reg(wq[ps][j]);
        pin_reg(bq[ps][j]);
      }
  }
  wait_input();
  for (int i = threadIdx.x; i < D / 2; i += blockDim.x) a1[i] = a1g[(int64_t)b * (D / 2) + i];
  __syncthreads();
  if (pre) {
    for (int p0 = 0; p0 < passes; p0 += PB) {
      if (p0 > 0) request(p0);
#pragma unroll
      for (int ps = 0; ps < PB; ++ps) {
        const int n0 = wid * 4 + (p0 + ps) * nw * 4;
        if (n0 < N5) {
          float acc[4] = {0.f, 0.f, 0.f, 0.f};
          if (lane < K5 / V) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float wf[V];
              tp_unpack<T>(wq[ps][j], wf);
#pragma unroll
              for (int e = 0; e < V; ++e) acc[j] = fmaf(wf[e], a1[lane * V + e], acc[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = wave_sum(acc[j]);
            if (lane == 0 && n0 + j < N5) a2[n0 + j] = Elem<T>::round(gelu_erf(Elem<T>::round(a + bq[ps][j])));
          }
        }
      }
    }
    __syncthreads();
  } else {
    tp_dense<T>(a1, a2, w5, b5, D / 2, D / 4, true);
  }
  tp_dense<T>(a2, a3, w7, b7, D / 4, 2, false);
  if (threadIdx.x == 0) {
    if (logits) {
      logits[b * 2] = a3[0];
      logits[b * 2 + 1] = a3[1];
    }
    decision[b] = a3[0] > a3[1] ? 1 : 0;  // strict '>' on raw logits, DML:2388-2391
  }
}

}  // namespace dl
