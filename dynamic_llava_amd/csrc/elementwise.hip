// HBM-bound row kernels: RMSNorm (+fused residual add), LayerNorm (+row gather), SiLU*up.
// One 256-thread workgroup per row, 16-byte accesses, the row cached in registers between the
// statistics pass and the scale pass (each element is read from HBM exactly once).
#include "act_round.h"

namespace dl {

// DL_EXACT_ACT=1: dl_quick_gelu / dl_silu_mul evaluate the exact expressions (expf, IEEE divide, software roundings) -- what the guarded fast forms are tested against
static inline bool exact_act() {
  const char* e = getenv("DL_EXACT_ACT");
  return e && e[0] == '1';
}

constexpr int kThreads = 256;
constexpr int kMaxVecPerThread = 8;
constexpr int kPartsBatch = 8;  // split-K slices whose loads are in flight together in the consumers of fp32 partials  // H <= 256 * 8 * kVec  (16384 for 2-byte types, 8192 for fp32)

// ---- RMSNorm: DML:134-139.  ADD: h = cast(h + delta) first (DML:1289/1295), written back. ----
// ADD = 2: delta = cast(sum_s parts[s, row, :]) -- the fp32 split-K partials of dl_gemm_smallm(defer_reduce), summed in slice order.
template <typename T, int ADD>
__global__ __launch_bounds__(kThreads) void rmsnorm_kernel(void* __restrict__ h_, const void* __restrict__ delta_,
                                                            const void* __restrict__ w_, void* __restrict__ out_, int H,
                                                            float eps, int n_slices = 0, int64_t slice_stride = 0, int out_tiles = 0) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  S* h = reinterpret_cast<S*>(h_) + row * H;
  const S* delta = ADD == 1 ? reinterpret_cast<const S*>(delta_) + row * H : nullptr;
  const float* parts = ADD == 2 ? reinterpret_cast<const float*>(delta_) + row * H : nullptr;
  const int nvec = H / V;
  float x[kMaxVecPerThread][V];
  float ss = 0.f;
  if constexpr (ADD == 2) {
    // Vectors are taken in PAIRS and every request of a pair -- the residual rows and all slices of a batch, for both vectors -- is issued
    // before the first addition (H = 4096 in a 2-byte type is exactly one pair per thread).  Vector by vector, the second one's slices were
    // requested only after the first one's additions: a second dependent memory round trip in a launch that is nothing but latency
    // (prefill M = 170: 8.9 us per launch, twice per layer).  The additions keep their slice order: bit-identical.
#pragma unroll
    for (int i0 = 0; i0 < kMaxVecPerThread; i0 += 2) {
      int vv[2];
      bool ok[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int v = threadIdx.x + (i0 + e) * kThreads;
        ok[e] = v < nvec;
        vv[e] = ok[e] ? v : 0;  // out-of-range lanes re-read vector 0 (always there) so that the loads stay unconditional
      }
      if (i0 * kThreads >= nvec) break;  // (uniform) no thread of the workgroup has this pair
      float d[2][V];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        load16<T>(h + vv[e] * V, x[i0 + e]);
#pragma unroll
        for (int j = 0; j < V; ++j) d[e][j] = 0.f;
      }
      for (int s0 = 0; s0 < n_slices; s0 += kPartsBatch) {
        float4 pv[2][kPartsBatch][V / 4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int s = 0; s < kPartsBatch; ++s)
#pragma unroll
            for (int q = 0; q < V / 4; ++q)
              pv[e][s][q] = *reinterpret_cast<const float4*>(parts + (s0 + s < n_slices ? s0 + s : s0) * slice_stride + vv[e] * V + q * 4);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int s = 0; s < kPartsBatch; ++s)
            if (s0 + s < n_slices) {
#pragma unroll
              for (int q = 0; q < V / 4; ++q) {
                d[e][q * 4] += pv[e][s][q].x;
                d[e][q * 4 + 1] += pv[e][s][q].y;
                d[e][q * 4 + 2] += pv[e][s][q].z;
                d[e][q * 4 + 3] += pv[e][s][q].w;
              }
            }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (ok[e]) {
#pragma unroll
          for (int j = 0; j < V; ++j) x[i0 + e][j] = Elem<T>::round(x[i0 + e][j] + Elem<T>::round(d[e][j]));
          store16<T>(h + vv[e] * V, x[i0 + e]);
#pragma unroll
          for (int j = 0; j < V; ++j) ss += x[i0 + e][j] * x[i0 + e][j];
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < kMaxVecPerThread; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec) {
        load16<T>(h + v * V, x[i]);
        if constexpr (ADD == 1) {
          float d[V];
          load16<T>(delta + v * V, d);
#pragma unroll
          for (int j = 0; j < V; ++j) x[i][j] = Elem<T>::round(x[i][j] + d[j]);
          store16<T>(h + v * V, x[i]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) ss += x[i][j] * x[i][j];
      }
    }
  }
  if (w_ == nullptr) return;  // residual add only
  const float tot = block_sum<4>(ss, red);
  const float rstd = rsqrtf(tot / (float)H + eps);
  const S* w = reinterpret_cast<const S*>(w_);
  S* out = reinterpret_cast<S*>(out_) + row * H;
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      float wv[V], o[V];
      load16<T>(w + v * V, wv);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = wv[j] * Elem<T>::round(x[i][j] * rstd);  // cast, THEN weight
      if (out_tiles > 0)  // the consumer is dl_linear_packed: the row's chunks go where its matrix-core fragments expect them
        store16<T>(reinterpret_cast<S*>(out_) + lp_x_chunk_offset(row, v, out_tiles), o);
      else
        store16<T>(out + v * V, o);
    }
  }
}

// ---- nn.LayerNorm over the last dim (predictors: DML:1325,1375; CTL:293,308), optional row gather ----
// ADD (the CLIP encoder layer's `residual + hidden_states` followed by the next LayerNorm, HF modeling_clip CLIPEncoderLayer.forward):
// x = cast(x + delta) first, written back in place.
template <typename T, bool ADD>
__global__ __launch_bounds__(kThreads) void layernorm_kernel(const void* x_, const int32_t* __restrict__ row_index,
                                                              const void* __restrict__ delta_, const void* __restrict__ w_,
                                                              const void* __restrict__ b_, void* __restrict__ out_, int H, float eps) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int64_t src = row_index ? (int64_t)row_index[row] : row;
  const S* xr = reinterpret_cast<const S*>(x_) + src * H;
  const S* delta = ADD == 1 ? reinterpret_cast<const S*>(delta_) + row * H : nullptr;
  const float* parts = ADD == 2 ? reinterpret_cast<const float*>(delta_) + row * H : nullptr;
  const int nvec = H / V;
  float x[kMaxVecPerThread][V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      load16<T>(xr + v * V, x[i]);
      if constexpr (ADD) {
        float d[V];
        load16<T>(delta + v * V, d);
#pragma unroll
        for (int j = 0; j < V; ++j) x[i][j] = Elem<T>::round(x[i][j] + d[j]);
        store16<T>(const_cast<S*>(xr) + v * V, x[i]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) s += x[i][j];
    }
  }
  if (w_ == nullptr) return;  // residual add only
  const float mean = block_sum<4>(s, red) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d = x[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(block_sum<4>(q, red) / (float)H + eps);
  const S* w = reinterpret_cast<const S*>(w_);
  const S* b = reinterpret_cast<const S*>(b_);
  S* out = reinterpret_cast<S*>(out_) + row * H;
#pragma unroll
  for (int i = 0; i < kMaxVecPerThread; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      float wv[V], bv[V], o[V];
      load16<T>(w + v * V, wv);
      load16<T>(b + v * V, bv);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = (x[i][j] - mean) * rstd * wv[j] + bv[j];
      store16<T>(out + v * V, o);
    }
  }
}

// ---- LayerNorm, a WAVE per row (round 6: the CLIP tower's H = 1024 rows between dl_linear_tiles calls) ----
// The block-per-row kernel above spends a 256-thread workgroup, two LDS reductions and two barriers on a 2 KB row (577 workgroups of half-idle
// threads: 5.2 us per launch, twice per encoder layer).  Here a row lives in ONE wave (VPL 16-byte vectors per lane, lanes contiguous: 1 KiB per load
// instruction), both reductions are DPP + readlane, one 64-thread workgroup per row (577 workgroups spread over all CUs) and nothing synchronises.
// ADD = 1: x = cast(x + delta) first (16-bit delta rows).  ADD = 2: delta = cast(sum_s parts[s][row][:] + bias) -- the fp32 k-range partial sums of
// dl_linear_tiles(DL_LT_PARTS), added in range order, then the Linear's bias, one rounding (what F.linear would have returned), then the residual add.
// out_tiles > 0: the normalised row goes out in dl_linear_tiles' fragment order (its x_packed input).
template <typename T, int ADD, int VPL, int NS>
__global__ __launch_bounds__(64) void layernorm_wave_kernel(void* x_, const void* __restrict__ delta_, int n_slices, int64_t slice_stride,
                                                             const void* __restrict__ bias_, const void* __restrict__ w_, const void* __restrict__ b_,
                                                             void* __restrict__ out_, int64_t rows, int H, float eps, int out_tiles) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  const int lane = threadIdx.x;
  const int64_t row = blockIdx.x;
  S* xr = reinterpret_cast<S*>(x_) + row * H;
  const int nvec = H / V;
  float x[VPL][V];
  float s = 0.f;
  if constexpr (ADD == 2) {
    // every request of the row -- the residual vectors, all NS slices of both vectors, the bias -- is issued before the first addition: the launch is one
    // memory round trip deep (first build, slice after slice: 8.2 us for 13 MB).  NS = 0: any slice count, slice by slice.
    static_assert(V == 8, "16-bit types");
    float4 pv[VPL][NS > 0 ? NS : 1][2];
    float bb[VPL][V];
    int vv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int v = lane + 64 * i;
      vv[i] = v < nvec ? v : 0;  // out-of-range lanes re-read vector 0 so that the loads stay unconditional
      load16<T>(xr + vv[i] * V, x[i]);
      const float* pp = reinterpret_cast<const float*>(delta_) + row * H + vv[i] * V;
      if constexpr (NS > 0) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          pv[i][sl][0] = *reinterpret_cast<const float4*>(pp + sl * slice_stride);
          pv[i][sl][1] = *reinterpret_cast<const float4*>(pp + sl * slice_stride + 4);
        }
      }
      if (bias_) load16<T>(reinterpret_cast<const S*>(bias_) + vv[i] * V, bb[i]);
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float d[V];
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = 0.f;
      if constexpr (NS > 0) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          d[0] += pv[i][sl][0].x; d[1] += pv[i][sl][0].y; d[2] += pv[i][sl][0].z; d[3] += pv[i][sl][0].w;
          d[4] += pv[i][sl][1].x; d[5] += pv[i][sl][1].y; d[6] += pv[i][sl][1].z; d[7] += pv[i][sl][1].w;
        }
      } else {
        const float* pp = reinterpret_cast<const float*>(delta_) + row * H + vv[i] * V;
        for (int sl = 0; sl < n_slices; ++sl) {
          const float4 a = *reinterpret_cast<const float4*>(pp + sl * slice_stride), c = *reinterpret_cast<const float4*>(pp + sl * slice_stride + 4);
          d[0] += a.x; d[1] += a.y; d[2] += a.z; d[3] += a.w;
          d[4] += c.x; d[5] += c.y; d[6] += c.z; d[7] += c.w;
        }
      }
      if (bias_) {
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] += bb[i][j];
      }
      if (lane + 64 * i < nvec) {
#pragma unroll
        for (int j = 0; j < V; ++j) x[i][j] = Elem<T>::round(x[i][j] + Elem<T>::round(d[j]));
        store16<T>(xr + vv[i] * V, x[i]);
#pragma unroll
        for (int j = 0; j < V; ++j) s += x[i][j];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
        load16<T>(xr + v * V, x[i]);
        if constexpr (ADD == 1) {
          float d[V];
          load16<T>(reinterpret_cast<const S*>(delta_) + row * H + v * V, d);
#pragma unroll
          for (int j = 0; j < V; ++j) x[i][j] = Elem<T>::round(x[i][j] + d[j]);
          store16<T>(xr + v * V, x[i]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) s += x[i][j];
      }
    }
  }
  if (w_ == nullptr) return;  // residual add only
  // the affine parameters are requested BEFORE the two reductions (their round trip hides behind the statistics instead of following them)
  const S* w = reinterpret_cast<const S*>(w_);
  const S* b = reinterpret_cast<const S*>(b_);
  uint4 wraw[VPL], braw[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = lane + 64 * i < nvec ? lane + 64 * i : 0;
    wraw[i] = *reinterpret_cast<const uint4*>(w + v * V);
    braw[i] = *reinterpret_cast<const uint4*>(b + v * V);
    pin_reg(wraw[i]);
    pin_reg(braw[i]);
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (lane + 64 * i < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d = x[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  S* out = reinterpret_cast<S*>(out_);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      float wv[V], bv[V], o[V];
      load16<T>(&wraw[i], wv);
      load16<T>(&braw[i], bv);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = (x[i][j] - mean) * rstd * wv[j] + bv[j];
      if (out_tiles > 0)
        store16<T>(out + lp_x_chunk_offset(row, v, out_tiles), o);
      else
        store16<T>(out + row * H + v * V, o);
    }
  }
}

template <typename T, int ADD>
static int ln_wave_launch(void* x, const void* delta, int n_slices, const void* bias, const void* w, const void* b, void* out, int64_t rows, int H, float eps,
                          int out_packed, hipStream_t st) {
  const int nvec = H / Elem<T>::kVec;
  const int vpl = (nvec + 63) / 64;
  const int out_tiles = out_packed ? (int)((rows + 15) / 16) : 0;
  const dim3 grid((unsigned)rows);
#define LN_WAVE_GO(v_, ns_)                                                                                                                                 \
  {                                                                                                                                                         \
    hipLaunchKernelGGL((layernorm_wave_kernel<T, ADD, v_, ns_>), grid, dim3(64), 0, st, x, delta, n_slices, (int64_t)rows * H, bias, w, b, out, rows, H,    \
                       eps, out_tiles);                                                                                                                     \
    return DL_OK;                                                                                                                                           \
  }
#define LN_WAVE_CASE(v_)                                         \
  case v_:                                                       \
    if constexpr (ADD == 2) {                                    \
      if (v_ <= 2 && n_slices == 2) LN_WAVE_GO(v_, 2)            \
      if (v_ <= 2 && n_slices == 4) LN_WAVE_GO(v_, 4)            \
      LN_WAVE_GO(v_, 0)                                          \
    } else                                                       \
      LN_WAVE_GO(v_, 0)
  switch (vpl) {
    LN_WAVE_CASE(1);
    LN_WAVE_CASE(2);
    LN_WAVE_CASE(3);
    LN_WAVE_CASE(4);
    LN_WAVE_CASE(8);
  }
#undef LN_WAVE_CASE
#undef LN_WAVE_GO
  set_error("layernorm (wave per row): H=%d is not built (H / 8 <= 256 or == 512 vectors of 16 bytes)", H);
  return DL_ERR_ARG;
}

// ---- CLIP's QuickGELU (HF activations.QuickGELUActivation: `input * torch.sigmoid(1.702 * input)`), three roundings as in eager ----
// EXACT = false (the product, 16-bit types): sigmoid on v_exp_f32 / v_rcp_f32 with the rounding-boundary guard of act_round.h -- the bits of the exact expression
// (tests hold the two instantiations against each other; DL_EXACT_ACT=1 selects the exact one), at a third of its VALU: at a batch of images this launch was
// VALU-bound (75 M elements x ~40 instructions: 98 us where the bytes need 60).
template <typename T, bool EXACT>
__global__ __launch_bounds__(kThreads) void quick_gelu_kernel(const void* __restrict__ x_, void* __restrict__ out_, int64_t nvec) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  const S* x = reinterpret_cast<const S*>(x_);
  S* out = reinterpret_cast<S*>(out_);
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < nvec; idx += (int64_t)gridDim.x * kThreads) {
    float a[V], o[V];
    load16<T>(x + idx * V, a);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if constexpr (EXACT || Elem<T>::kBytes == 4) {
        const float t = Elem<T>::round(1.702f * a[j]);
        const float sg = Elem<T>::round(1.0f / (1.0f + expf(-t)));
        o[j] = a[j] * sg;
      } else {
        o[j] = a[j] * sigmoid_rounded<T>(hw_round<T>(1.702f * a[j]));
      }
    }
    store16<T>(out + idx * V, o);
  }
}

// ---- act_fn(gate) * up, DML:328 (two roundings: after silu, after the product) ----
template <typename T, bool EXACT>
__global__ __launch_bounds__(kThreads) void silu_mul_kernel(const void* __restrict__ gu_, void* __restrict__ out_, int64_t rows,
                                                             int I) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  const S* gu = reinterpret_cast<const S*>(gu_);
  S* out = reinterpret_cast<S*>(out_);
  const int vec_per_row = I / V;
  const int64_t total = rows * vec_per_row;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
    const int64_t r = idx / vec_per_row;
    const int c = (int)(idx - r * vec_per_row) * V;
    float g[V], u[V], o[V];
    load16<T>(gu + r * 2 * I + c, g);
    load16<T>(gu + r * 2 * I + I + c, u);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float sg;
      if constexpr (EXACT || Elem<T>::kBytes == 4) sg = Elem<T>::round(g[j] / (1.0f + expf(-g[j])));
      else sg = silu_rounded<T>(g[j]);  // (act_round.h: the exact expression's bits; dl_silu_mul_parts keeps the exact form)
      o[j] = sg * u[j];
    }
    store16<T>(out + r * I + c, o);
  }
}

// act_fn(gate) * up (DML:328) reading the gate|up projection as fp32 split-K partials [n_slices][rows][2I] (dl_gemm_smallm with
// defer_reduce): gate / up = cast(sum over slices), then the two roundings of the eager op.
template <typename T>
__global__ __launch_bounds__(kThreads) void silu_mul_parts_kernel(const float* __restrict__ parts, int n_slices, void* __restrict__ out_,
                                                                   int64_t rows, int I) {
  const int q_per_row = I / 4;
  const int64_t total = rows * q_per_row, slice_stride = rows * 2 * (int64_t)I;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
    const int64_t r = idx / q_per_row;
    const int c = (int)(idx - r * q_per_row) * 4;
    float g[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < n_slices; s0 += kPartsBatch) {  // loads of a batch of slices first, additions in slice order
      float4 pg[kPartsBatch], pu[kPartsBatch];
#pragma unroll
      for (int s = 0; s < kPartsBatch; ++s) {
        const int64_t so = (int64_t)(s0 + s < n_slices ? s0 + s : s0) * slice_stride;
        pg[s] = *reinterpret_cast<const float4*>(parts + so + r * 2 * I + c);
        pu[s] = *reinterpret_cast<const float4*>(parts + so + r * 2 * I + I + c);
      }
#pragma unroll
      for (int s = 0; s < kPartsBatch; ++s)
        if (s0 + s < n_slices) {
          g[0] += pg[s].x; g[1] += pg[s].y; g[2] += pg[s].z; g[3] += pg[s].w;
          u[0] += pu[s].x; u[1] += pu[s].y; u[2] += pu[s].z; u[3] += pu[s].w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = Elem<T>::round(g[j]), uj = Elem<T>::round(u[j]);
      const float sg = Elem<T>::round(gj / (1.0f + expf(-gj)));
      store1<T>(out_, r * I + c + j, sg * uj);
    }
  }
}

}  // namespace dl

using namespace dl;

extern "C" int dl_rmsnorm(const void* x, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream) {
  DL_REQUIRE(rows >= 0 && H > 0, "dl_rmsnorm: bad shape rows=%lld H=%d", (long long)rows, H);
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(x && w && out, "dl_rmsnorm: NULL pointer");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0 && H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_rmsnorm: unsupported H=%d", H);
    hipLaunchKernelGGL((rmsnorm_kernel<T, 0>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), const_cast<void*>(x),
                       nullptr, w, out, H, eps);
  });
  DL_CHECK_LAUNCH("dl_rmsnorm");
  return DL_OK;
}

extern "C" int dl_add_rmsnorm(void* h, const void* delta, const void* w, void* out, int64_t rows, int H, float eps, int dtype,
                              void* stream) {
  DL_REQUIRE(rows >= 0 && H > 0, "dl_add_rmsnorm: bad shape");
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(h && delta, "dl_add_rmsnorm: NULL pointer");
  DL_REQUIRE((w == nullptr) == (out == nullptr), "dl_add_rmsnorm: w and out must both be given or both be NULL");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0 && H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_add_rmsnorm: unsupported H=%d", H);
    hipLaunchKernelGGL((rmsnorm_kernel<T, 1>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), h, delta, w, out, H,
                       eps);
  });
  DL_CHECK_LAUNCH("dl_add_rmsnorm");
  return DL_OK;
}

extern "C" int dl_add_rmsnorm_parts(void* h, const float* parts, int n_slices, const void* w, void* out, int64_t rows, int H, float eps,
                                    int dtype, void* stream) {
  DL_REQUIRE(rows >= 0 && H > 0 && ((uintptr_t)parts & 15) == 0, "dl_add_rmsnorm_parts: bad shape / alignment");
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(h && parts && n_slices >= 1, "dl_add_rmsnorm_parts: bad arguments");
  DL_REQUIRE((w == nullptr) == (out == nullptr), "dl_add_rmsnorm_parts: w and out must both be given or both be NULL");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0 && H % 4 == 0 && H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_add_rmsnorm_parts: unsupported H=%d", H);
    hipLaunchKernelGGL((rmsnorm_kernel<T, 2>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), h, parts, w, out, H, eps, n_slices,
                       (int64_t)rows * H);
  });
  DL_CHECK_LAUNCH("dl_add_rmsnorm_parts");
  return DL_OK;
}

// ---- the same three launches with the normalised rows written in dl_linear_packed's activation order (dl_pack_x_tiles): 2-byte types, H % 64 == 0 ----
#define DL_PACKED_NORM_CHECKS(name)                                                                                                              \
  DL_REQUIRE(rows >= 0 && rows <= 256 && H > 0 && H % 64 == 0, name ": rows=%lld (<= 256), H=%d (multiple of 64)", (long long)rows, H);          \
  if (rows == 0) return DL_OK;                                                                                                                   \
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, name ": bf16 / fp16 only");                                                                    \
  DL_REQUIRE(w && out && ((uintptr_t)out & 15) == 0, name ": NULL / unaligned pointer")

extern "C" int dl_rmsnorm_packed(const void* x, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream) {
  DL_PACKED_NORM_CHECKS("dl_rmsnorm_packed");
  DL_REQUIRE(x && x != out, "dl_rmsnorm_packed: x is NULL or aliases out");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_rmsnorm_packed: unsupported H=%d", H);
    hipLaunchKernelGGL((rmsnorm_kernel<T, 0>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), const_cast<void*>(x), nullptr, w, out, H, eps, 0, (int64_t)0,
                       lp_x_tiles(rows));
  });
  DL_CHECK_LAUNCH("dl_rmsnorm_packed");
  return DL_OK;
}

extern "C" int dl_add_rmsnorm_packed(void* h, const void* delta, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream) {
  DL_PACKED_NORM_CHECKS("dl_add_rmsnorm_packed");
  DL_REQUIRE(h && delta && h != out, "dl_add_rmsnorm_packed: NULL pointer / out aliases h");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_add_rmsnorm_packed: unsupported H=%d", H);
    hipLaunchKernelGGL((rmsnorm_kernel<T, 1>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), h, delta, w, out, H, eps, 0, (int64_t)0, lp_x_tiles(rows));
  });
  DL_CHECK_LAUNCH("dl_add_rmsnorm_packed");
  return DL_OK;
}

extern "C" int dl_add_rmsnorm_parts_packed(void* h, const float* parts, int n_slices, const void* w, void* out, int64_t rows, int H, float eps, int dtype,
                                           void* stream) {
  DL_PACKED_NORM_CHECKS("dl_add_rmsnorm_parts_packed");
  DL_REQUIRE(h && parts && n_slices >= 1 && ((uintptr_t)parts & 15) == 0 && h != out, "dl_add_rmsnorm_parts_packed: bad arguments");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % 4 == 0 && H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_add_rmsnorm_parts_packed: unsupported H=%d", H);
    hipLaunchKernelGGL((rmsnorm_kernel<T, 2>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), h, parts, w, out, H, eps, n_slices, (int64_t)rows * H,
                       lp_x_tiles(rows));
  });
  DL_CHECK_LAUNCH("dl_add_rmsnorm_parts_packed");
  return DL_OK;
}

extern "C" int dl_layernorm(const void* x, const int32_t* row_index, const void* w, const void* b, void* out, int64_t rows, int H,
                            float eps, int dtype, void* stream) {
  DL_REQUIRE(rows >= 0 && H > 0, "dl_layernorm: bad shape");
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(x && w && b && out, "dl_layernorm: NULL pointer");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0 && H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_layernorm: unsupported H=%d", H);
    hipLaunchKernelGGL((layernorm_kernel<T, false>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), x, row_index, nullptr, w,
                       b, out, H, eps);
  });
  DL_CHECK_LAUNCH("dl_layernorm");
  return DL_OK;
}

extern "C" int dl_add_layernorm(void* h, const void* delta, const void* w, const void* b, void* out, int64_t rows, int H, float eps,
                                int dtype, void* stream) {
  DL_REQUIRE(rows >= 0 && H > 0, "dl_add_layernorm: bad shape");
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(h && delta, "dl_add_layernorm: NULL pointer");
  DL_REQUIRE((w == nullptr) == (out == nullptr) && (w == nullptr) == (b == nullptr), "dl_add_layernorm: w, b and out must all be given or all be NULL");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0 && H <= kThreads * kMaxVecPerThread * Elem<T>::kVec, "dl_add_layernorm: unsupported H=%d", H);
    hipLaunchKernelGGL((layernorm_kernel<T, true>), dim3((unsigned)rows), dim3(kThreads), 0, as_stream(stream), h, nullptr, delta, w, b,
                       out, H, eps);
  });
  DL_CHECK_LAUNCH("dl_add_layernorm");
  return DL_OK;
}

#define DL_LN_ROWS_CHECKS(name)                                                                                                                     \
  DL_REQUIRE(rows >= 0 && H > 0, name ": bad shape");                                                                                               \
  if (rows == 0) return DL_OK;                                                                                                                      \
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, name ": bf16 / fp16 only");                                                                       \
  DL_REQUIRE(H % 8 == 0 && (!out_packed || H % 64 == 0), name ": H=%d must be a multiple of 8 (of 64 for a fragment-order output)", H);             \
  DL_REQUIRE((w == nullptr) == (out == nullptr) && (w == nullptr) == (b == nullptr), name ": w, b and out must all be given or all be NULL");      \
  DL_REQUIRE(!out || ((uintptr_t)out & 15) == 0, name ": unaligned output")

extern "C" int dl_layernorm_rows(const void* x, const void* w, const void* b, void* out, int64_t rows, int H, float eps, int out_packed, int dtype, void* stream) {
  DL_LN_ROWS_CHECKS("dl_layernorm_rows");
  DL_REQUIRE(x && w && x != out, "dl_layernorm_rows: NULL pointer / out aliases x");
  int rc;
  if (dtype == DL_BF16)
    rc = ln_wave_launch<bf16_t, 0>(const_cast<void*>(x), nullptr, 0, nullptr, w, b, out, rows, H, eps, out_packed, as_stream(stream));
  else
    rc = ln_wave_launch<f16_t, 0>(const_cast<void*>(x), nullptr, 0, nullptr, w, b, out, rows, H, eps, out_packed, as_stream(stream));
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_layernorm_rows");
  return DL_OK;
}

extern "C" int dl_add_layernorm_rows(void* h, const void* delta, const void* w, const void* b, void* out, int64_t rows, int H, float eps, int out_packed, int dtype,
                                     void* stream) {
  DL_LN_ROWS_CHECKS("dl_add_layernorm_rows");
  DL_REQUIRE(h && delta && h != out, "dl_add_layernorm_rows: NULL pointer / out aliases h");
  int rc;
  if (dtype == DL_BF16)
    rc = ln_wave_launch<bf16_t, 1>(h, delta, 0, nullptr, w, b, out, rows, H, eps, out_packed, as_stream(stream));
  else
    rc = ln_wave_launch<f16_t, 1>(h, delta, 0, nullptr, w, b, out, rows, H, eps, out_packed, as_stream(stream));
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_add_layernorm_rows");
  return DL_OK;
}

extern "C" int dl_add_layernorm_parts(void* h, const float* parts, int n_slices, const void* bias, const void* w, const void* b, void* out, int64_t rows, int H,
                                      float eps, int out_packed, int dtype, void* stream) {
  DL_LN_ROWS_CHECKS("dl_add_layernorm_parts");
  DL_REQUIRE(h && parts && n_slices >= 1 && ((uintptr_t)parts & 15) == 0 && h != out, "dl_add_layernorm_parts: bad arguments");
  int rc;
  if (dtype == DL_BF16)
    rc = ln_wave_launch<bf16_t, 2>(h, parts, n_slices, bias, w, b, out, rows, H, eps, out_packed, as_stream(stream));
  else
    rc = ln_wave_launch<f16_t, 2>(h, parts, n_slices, bias, w, b, out, rows, H, eps, out_packed, as_stream(stream));
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_add_layernorm_parts");
  return DL_OK;
}

extern "C" int dl_quick_gelu(const void* x, void* out, int64_t n, int dtype, void* stream) {
  DL_REQUIRE(n >= 0, "dl_quick_gelu: bad size");
  if (n == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(x && out, "dl_quick_gelu: NULL pointer");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(n % Elem<T>::kVec == 0, "dl_quick_gelu: n must be a multiple of %d", Elem<T>::kVec);
    const int64_t nvec = n / Elem<T>::kVec;
    const int64_t blocks = (nvec + kThreads - 1) / kThreads;
    if (exact_act()) {
      hipLaunchKernelGGL((quick_gelu_kernel<T, true>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kThreads), 0, as_stream(stream), x, out,
                       nvec);
    } else {
      hipLaunchKernelGGL((quick_gelu_kernel<T, false>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kThreads), 0, as_stream(stream), x, out,
                       nvec);
    }
  });
  DL_CHECK_LAUNCH("dl_quick_gelu");
  return DL_OK;
}

extern "C" int dl_silu_mul_parts(const float* parts, int n_slices, void* out, int64_t rows, int I, int dtype, void* stream) {
  DL_REQUIRE(rows >= 0 && I > 0 && I % 4 == 0 && ((uintptr_t)parts & 15) == 0, "dl_silu_mul_parts: bad shape / alignment");
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(parts && out && n_slices >= 1, "dl_silu_mul_parts: bad arguments");
  DL_DISPATCH_DTYPE(dtype, T, {
    const int64_t total = rows * (I / 4);
    const int64_t blocks = (total + kThreads - 1) / kThreads;
    hipLaunchKernelGGL((silu_mul_parts_kernel<T>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kThreads), 0, as_stream(stream), parts,
                       n_slices, out, rows, I);
  });
  DL_CHECK_LAUNCH("dl_silu_mul_parts");
  return DL_OK;
}

extern "C" int dl_silu_mul(const void* gate_up, void* out, int64_t rows, int I, int dtype, void* stream) {
  DL_REQUIRE(rows >= 0 && I > 0, "dl_silu_mul: bad shape");
  if (rows == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(gate_up && out, "dl_silu_mul: NULL pointer");
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(I % Elem<T>::kVec == 0, "dl_silu_mul: I=%d must be a multiple of %d", I, Elem<T>::kVec);
    const int64_t total = rows * (I / Elem<T>::kVec);
    const int64_t blocks = (total + kThreads - 1) / kThreads;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    if (exact_act()) hipLaunchKernelGGL((silu_mul_kernel<T, true>), dim3(grid), dim3(kThreads), 0, as_stream(stream), gate_up, out, rows, I);
    else hipLaunchKernelGGL((silu_mul_kernel<T, false>), dim3(grid), dim3(kThreads), 0, as_stream(stream), gate_up, out, rows, I);
  });
  DL_CHECK_LAUNCH("dl_silu_mul");
  return DL_OK;
}
