// F9 (decode) + F11: one query token per row against a ragged, evicted KV slab.
//
// HBM-bound (AI ~ 1 flop/byte): the only thing that matters is keeping ~every CU streaming K/V rows
// with 16-byte loads.  Split-KV: grid = (n_splits, n_heads, B); a 256-thread workgroup owns one
// contiguous key range of one (row, head).  Inside a wave, D/kVec lanes cooperate on one key
// (16 lanes x 16 B = one 256-byte K row for D=128 bf16), so a wave-wide load instruction is 4 full
// rows = 1 KiB fully coalesced.  Each lane group keeps its own online-softmax state (m, l, o[kVec]);
// groups merge through LDS once per workgroup, splits merge in attn_decode_combine_kernel.
// Per-row lengths are read from the device (kv_len[b] + extra): no host sync, graph-capturable,
// and a slot beyond the true length (an evicted token) is simply never read.
#include "dl_common.h"

namespace dl {

// tools/attn_timing.hip compiles this file with -DDL_ATTN_TIMING to stamp the phases of one workgroup (100 MHz wall clock).
#ifdef DL_ATTN_TIMING
__device__ long long g_attn_stamps[8];
#define DL_STAMP(i, drain)                                                                        \
  do {                                                                                            \
    if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_attn_stamps[i] = wall_clock64(); \
  } while (0)
#else
#define DL_STAMP(i, drain)
#endif

// U = key rows each lane group requests per loop trip.  U = 16 puts 256 keys (NW = 4) in flight per workgroup in one round trip:
// at decode batch 1 the kernel's time is the number of dependent HBM round trips, not bytes.

// FUSED: the RoPE of q and of the new key (DML:260-285) and the KV-slab append (CU:109-268) happen inside the attention
// kernel: q|k|v are read un-rotated from the projection output, the new token's rotated key / value are used from
// registers by the one lane group that owns key index kv_len[b] and written to slab slot kv_len[b] for later steps.
template <typename T, bool UPPER>
__device__ __forceinline__ void rope16(const float (&own)[Elem<T>::kVec], const float (&par)[Elem<T>::kVec], const float (&cs)[Elem<T>::kVec],
                                       const float (&sn)[Elem<T>::kVec], float (&out)[Elem<T>::kVec]) {
#pragma unroll
  for (int i = 0; i < Elem<T>::kVec; ++i)  // x*cos + rotate_half(x)*sin, each op rounded (DML:283-284); rotate_half = cat(-x2, x1)
    out[i] = Elem<T>::round(Elem<T>::round(own[i] * cs[i]) + Elem<T>::round((UPPER ? par[i] : -par[i]) * sn[i]));
}

// sum over the LPK lanes that share one key (8, 16 or 32 lanes, aligned): DPP inside a 16-lane row, one crossbar step beyond it
template <int LPK>
__device__ __forceinline__ float lpk_sum(float a) {
  if constexpr (LPK == 8) return row8_sum(a);
  else if constexpr (LPK == 16) return row16_sum(a);
  else if constexpr (LPK == 32) {
    a = row16_sum(a);
    return a + __shfl_xor(a, 16, 64);
  } else {
#pragma unroll
    for (int w = LPK / 2; w > 0; w >>= 1) a += __shfl_xor(a, w, 64);
    return a;
  }
}

template <typename T>
__device__ __forceinline__ void unpack_kv(const uint4& r, float (&f)[Elem<T>::kVec]) {
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

template <typename T, int D, int NW, bool FUSED, int U>
__global__ __launch_bounds__(NW * 64) void attn_decode_split_kernel(
    const void* __restrict__ q_, int64_t q_row_stride, const void* k_slab_, const void* v_slab_, int64_t stride_b, int64_t stride_h,
    const int32_t* __restrict__ kv_len, int extra, float* __restrict__ ws, void* __restrict__ out_, int64_t out_row_stride, int n_rep,
    float scale, const void* __restrict__ cos_, const void* __restrict__ sin_, int n_pos, const int32_t* __restrict__ pos_base, int T_cap,
    int n_kv_heads, int chunk_keys) {
  constexpr int V = Elem<T>::kVec;
  constexpr int LPK = D / V;          // lanes per key
  constexpr int KPW = 64 / LPK;       // keys per wave per load instruction
  constexpr int NG = NW * KPW;        // lane groups per workgroup
  using S = typename Elem<T>::storage;
  __shared__ float sm_m[NG], sm_l[NG];
  __shared__ float sm_o[NG][D];

  DL_STAMP(0, false);
  const int split = blockIdx.x, n_splits = gridDim.x, h = blockIdx.y, b = blockIdx.z;
  const int n_heads = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane / LPK, c = (lane % LPK) * V;
  const int kvh = h / n_rep;
  const S* row = reinterpret_cast<const S*>(q_) + (int64_t)b * q_row_stride;
  const S* kb = reinterpret_cast<const S*>(k_slab_) + (int64_t)b * stride_b + (int64_t)kvh * stride_h + c;
  const S* vb = reinterpret_cast<const S*>(v_slab_) + (int64_t)b * stride_b + (int64_t)kvh * stride_h + c;
  constexpr int HALF = D / 2;
  const int cpar = c < HALF ? c + HALF : c - HALF;

  // ---- every load that does not depend on another load is issued up front, K/V first: a dependent HBM round trip costs
  // ~1.5 us here, so the kernel's latency is (number of round trips), not bytes.
  // With a host-provided chunk (chunk_keys > 0, needs T_cap) the key range of this split does not depend on kv_len either: the
  // K/V rows are requested speculatively (any slot < T_cap is readable) and masked once kv_len[b] has arrived. ----
  const int T_old = kv_len[b];
  const bool spec = chunk_keys > 0 && T_cap > 0;
  uint4 kraw[U], vraw[U];
  int chunk = 0, k0 = 0;
  if (spec) {
    chunk = (chunk_keys + NG - 1) / NG * NG;
    k0 = split * chunk;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = min(k0 + (u * NW + wid) * KPW + g, T_cap - 1);
      kraw[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * D);
      vraw[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * D);
    }
  }
  const int Tn = T_old + (FUSED ? 1 : extra);
  if (!spec) {
    chunk = (Tn + n_splits - 1) / n_splits;
    chunk = (chunk + NG - 1) / NG * NG;
    k0 = split * chunk;
  }
  // this split's keys [k0, k1s); with a host chunk the last split also takes whatever the host's length bound missed
  const int k1s = (spec && split == n_splits - 1) ? Tn : min(Tn, k0 + chunk);
  const int k1 = FUSED ? min(k1s, T_old) : k1s;      // ... of which [k0, k1) are read from the slab
  bool ok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int key = k0 + (u * NW + wid) * KPW + g;
    ok[u] = key < k1;
    if (!spec) {
      const int64_t off = (int64_t)(ok[u] ? key : (k0 < k1 ? k0 : 0)) * D;
      kraw[u] = *reinterpret_cast<const uint4*>(kb + off);
      vraw[u] = *reinterpret_cast<const uint4*>(vb + off);
    }
  }
  float qv[V], cs[V], sn[V];
  const bool owns_new = FUSED && T_old >= k0 && T_old < k1s && wid == 0 && g == 0;
  float kn[V], vn[V];
  if constexpr (FUSED) {
    int p = pos_base[b];
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    float own[V], par[V], kown[V], kpar[V];
    load16<T>(reinterpret_cast<const S*>(cos_) + (int64_t)p * D + (c % HALF), cs);  // table = cat(freqs, freqs)
    load16<T>(reinterpret_cast<const S*>(sin_) + (int64_t)p * D + (c % HALF), sn);
    load16<T>(row + (int64_t)h * D + c, own);
    load16<T>(row + (int64_t)h * D + cpar, par);
    if (owns_new) {
      const S* krow = row + (int64_t)(n_heads + kvh) * D;
      load16<T>(krow + c, kown);
      load16<T>(krow + cpar, kpar);
      load16<T>(row + (int64_t)(n_heads + n_kv_heads + kvh) * D + c, vn);
    }
    if (c < HALF) rope16<T, false>(own, par, cs, sn, qv); else rope16<T, true>(own, par, cs, sn, qv);
    if (owns_new) {
      if (c < HALF) rope16<T, false>(kown, kpar, cs, sn, kn); else rope16<T, true>(kown, kpar, cs, sn, kn);
    }
  } else {
    load16<T>(row + (int64_t)h * D + c, qv);
  }

  DL_STAMP(1, true);  // every up-front load has landed
  float m = -INFINITY, l = 0.f, o[V];
#pragma unroll
  for (int i = 0; i < V; ++i) o[i] = 0.f;

  // keys of this workgroup are dealt round-robin: key = base + (u * NW + wid) * KPW + g
  for (int base = k0; base < k1; base += NG * U) {
    if (base != k0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int key = base + (u * NW + wid) * KPW + g;
        ok[u] = key < k1;
        const int64_t off = (int64_t)(ok[u] ? key : k0) * D;
        kraw[u] = *reinterpret_cast<const uint4*>(kb + off);
        vraw[u] = *reinterpret_cast<const uint4*>(vb + off);
      }
    }
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kx[V];
      unpack_kv<T>(kraw[u], kx);
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) a += qv[i] * kx[i];
      a = lpk_sum<LPK>(a);
      s[u] = ok[u] ? a * scale : -INFINITY;
    }
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) mn = fmaxf(mn, s[u]);
    if (mn > -INFINITY) {
      const float alpha = __expf(m - mn);  // m = -inf -> 0
      l *= alpha;
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] *= alpha;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float vx[V];
        unpack_kv<T>(vraw[u], vx);
        const float p = __expf(s[u] - mn);  // masked key: exp(-inf) = 0
        l += p;
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] += ok[u] ? p * vx[i] : 0.f;  // a speculatively read slot past the length may hold NaN bits
      }
      m = mn;
    }
  }

  if constexpr (FUSED) {
    // the new token (key index T_old): owned by lane group (wave 0, g 0) of the split whose range contains it
    if (owns_new) {
      if (h % n_rep == 0 && T_old < T_cap) {  // one writer per kv head; eviction = the length is simply not advanced later
        S* kd = const_cast<S*>(kb) + (int64_t)T_old * D;
        S* vd = const_cast<S*>(vb) + (int64_t)T_old * D;
        store16<T>(kd, kn);
        store16<T>(vd, vn);
      }
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) a += qv[i] * kn[i];
      a = lpk_sum<LPK>(a);
      const float sc_ = a * scale;
      const float mn = fmaxf(m, sc_);
      const float alpha = __expf(m - mn);
      const float p = __expf(sc_ - mn);
      l = l * alpha + p;
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = o[i] * alpha + p * vn[i];
      m = mn;
    }
  }

  DL_STAMP(2, true);  // scores / softmax / PV (and the appended token) done
  // merge the NG lane groups of this workgroup
  const int gg = wid * KPW + g;
  if ((lane % LPK) == 0) {
    sm_m[gg] = m;
    sm_l[gg] = l;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) sm_o[gg][c + i] = o[i];
  __syncthreads();
  if (tid < D) {
    // NG <= 32 partials: every LDS read is issued before the first use (a rolled loop pays the LDS latency NG times over)
    float mg[NG], lg[NG], og[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      mg[i] = sm_m[i];
      lg[i] = sm_l[i];
      og[i] = sm_o[i][tid];
    }
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < NG; ++i) M = fmaxf(M, mg[i]);
    float L = 0.f, O = 0.f;
    if (M > -INFINITY) {
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const float w = __expf(mg[i] - M);  // empty group: exp(-inf) = 0
        L += lg[i] * w;
        O += og[i] * w;
      }
    }
    DL_STAMP(3, false);  // workgroup merge done
    if (n_splits == 1) {
      store1<T>(out_, (int64_t)b * out_row_stride + (int64_t)h * D + tid, L > 0.f ? O / L : 0.f);
    } else {
      float* p = ws + (((int64_t)b * n_heads + h) * n_splits + split) * (D + kAttnPartPad);
      p[kAttnPartPad + tid] = O;
      if (tid == 0) {
        p[0] = M;
        p[1] = L;
      }
    }
  }
  DL_STAMP(4, true);  // stores acknowledged
}

constexpr int kMaxSplits = 128;

// Merge the split partials of one (row, head).  The partials were written by other CUs a microsecond ago, so every dependent
// load is a full fabric round trip: all (m, l, o) values of up to 8 splits are requested at once (attn_split_merge).
template <typename T, int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const float* __restrict__ ws, void* __restrict__ out_,
                                                                int64_t out_row_stride, int n_splits) {
  const int h = blockIdx.x, b = blockIdx.y, n_heads = gridDim.x, d = threadIdx.x;
  float o[1];
  attn_split_merge<1>(ws + ((int64_t)b * n_heads + h) * n_splits * (D + kAttnPartPad), n_splits, D, d, o);
  store1<T>(out_, (int64_t)b * out_row_stride + (int64_t)h * D + d, o[0]);
}

template <typename T, int D, int NW, bool FUSED, int U>
static void launch_split(const void* q, int64_t q_row_stride, const void* k_slab, const void* v_slab, int64_t stride_b, int64_t stride_h,
                         const int32_t* kv_len, int extra, void* out, int64_t out_row_stride, void* workspace, int n_splits, int B,
                         int n_heads, int n_kv_heads, const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* pos_base,
                         int T_cap, int chunk_keys, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)D);
  hipLaunchKernelGGL((attn_decode_split_kernel<T, D, NW, FUSED, U>), dim3((unsigned)n_splits, (unsigned)n_heads, (unsigned)B), dim3(NW * 64), 0,
                     st, q, q_row_stride, k_slab, v_slab, stride_b, stride_h, kv_len, extra, reinterpret_cast<float*>(workspace), out,
                     out_row_stride, n_heads / n_kv_heads, scale, cos_tab, sin_tab, n_pos, pos_base, T_cap, n_kv_heads, chunk_keys);
  if (n_splits > 1)
    hipLaunchKernelGGL((attn_decode_combine_kernel<T, D>), dim3((unsigned)n_heads, (unsigned)B), dim3(D), 0, st,
                       reinterpret_cast<const float*>(workspace), out, out_row_stride, n_splits);
}

}  // namespace dl

using namespace dl;

extern "C" int64_t dl_attn_decode_workspace_bytes(int B, int n_heads, int head_dim, int n_splits) {
  if (n_splits <= 1) return 0;
  return (int64_t)B * n_heads * n_splits * (head_dim + kAttnPartPad) * (int64_t)sizeof(float);
}

extern "C" int dl_attn_decode(const void* q, int64_t q_row_stride, const void* k_slab, const void* v_slab, int64_t slab_stride_b,
                              int64_t slab_stride_h, const int32_t* kv_len, int extra, void* out, int64_t out_row_stride,
                              void* workspace, int n_splits, int B, int n_heads, int n_kv_heads, int head_dim, int dtype,
                              void* stream) {
  DL_REQUIRE(q && k_slab && v_slab && kv_len && out, "dl_attn_decode: NULL pointer");
  DL_REQUIRE(B > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0, "dl_attn_decode: bad head counts");
  DL_REQUIRE(n_splits >= 1 && n_splits <= kMaxSplits, "dl_attn_decode: n_splits=%d must be in [1, %d]", n_splits, kMaxSplits);
  DL_REQUIRE(n_splits == 1 || workspace, "dl_attn_decode: workspace required when n_splits > 1");
  DL_REQUIRE(head_dim == 128 || head_dim == 64, "dl_attn_decode: head_dim=%d unsupported (64 or 128)", head_dim);
  hipStream_t st = as_stream(stream);
  DL_DISPATCH_DTYPE(dtype, T, {
    if (head_dim == 128)
      launch_split<T, 128, 4, false, 4>(q, q_row_stride, k_slab, v_slab, slab_stride_b, slab_stride_h, kv_len, extra, out, out_row_stride,
                                     workspace, n_splits, B, n_heads, n_kv_heads, nullptr, nullptr, 0, nullptr, 0, 0, st);
    else
      launch_split<T, 64, 4, false, 4>(q, q_row_stride, k_slab, v_slab, slab_stride_b, slab_stride_h, kv_len, extra, out, out_row_stride,
                                    workspace, n_splits, B, n_heads, n_kv_heads, nullptr, nullptr, 0, nullptr, 0, 0, st);
  });
  DL_CHECK_LAUNCH("dl_attn_decode");
  return DL_OK;
}

extern "C" int dl_attn_decode_rope(const void* qkv, int64_t qkv_row_stride, const void* cos_tab, const void* sin_tab, int n_pos,
                                   const int32_t* pos_base, const int32_t* kv_len, void* k_slab, void* v_slab, int64_t slab_stride_b,
                                   int64_t slab_stride_h, int T_cap, void* out, int64_t out_row_stride, void* workspace, int n_splits,
                                   int keys_in_flight, int chunk_keys, int B, int n_heads, int n_kv_heads, int head_dim, int dtype,
                                   void* stream) {
  DL_REQUIRE(keys_in_flight == 64 || keys_in_flight == 256, "dl_attn_decode_rope: keys_in_flight must be 64 or 256");
  DL_REQUIRE(chunk_keys >= 0, "dl_attn_decode_rope: chunk_keys must be >= 0");
  DL_REQUIRE(qkv && cos_tab && sin_tab && pos_base && kv_len && k_slab && v_slab && out, "dl_attn_decode_rope: NULL pointer");
  DL_REQUIRE(B > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && n_pos > 0 && T_cap > 0, "dl_attn_decode_rope: bad shape");
  DL_REQUIRE(n_splits >= 1 && n_splits <= kMaxSplits, "dl_attn_decode_rope: n_splits=%d must be in [1, %d]", n_splits, kMaxSplits);
  DL_REQUIRE(n_splits == 1 || workspace, "dl_attn_decode_rope: workspace required when n_splits > 1");
  DL_REQUIRE(head_dim == 128 || head_dim == 64, "dl_attn_decode_rope: head_dim=%d unsupported (64 or 128)", head_dim);
  hipStream_t st = as_stream(stream);
  // a 16-byte-per-lane row covers head_dim with D/kVec lanes, so one wave-wide load is 64/(D/kVec) keys: U is chosen so that
  // NW * keys-per-load * U = keys_in_flight for the 16-bit dtypes at head_dim 128 (the production shape)
#define DL_FUSED_ARGS qkv, qkv_row_stride, k_slab, v_slab, slab_stride_b, slab_stride_h, kv_len, 1, out, out_row_stride, workspace, n_splits, B, n_heads, n_kv_heads, cos_tab, sin_tab, n_pos, pos_base, T_cap, chunk_keys, st
  DL_DISPATCH_DTYPE(dtype, T, {
    if (head_dim == 128) {
      if (keys_in_flight == 256) launch_split<T, 128, 4, true, 16>(DL_FUSED_ARGS); else launch_split<T, 128, 4, true, 4>(DL_FUSED_ARGS);
    } else {
      if (keys_in_flight == 256) launch_split<T, 64, 4, true, 16>(DL_FUSED_ARGS); else launch_split<T, 64, 4, true, 4>(DL_FUSED_ARGS);
    }
  });
#undef DL_FUSED_ARGS
  DL_CHECK_LAUNCH("dl_attn_decode_rope");
  return DL_OK;
}
