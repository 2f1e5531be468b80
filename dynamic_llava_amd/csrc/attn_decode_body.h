// Body of the split-KV decode attention (F9 decode + F11: RoPE of q / new key, KV-slab append, ragged attention over an evicted
// slab), shared by the stand-alone launch (attn_decode.hip: one 256-thread workgroup per (split, head, row)) and by the attention workgroups
// of the fused batch-1 q|k|v launch (gemv.hip, dl_gemv_qkv_attn: 1..4 workgroups per head fed their q / k / v rows as granules).  Sharing the
// code keeps the two paths in one arithmetic: same key -> lane-group dealing, same online-softmax batching, same merge orders (the fused
// launch folds the new token in after the slab merge -- attn_split_finish_newlast -- so its output is in the same rounding class, not the
// same bits; projection row, residual stream and appended K/V are bit-identical).
//
// Two parts, so that a caller can have the K/V rows in flight before the query exists:
//   attn_split_issue  -- needs only kv_len / the slab: key range of the split, first K/V rows requested
//   attn_split_finish -- needs the q (and, for the split that owns the new token, k / v) rows: RoPE, scores, online softmax,
//                        P.V, the appended token, merge of the workgroup's lane groups -> (M, L, O[d]) for thread d < D
#pragma once
#include "dl_common.h"

namespace dl {

// tools/qa_timing.hip compiles gemv.hip with -DDL_QA_TIMING: per-workgroup wall-clock stamps of dl_gemv_qkv_attn (100 MHz)
#ifdef DL_QA_TIMING
__device__ long long g_qa_stamps[1200][8];
#define DL_QSTAMP(i)                                                              \
  do {                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 1200) g_qa_stamps[blockIdx.x][i] = wall_clock64(); \
  } while (0)
#else
#define DL_QSTAMP(i)
#endif


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

// K/V rows are read once per decode step: non-temporal (the weight streams' policy), so that they do not displace what the step re-reads
__device__ __forceinline__ uint4 kv_ld16(const void* p) {
  typedef uint32_t kv_u32x4_t __attribute__((ext_vector_type(4)));
  const kv_u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const kv_u32x4_t*>(p));
  return make_uint4(r.x, r.y, r.z, r.w);
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

// Geometry of one split: D/kVec lanes cooperate on one key (16 lanes x 16 B = one 256-byte K row for D=128 bf16), so a wave-wide load
// is KPW full rows; NW waves = NG lane groups; U key rows per lane group are requested per loop trip.
template <typename T, int D, int NW, int U>
struct AttnSplitState {
  static constexpr int V = Elem<T>::kVec;
  static constexpr int LPK = D / V;
  static constexpr int KPW = 64 / LPK;
  static constexpr int NG = NW * KPW;
  using S = typename Elem<T>::storage;
  const S* kb;
  const S* vb;
  int T_old, Tn, chunk, k0, k1s, k1;
  int wid, lane, g, c;  // wave inside the (virtual) workgroup, lane, lane group inside the wave, first head dim of this lane
  bool spec;
  bool ok[U];
  uint4 kraw[U], vraw[U];
  // second trip, requested ahead of the first one's consumption (attn_split_prefetch2: callers that have to wait for q anyway)
  bool pre2;
  bool ok2[U <= 4 ? U : 1];
  uint4 kraw2[U <= 4 ? U : 1], vraw2[U <= 4 ? U : 1];
};

// Part 1.  `vtid`: thread index inside the (virtual) workgroup of NW waves.  Every load that does not depend on another load is
// issued here, K/V first: a dependent HBM round trip costs ~1.5 us, so the latency is (number of round trips), not bytes.  With a
// host-provided chunk (chunk_keys > 0, needs T_cap) the key range does not depend on kv_len either: the rows are requested
// speculatively (any slot < T_cap is readable) and masked once kv_len[b] has arrived.
template <typename T, int D, int NW, bool FUSED, int U>
__device__ __forceinline__ void attn_split_issue(AttnSplitState<T, D, NW, U>& s, int vtid, const void* k_slab_, const void* v_slab_,
                                                 int64_t stride_b, int64_t stride_h, int T_old, int extra, int b, int kvh, int split,
                                                 int n_splits, int T_cap, int chunk_keys) {
  using St = AttnSplitState<T, D, NW, U>;
  using S = typename St::S;
  constexpr int V = St::V, LPK = St::LPK, KPW = St::KPW, NG = St::NG;
  s.lane = vtid & 63;
  s.wid = vtid >> 6;
  s.g = s.lane / LPK;
  s.c = (s.lane % LPK) * V;
  s.kb = reinterpret_cast<const S*>(k_slab_) + (int64_t)b * stride_b + (int64_t)kvh * stride_h + s.c;
  s.vb = reinterpret_cast<const S*>(v_slab_) + (int64_t)b * stride_b + (int64_t)kvh * stride_h + s.c;
  s.T_old = T_old;
  s.pre2 = false;
  s.spec = chunk_keys > 0 && T_cap > 0;
  s.chunk = 0;
  s.k0 = 0;
  if (s.spec) {
    s.chunk = (chunk_keys + NG - 1) / NG * NG;
    s.k0 = split * s.chunk;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = min(s.k0 + (u * NW + s.wid) * KPW + s.g, T_cap - 1);
      s.kraw[u] = kv_ld16(s.kb + (int64_t)key * D);
      s.vraw[u] = kv_ld16(s.vb + (int64_t)key * D);
    }
  }
  s.Tn = T_old + (FUSED ? 1 : extra);
  if (!s.spec) {
    s.chunk = (s.Tn + n_splits - 1) / n_splits;
    s.chunk = (s.chunk + NG - 1) / NG * NG;
    s.k0 = split * s.chunk;
  }
  // this split's keys [k0, k1s); with a host chunk the last split also takes whatever the host's length bound missed
  s.k1s = (s.spec && split == n_splits - 1) ? s.Tn : min(s.Tn, s.k0 + s.chunk);
  s.k1 = FUSED ? min(s.k1s, T_old) : s.k1s;  // ... of which [k0, k1) are read from the slab
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int key = s.k0 + (u * NW + s.wid) * KPW + s.g;
    s.ok[u] = key < s.k1;
    if (!s.spec) {
      const int64_t off = (int64_t)(s.ok[u] ? key : (s.k0 < s.k1 ? s.k0 : 0)) * D;
      s.kraw[u] = kv_ld16(s.kb + off);
      s.vraw[u] = kv_ld16(s.vb + off);
    }
  }
}

// Optional, between the two parts: request the SECOND trip's rows too (U <= 4).  attn_split_finish requests trip i + 1 when it starts on trip i;
// a caller that cannot start yet (dl_gemv_qkv_attn waits for q) puts two trips in flight meanwhile.  Loads only: results unchanged.
template <typename T, int D, int NW, int U>
__device__ __forceinline__ void attn_split_prefetch2(AttnSplitState<T, D, NW, U>& s) {
  using St = AttnSplitState<T, D, NW, U>;
  constexpr int KPW = St::KPW, NG = St::NG;
  if constexpr (U <= 4) {
    const int nbase = s.k0 + NG * U;
    if (nbase < s.k1) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int key = nbase + (u * NW + s.wid) * KPW + s.g;
        s.ok2[u] = key < s.k1;
        const int64_t off = (int64_t)(s.ok2[u] ? key : s.k0) * D;
        s.kraw2[u] = kv_ld16(s.kb + off);
        s.vraw2[u] = kv_ld16(s.vb + off);
      }
      s.pre2 = true;
    }
  }
}

// ---- the two halves of attn_split_finish that the "new key last" variant below shares with it (same code, same order, same bits) ----
// Online softmax over this split's slab keys [k0, k1): per lane group (m, l, o[V]).
template <typename T, int D, int NW, int U>
__device__ __forceinline__ void attn_split_keys(AttnSplitState<T, D, NW, U>& s, const float (&qv)[Elem<T>::kVec], float scale, float& m, float& l,
                                                float (&o)[Elem<T>::kVec]) {
  using St = AttnSplitState<T, D, NW, U>;
  constexpr int V = St::V, LPK = St::LPK, KPW = St::KPW, NG = St::NG;
  const int g = s.g, wid = s.wid;
  m = -INFINITY;
  l = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) o[i] = 0.f;

  // keys of this workgroup are dealt round-robin: key = base + (u * NW + wid) * KPW + g
  // the rows of trip i + 1 are requested before trip i is consumed (U <= 4: the registers are there): a row of T keys is a chain of
  // T / (NG U) trips, and without this every trip paid a full memory round trip (the longest row of a ragged batch sets the launch time)
  constexpr bool kPrefetch = U <= 4;
  uint4 kpre[kPrefetch ? U : 1], vpre[kPrefetch ? U : 1];
  bool okpre[kPrefetch ? U : 1];
  for (int base = s.k0; base < s.k1; base += NG * U) {
    if constexpr (kPrefetch) {
      const int nbase = base + NG * U;
      if (nbase < s.k1) {
        if (base == s.k0 && s.pre2) {  // already in flight (attn_split_prefetch2)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            okpre[u] = s.ok2[u];
            kpre[u] = s.kraw2[u];
            vpre[u] = s.vraw2[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int key = nbase + (u * NW + wid) * KPW + g;
            okpre[u] = key < s.k1;
            const int64_t off = (int64_t)(okpre[u] ? key : s.k0) * D;
            kpre[u] = kv_ld16(s.kb + off);
            vpre[u] = kv_ld16(s.vb + off);
          }
        }
      }
    } else if (base != s.k0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int key = base + (u * NW + wid) * KPW + g;
        s.ok[u] = key < s.k1;
        const int64_t off = (int64_t)(s.ok[u] ? key : s.k0) * D;
        s.kraw[u] = kv_ld16(s.kb + off);
        s.vraw[u] = kv_ld16(s.vb + off);
      }
    }
    float sc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kx[V];
      unpack_kv<T>(s.kraw[u], kx);
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) a += qv[i] * kx[i];
      a = lpk_sum<LPK>(a);
      sc[u] = s.ok[u] ? a * scale : -INFINITY;
    }
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) mn = fmaxf(mn, sc[u]);
    if (mn > -INFINITY) {
      const float alpha = __expf(m - mn);  // m = -inf -> 0
      l *= alpha;
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] *= alpha;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float vx[V];
        unpack_kv<T>(s.vraw[u], vx);
        const float p = __expf(sc[u] - mn);  // masked key: exp(-inf) = 0
        l += p;
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] += s.ok[u] ? p * vx[i] : 0.f;  // a speculatively read slot past the length may hold NaN bits
      }
      m = mn;
    }
    if constexpr (kPrefetch) {
      if (base + NG * U < s.k1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          s.kraw[u] = kpre[u];
          s.vraw[u] = vpre[u];
          s.ok[u] = okpre[u];
        }
      }
    }
  }
}

// Merge of the NG lane groups' (m, l, o) through LDS -> (M, L, O) for threads vtid < D.  Contains ONE __syncthreads().
template <typename T, int D, int NW, int U>
__device__ __forceinline__ void attn_split_lds_merge(const AttnSplitState<T, D, NW, U>& s, int vtid, float m, float l, const float (&o)[Elem<T>::kVec], float* sm_m,
                                                     float* sm_l, float* sm_o, float& M_out, float& L_out, float& O_out) {
  using St = AttnSplitState<T, D, NW, U>;
  constexpr int V = St::V, LPK = St::LPK, KPW = St::KPW, NG = St::NG;
  const int c = s.c, g = s.g, wid = s.wid, lane = s.lane;
  const int gg = wid * KPW + g;
  if ((lane % LPK) == 0) {
    sm_m[gg] = m;
    sm_l[gg] = l;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) sm_o[gg * D + c + i] = o[i];
  __syncthreads();
  M_out = -INFINITY;
  L_out = 0.f;
  O_out = 0.f;
  if (vtid < D) {
    // NG <= 32 partials: every LDS read is issued before the first use (a rolled loop pays the LDS latency NG times over)
    float mg[NG], lg[NG], og[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      mg[i] = sm_m[i];
      lg[i] = sm_l[i];
      og[i] = sm_o[i * D + vtid];
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
    M_out = M;
    L_out = L;
    O_out = O;
  }
}

// A D-element row handed over either in the element type T (NP == 0) or as NP fp32 partial sums `part_stride` floats apart (dl_linear_packed's LP_EPI_PARTS
// output: the projection's k ranges): added in range order and rounded to T once -- the value the projection's own store epilogue would have written.  NP is a
// compile-time constant: straight-line code, every range's loads in flight together (a runtime loop or branch here makes hipcc wait for ALL outstanding loads
// at the join -- the slab rows attn_split_issue requested before -- which cost the 1024-workgroup launch 2.7 us when it was tried).
template <typename T, int NP>
__device__ __forceinline__ void load16_row(const void* row_, int c, int64_t part_stride, float (&f)[Elem<T>::kVec]) {
  if constexpr (NP == 0) {
    load16<T>(reinterpret_cast<const typename Elem<T>::storage*>(row_) + c, f);
  } else {
    constexpr int V = Elem<T>::kVec;
    const float* a = reinterpret_cast<const float*>(row_) + c;
    float4 x[NP][V / 4];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int i = 0; i < V / 4; ++i) x[r][i] = *reinterpret_cast<const float4*>(a + (int64_t)r * part_stride + 4 * i);
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
      float4 t = x[0][i];
#pragma unroll
      for (int r = 1; r < NP; ++r) t.x += x[r][i].x, t.y += x[r][i].y, t.z += x[r][i].z, t.w += x[r][i].w;
      f[4 * i] = Elem<T>::round(t.x), f[4 * i + 1] = Elem<T>::round(t.y), f[4 * i + 2] = Elem<T>::round(t.z), f[4 * i + 3] = Elem<T>::round(t.w);
    }
  }
}

// Part 2.  qrow / krow / vrow: the D-element q, k, v vectors of this head (un-rotated when FUSED; any address space).  cos_/sin_:
// RoPE tables [n_pos, D]; pos: the new token's position.  sm_m/sm_l [NG], sm_o [NG][D]: LDS scratch of this (virtual) workgroup.
// `write_kv`: this workgroup stores the new token's rotated key / value at slab slot T_old (one writer per kv head).
// Contains ONE __syncthreads(): every wave of the real workgroup must call it.  Result for threads vtid < D: M, L (same for all)
// and O = un-normalised output of head dim vtid.
// `before_new` (default: nothing): called by EVERY thread right before the new token's key / value rows are read -- a caller whose k / v rows
// arrive late (dl_gemv_qkv_attn: they are the last outputs of the projection running in the same launch) waits for them there, after the
// scores / softmax / P.V over the slab keys, which need q only.  With a waiter the k / v loads and the key's RoPE move behind the call; the
// arithmetic and its order are the same.
struct AttnNoWait {
  __device__ __forceinline__ void operator()() const {}
};
template <typename T, int D, int NW, bool FUSED, int U, typename BeforeNew = AttnNoWait, int NP = 0>
__device__ __forceinline__ void attn_split_finish(AttnSplitState<T, D, NW, U>& s, int vtid, const void* qrow_, const void* krow_,
                                                  const void* vrow_, const void* cos_, const void* sin_, int n_pos, int pos, float scale,
                                                  bool write_kv, int T_cap, float* sm_m, float* sm_l, float* sm_o, float& M_out, float& L_out,
                                                  float& O_out, BeforeNew before_new = BeforeNew(), int64_t part_stride = 0) {
  constexpr bool LATE = !__is_same(BeforeNew, AttnNoWait);
  using St = AttnSplitState<T, D, NW, U>;
  using S = typename St::S;
  constexpr int V = St::V, LPK = St::LPK, KPW = St::KPW, NG = St::NG;
  constexpr int HALF = D / 2;
  const int c = s.c, g = s.g, wid = s.wid, lane = s.lane;
  const int cpar = c < HALF ? c + HALF : c - HALF;
  float qv[V], cs[V], sn[V];
  const bool owns_new = FUSED && s.T_old >= s.k0 && s.T_old < s.k1s && wid == 0 && g == 0;
  float kn[V], vn[V];
  if constexpr (FUSED) {
    int p = pos;
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    float own[V], par[V], kown[V], kpar[V];
    load16<T>(reinterpret_cast<const S*>(cos_) + (int64_t)p * D + (c % HALF), cs);  // table = cat(freqs, freqs)
    load16<T>(reinterpret_cast<const S*>(sin_) + (int64_t)p * D + (c % HALF), sn);
    load16_row<T, NP>(qrow_, c, part_stride, own);
    if (owns_new && !LATE) {
      load16_row<T, NP>(krow_, c, part_stride, kown);
      if constexpr (NP == 0) load16_row<T, NP>(krow_, cpar, part_stride, kpar);
      load16_row<T, NP>(vrow_, c, part_stride, vn);
    }
    // q's RoPE partner half (column c +- D/2) is what the lane LPK / 2 lanes away in this lane group (LPK = D / kVec lanes share a row) has just loaded (partial sums: summed and rounded): take it
    // from there instead of requesting it a second time -- half the requests in front of the first score (round 6: 1024-workgroup launches, 32 rows)
#pragma unroll
    for (int i = 0; i < V; ++i) par[i] = __shfl_xor(own[i], LPK / 2);
    if constexpr (NP > 0 && !LATE) {  // the new key's partner half the same way (every lane takes part in the exchange; only the owning lane group's values are used)
#pragma unroll
      for (int i = 0; i < V; ++i) kpar[i] = __shfl_xor(owns_new ? kown[i] : 0.f, LPK / 2);
    }
    if (c < HALF) rope16<T, false>(own, par, cs, sn, qv); else rope16<T, true>(own, par, cs, sn, qv);
    if (owns_new && !LATE) {
      if (c < HALF) rope16<T, false>(kown, kpar, cs, sn, kn); else rope16<T, true>(kown, kpar, cs, sn, kn);
    }
  } else {
    load16<T>(reinterpret_cast<const S*>(qrow_) + c, qv);
  }

  float m, l, o[V];
  attn_split_keys<T, D, NW, U>(s, qv, scale, m, l, o);

  if constexpr (FUSED && LATE) {
    before_new();
    if (owns_new) {
      float kown[V], kpar[V];
      load16_row<T, NP>(krow_, c, part_stride, kown);
      load16_row<T, NP>(krow_, cpar, part_stride, kpar);
      load16_row<T, NP>(vrow_, c, part_stride, vn);
      if (c < HALF) rope16<T, false>(kown, kpar, cs, sn, kn); else rope16<T, true>(kown, kpar, cs, sn, kn);
    }
  }
  if constexpr (FUSED) {
    // the new token (key index T_old): owned by lane group (wave 0, g 0) of the split whose range contains it
    if (owns_new) {
      if (write_kv && s.T_old < T_cap) {  // one writer per kv head; eviction = the length is simply not advanced later
        S* kd = const_cast<S*>(s.kb) + (int64_t)s.T_old * D;
        S* vd = const_cast<S*>(s.vb) + (int64_t)s.T_old * D;
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

  // merge the NG lane groups of this (virtual) workgroup
  attn_split_lds_merge<T, D, NW, U>(s, vtid, m, l, o, sm_m, sm_l, sm_o, M_out, L_out, O_out);
}

// "New key last" (round 4, dl_gemv_qkv_attn only): the caller's k / v rows are the LAST outputs of the projection that runs in the same launch, so
// whatever happens after they arrive is the launch's tail.  attn_split_finish folds the new token into lane group 0's partial and THEN merges the
// lane groups through LDS (barrier + NG-way merge behind the wait).  Here the slab keys' partials are merged FIRST -- while the weights still stream --
// and after the wait the D threads that hold (M, L, O[d]) fold the new token in themselves: RoPE of their own element of k, one 128-term dot product
// (wave reduction + one LDS word per wave), one softmax update, the slab append, the normalised output.  Same mathematics; the new token's term is
// added after the slab merge instead of before it, so the last bits may differ from attn_split_finish (kernel tests: noise class, not bits).
//   rows : LDS, the raw (un-rotated) q vector of this head, D elements; q_rot_lds: LDS, D elements (receives the rotated query).
//   fetch_kv(d, dpar, k_own, k_par, v): called by the threads vtid < D after the slab merge -- returns the raw k[d], k[dpar], v[d] of the new token
//          (dl_gemv_qkv_attn polls the projection's granules there: three requests per thread in one round trip, no LDS staging, no barrier).
//   red  : LDS scratch, >= NW floats.   out: the head's attention output for threads vtid < D.   Contains THREE __syncthreads().
template <typename T, int D, int NW, int U>
__device__ __forceinline__ void attn_split_pin_prefetched(AttnSplitState<T, D, NW, U>& s) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    pin_reg(s.kraw[u]);
    pin_reg(s.vraw[u]);
  }
  if constexpr (U <= 4) {
    if (s.pre2) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pin_reg(s.kraw2[u]);
        pin_reg(s.vraw2[u]);
      }
    }
  }
}

// The RoPE table entries of the new token's position, requested BEFORE the caller waits for q (a cold table row is an HBM + TLB round trip of 2-3 us:
// inside attn_split_finish_newlast it sat on the launch's critical path, right after q's arrival -- tools/qa_timing.hip)
template <typename T>
struct AttnRopeRow {
  float cs1, sn1;  // cos / sin of the finishing thread's own element
};
template <typename T, int D, int NW, int U>
__device__ __forceinline__ void attn_newlast_preload(const AttnSplitState<T, D, NW, U>& s, int vtid, const void* cos_, const void* sin_, int n_pos, int pos,
                                                     AttnRopeRow<T>& r) {
  using S = typename Elem<T>::storage;
  constexpr int HALF = D / 2;
  int p = pos;
  p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
  const S* cos_row = reinterpret_cast<const S*>(cos_) + (int64_t)p * D;  // table = cat(freqs, freqs)
  const S* sin_row = reinterpret_cast<const S*>(sin_) + (int64_t)p * D;
  const int d = vtid < D ? vtid : 0;
  r.cs1 = Elem<T>::to_f(cos_row[d % HALF]);
  r.sn1 = Elem<T>::to_f(sin_row[d % HALF]);
  pin_reg(r.cs1);
  pin_reg(r.sn1);
}

//   after_slab(M, L, O) -> bool: called by every thread once the slab keys of THIS workgroup are merged ((M, L, O) valid for vtid < D).  A workgroup
//          that holds only a PART of the head's keys (several workgroups per head) publishes its partial there and returns false (the function
//          returns at once); the head's primary workgroup folds the other parts into (M, L, O) there and returns true.
struct AttnSlabWhole {
  __device__ __forceinline__ bool operator()(float&, float&, float&) const { return true; }
};
template <typename T, int D, int NW, int U, typename FetchKV, typename AfterSlab = AttnSlabWhole>
__device__ __forceinline__ void attn_split_finish_newlast(AttnSplitState<T, D, NW, U>& s, int vtid, const typename Elem<T>::storage* rows, const AttnRopeRow<T>& rope,
                                                          typename Elem<T>::storage* q_rot_lds, float scale, bool write_kv, int T_cap, float* sm_m, float* sm_l,
                                                          float* sm_o, float* red, float& out, FetchKV fetch_kv, AfterSlab after_slab = AfterSlab()) {
  using St = AttnSplitState<T, D, NW, U>;
  using S = typename St::S;
  constexpr int V = St::V;
  constexpr int HALF = D / 2;
  static_assert(D % 64 == 0 && D / 64 <= NW, "the D finishing threads are whole waves of this workgroup");
  const int c = s.c;
  // the rotated query: each of the D finishing threads rotates ITS element (DML:283-284, the roundings of rope16) and publishes it in LDS; every lane
  // then reads its 16-byte slice.  (Rotating the slices lane by lane needed the table row in 16 more registers per lane, held across the wait for q --
  // registers this kernel does not have: it must stay within 128 to keep the grid resident.)
  const int d = vtid < D ? vtid : 0, dpar = d < HALF ? d + HALF : d - HALF;
  const float cs1 = rope.cs1, sn1 = rope.sn1;
  const float q_own = Elem<T>::to_f(rows[d]), q_par = Elem<T>::to_f(rows[dpar]);
  const float q_rot = Elem<T>::round(Elem<T>::round(q_own * cs1) + Elem<T>::round((d < HALF ? -q_par : q_par) * sn1));
  if (vtid < D) q_rot_lds[d] = Elem<T>::from_f(q_rot);
  __syncthreads();
  float qv[V];
  load16<T>(q_rot_lds + c, qv);
  DL_QSTAMP(4);  // q rotated
  float m, l, o[V];
  attn_split_keys<T, D, NW, U>(s, qv, scale, m, l, o);
  DL_QSTAMP(5);  // slab keys done (this wave)
  float M, L, O;
  attn_split_lds_merge<T, D, NW, U>(s, vtid, m, l, o, sm_m, sm_l, sm_o, M, L, O);
  DL_QSTAMP(2);  // slab keys merged
  out = 0.f;
  if (!after_slab(M, L, O)) return;
  float part = 0.f, k_rot = 0.f, vv = 0.f;
  S v_raw = S();
  if (vtid < D) {
    S k_own_raw, k_par_raw;
    fetch_kv(d, dpar, k_own_raw, k_par_raw, v_raw);
    const float k_own = Elem<T>::to_f(k_own_raw), k_par = Elem<T>::to_f(k_par_raw);
    k_rot = Elem<T>::round(Elem<T>::round(k_own * cs1) + Elem<T>::round((d < HALF ? -k_par : k_par) * sn1));
    vv = Elem<T>::to_f(v_raw);
    part = q_rot * k_rot;
  }
  part = wave_sum(part);
  if (s.lane == 0) red[s.wid] = part;
  __syncthreads();
  out = 0.f;
  if (vtid < D) {
    float dot = 0.f;
#pragma unroll
    for (int w = 0; w < D / 64; ++w) dot += red[w];
    const float sc_ = dot * scale;
    const float mn = fmaxf(M, sc_);
    const float alpha = __expf(M - mn);  // empty slab: M = -inf -> 0
    const float pn = __expf(sc_ - mn);
    out = (O * alpha + pn * vv) / (L * alpha + pn);
    if (write_kv && s.T_old < T_cap) {  // one writer per kv head; eviction = the length is simply not advanced later
      S* kd = const_cast<S*>(s.kb) - c + (int64_t)s.T_old * D;  // (kb / vb carry this lane's column offset c)
      S* vd = const_cast<S*>(s.vb) - c + (int64_t)s.T_old * D;
      kd[d] = Elem<T>::from_f(k_rot);
      vd[d] = v_raw;
    }
  }
}

}  // namespace dl
