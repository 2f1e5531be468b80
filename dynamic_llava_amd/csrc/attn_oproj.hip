// Batch-1 decode: RoPE + KV append + split-KV attention (DML:260-285, CU:109-268, DML:1061-1122) AND the o_proj GEMV (DML:1127) in
// ONE launch.
//
// Why: at batch 1 the attention moves ~4-11 MB and takes ~9 us of pure latency during which HBM idles, and the o_proj launch behind it
// pays a boundary (~1.2 us) plus a cold start before its 33.5 MB stream.  Here the two run side by side: workgroups [0, n_att) are the
// attention (the same body and merge as dl_attn_decode_rope with the in-kernel combine: attn_decode_body.h), workgroups [n_att, ...)
// are the GEMV -- each wave requests its weight rows at once, so the whole of W_o streams WHILE the attention runs, then picks the
// attention output up from 8-byte {tag, value} granules (granule.h) as soon as split 0 of every head has published it.  Same
// arithmetic and summation order as dl_gemv's batch-1 plain kernel (gemv_dot.h) and as the two-launch attention: bit-identical.
// All workgroups must be resident (n_att + gemv workgroups <= 1024: checked by the host entry); waits are bounded (NaN on give-up).
#include "attn_decode_body.h"
#include "gemv_dot.h"
#include "granule.h"

namespace dl {

constexpr int kAoThreads = 256;

template <typename T, int XB>
__global__ __launch_bounds__(kAoThreads) void attn_decode_oproj_kernel(
    const void* __restrict__ qkv_, const void* k_slab_, const void* v_slab_, int64_t stride_h, const int32_t* __restrict__ kv_len,
    u64_t* __restrict__ ws, void* __restrict__ attn_out_, int n_rep, float scale, const void* __restrict__ cos_, const void* __restrict__ sin_,
    int n_pos, const int32_t* __restrict__ pos_base, int T_cap, int n_heads, int n_kv_heads, int n_splits, int call_tag,
    const void* __restrict__ Wo_, int N, int K, void* __restrict__ y_) {
  constexpr int D = 128, NW = 4, U = 4, V = 8;
  constexpr int PG = D + 2;  // granules of one split partial: M, L, O[D]
  using S = uint16_t;
  using St = AttnSplitState<T, D, NW, U>;
  constexpr int NG = St::NG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_att = n_splits * n_heads;
  const uint32_t tag = ((((uint32_t)pos_base[0] & 0x7fffffu) << 8) | ((uint32_t)call_tag & 0xffu)) + 1u;
  u64_t* g_attn = ws + (int64_t)n_heads * n_splits * PG;  // attention output: pair u = elements 2u, 2u+1

  if (bid < n_att) {
    // ------------------------------------------------ attention role ------------------------------------------------
    float* sm_m = reinterpret_cast<float*>(smem);
    float* sm_l = sm_m + NG;
    float* sm_o = sm_l + NG;
    float* comb = sm_o + NG * D;  // [n_splits][D + kAttnPartPad]
    const int split = bid % n_splits, h = bid / n_splits;
    const int kvh = h / n_rep;
    const S* row = reinterpret_cast<const S*>(qkv_);
    const int T_old = kv_len[0];
    St st;
    attn_split_issue<T, D, NW, true, U>(st, tid, k_slab_, v_slab_, 0, stride_h, T_old, 1, 0, kvh, split, n_splits, T_cap, 0);
    float M, L, O;
    attn_split_finish<T, D, NW, true, U>(st, tid, row + (int64_t)h * D, row + (int64_t)(n_heads + kvh) * D,
                                         row + (int64_t)(n_heads + n_kv_heads + kvh) * D, cos_, sin_, n_pos, pos_base[0], scale,
                                         h % n_rep == 0, T_cap, sm_m, sm_l, sm_o, M, L, O);
    float val = 0.f;
    bool have = false;
    if (n_splits == 1) {
      val = L > 0.f ? O / L : 0.f;
      have = true;
    } else if (split != 0) {
      if (tid < D) {
        u64_t* pr = ws + ((int64_t)h * n_splits + split) * PG;
        gr_store(pr + 2 + tid, tag, __float_as_uint(O));
        if (tid == 0) {
          gr_store(pr, tag, __float_as_uint(M));
          gr_store(pr + 1, tag, __float_as_uint(L));
        }
      }
    } else {
      // split 0 merges: own partial straight into the staging area, the others as they arrive (split order, as the combine kernel)
      if (tid < D) {
        comb[kAttnPartPad + tid] = O;
        if (tid == 0) {
          comb[0] = M;
          comb[1] = L;
        }
      }
      bool bad = false;
      const u64_t* gws = ws + (int64_t)h * n_splits * PG;
      for (int i = PG + tid; i < n_splits * PG; i += kAoThreads) {
        u64_t v = 0;
        for (int spins = 0;; ++spins) {
          v = gr_load(gws + i);
          if ((uint32_t)(v >> 32) == tag) break;
          if (spins > (1 << 22)) {
            bad = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        const int s_ = i / PG, e = i % PG;
        comb[s_ * (D + kAttnPartPad) + (e < 2 ? e : e + 2)] = __uint_as_float((uint32_t)v);
      }
      const int any_bad = __syncthreads_or(bad ? 1 : 0);
      if (tid < D) {
        float o1[1];
        attn_split_merge<1>(comb, n_splits, D, tid, o1);
        val = any_bad ? __uint_as_float(0x7fc00000u) : o1[0];
      }
      have = true;
    }
    if (have && tid < D) {
      const uint32_t mine = Elem<T>::from_f(val);
      const uint32_t up = __shfl_down(mine, 1, 64);
      if ((tid & 1) == 0) gr_store(g_attn + (int64_t)h * (D / 2) + tid / 2, tag, mine | (up << 16));
      reinterpret_cast<S*>(attn_out_)[(int64_t)h * D + tid] = (S)mine;  // also kept in memory (debug records, chunked callers)
    }
    return;
  }

  // ---------------------------------------------------- o_proj role ----------------------------------------------------
  // one weight row per wave and trip, every chunk of the first two rows requested before the attention output is waited for
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);  // [K / 2] words: x as the attention role publishes it
  const int g = bid - n_att, G2 = (int)gridDim.x - n_att;
  const int nvec = K / V;
  const int groups = (N + 3) / 4;
  const S* W = reinterpret_cast<const S*>(Wo_);
  uint4 w0[XB * 8], w1[XB * 8];
  const int n0r = g * 4 + wid, n1r = (g + G2) * 4 + wid;
  const bool live0 = g < groups && n0r < N, live1 = g + G2 < groups && n1r < N;
  {
    const S* p0 = W + (int64_t)(live0 ? n0r : N - 1) * K;
    const S* p1 = W + (int64_t)(live1 ? n1r : N - 1) * K;
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) {
      const int v = lane + 64 * c;
      w0[c] = v < nvec ? ldg_nt(p0 + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
      w1[c] = v < nvec ? ldg_nt(p1 + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  bool bad = false;
  if (wid == 0) {
    // readiness sample: the last pair of every head (a head's 64 pairs are published together by one workgroup), then the whole vector
    const int n_gr = K / 2;
    for (int spins = 0;; ++spins) {
      const u64_t v = gr_load(g_attn + (int64_t)(lane % n_heads) * (D / 2) + (D / 2 - 1));
      if (__all((uint32_t)(v >> 32) == tag)) break;
      if (spins > (1 << 22)) {
        bad = true;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    constexpr int GU = XB * 32;  // K / 2 granules over 64 lanes
    u64_t v[GU];
    u64_t got = 0;
    for (int spins = 0; !bad; ++spins) {
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int idx = k * 64 + lane;
        if (idx < n_gr && !((got >> k) & 1)) v[k] = gr_load(g_attn + idx);
      }
      bool ok = true;
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int idx = k * 64 + lane;
        if (idx < n_gr && !((got >> k) & 1)) {
          const bool hh = (uint32_t)(v[k] >> 32) == tag;
          got |= (u64_t)hh << k;
          ok &= hh;
        }
      }
      if (__all(ok)) break;
      if (spins > (1 << 20)) {
        bad = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int k = 0; k < GU; ++k) {
      const int idx = k * 64 + lane;
      if (idx < n_gr) xs[idx] = (uint32_t)v[k];
    }
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  uint4 xr[XB * 8];
#pragma unroll
  for (int c = 0; c < XB * 8; ++c) {
    const int v = lane + 64 * c;
    xr[c] = v < nvec ? *reinterpret_cast<const uint4*>(xs + v * 4) : make_uint4(0u, 0u, 0u, 0u);
  }
  S* y = reinterpret_cast<S*>(y_);
  {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) acc = dot16<T>(w0[c], xr[c], acc);
    acc = wave_sum(acc);
    if (lane == 0 && live0) y[n0r] = any_bad ? (S)0x7fc0u : Elem<T>::from_f(acc);
  }
  {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) acc = dot16<T>(w1[c], xr[c], acc);
    acc = wave_sum(acc);
    if (lane == 0 && live1) y[n1r] = any_bad ? (S)0x7fc0u : Elem<T>::from_f(acc);
  }
  for (int grp = g + 2 * G2; grp < groups; grp += G2) {  // more rows than two trips cover (narrow grids): plain streaming
    int n = grp * 4 + wid;
    const bool live = n < N;
    n = live ? n : N - 1;
    const S* wp = W + (int64_t)n * K;
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) {
      const int v = lane + 64 * c;
      w0[c] = v < nvec ? ldg_nt(wp + (int64_t)v * V) : make_uint4(0u, 0u, 0u, 0u);
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < XB * 8; ++c) acc = dot16<T>(w0[c], xr[c], acc);
    acc = wave_sum(acc);
    if (lane == 0 && live) y[n] = any_bad ? (S)0x7fc0u : Elem<T>::from_f(acc);
  }
}

}  // namespace dl

using namespace dl;

extern "C" int dl_attn_decode_rope_oproj(const void* qkv, const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* pos_base,
                                         const int32_t* kv_len, void* k_slab, void* v_slab, int64_t slab_stride_h, int T_cap, void* attn_out,
                                         void* workspace, int n_splits, int call_tag, int n_heads, int n_kv_heads, int head_dim, const void* w_o,
                                         int N, void* y, int dtype, void* stream) {
  DL_REQUIRE(qkv && cos_tab && sin_tab && pos_base && kv_len && k_slab && v_slab && attn_out && workspace && w_o && y,
             "dl_attn_decode_rope_oproj: NULL pointer");
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_attn_decode_rope_oproj: 16-bit dtypes only");
  DL_REQUIRE(head_dim == 128, "dl_attn_decode_rope_oproj: head_dim must be 128");
  DL_REQUIRE(n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && n_pos > 0 && T_cap > 0 && N > 0 && call_tag >= 0,
             "dl_attn_decode_rope_oproj: bad shape");
  DL_REQUIRE(n_splits >= 1 && n_splits <= 32, "dl_attn_decode_rope_oproj: n_splits=%d must be in [1, 32]", n_splits);
  const int K = n_heads * head_dim;
  DL_REQUIRE(K % 8 == 0 && K / 8 <= 64 * 16, "dl_attn_decode_rope_oproj: K=%d unsupported (<= 8192)", K);
  const int groups = (N + 3) / 4;
  int G2 = groups < 512 ? groups : 512;  // two rows per wave keep W_o's first 2 x 512 x 4 rows in flight from the start
  const int n_att = n_splits * n_heads;
  DL_REQUIRE(n_att + G2 <= 1024, "dl_attn_decode_rope_oproj: %d workgroups cannot all be resident", n_att + G2);
  const float scale = 1.0f / sqrtf((float)head_dim);
  const int xb = (K / 8 + 511) / 512;
  const size_t lds_att = (size_t)(2 * 16 + 16 * 128 + n_splits * (128 + kAttnPartPad)) * sizeof(float);
  const size_t lds_gemv = (size_t)K * 2;
  const size_t smem = lds_att > lds_gemv ? lds_att : lds_gemv;
  hipStream_t st = as_stream(stream);
#define DL_AO_LAUNCH(TT, XBV)                                                                                                           \
  hipLaunchKernelGGL((attn_decode_oproj_kernel<TT, XBV>), dim3((unsigned)(n_att + G2)), dim3(kAoThreads), smem, st, qkv, k_slab, v_slab, \
                     slab_stride_h, kv_len, reinterpret_cast<u64_t*>(workspace), attn_out, n_heads / n_kv_heads, scale, cos_tab, sin_tab, \
                     n_pos, pos_base, T_cap, n_heads, n_kv_heads, n_splits, call_tag, w_o, N, K, y)
  if (dtype == DL_BF16) {
    if (xb <= 1) DL_AO_LAUNCH(bf16_t, 1); else DL_AO_LAUNCH(bf16_t, 2);
  } else {
    if (xb <= 1) DL_AO_LAUNCH(f16_t, 1); else DL_AO_LAUNCH(f16_t, 2);
  }
#undef DL_AO_LAUNCH
  DL_CHECK_LAUNCH("dl_attn_decode_rope_oproj");
  return DL_OK;
}
