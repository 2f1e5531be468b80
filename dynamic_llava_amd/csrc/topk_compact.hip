// F2: order-preserving top-k select (DML:1898-1908) and F3/F4: token compaction + position ids
// (DML:1917-1983).  Integer results -> bit-exact against the oracle.
//
// top-k: one workgroup per row.  Scores become order-preserving uint32 keys in LDS; a radix select finds
// the k-th largest key, ties at the threshold are resolved by index (the pinned total order: score
// descending, then index ascending).  Survivors are emitted in index order with a wavefront ballot +
// cross-wave prefix (no sort => deterministic).
#include "dl_common.h"

namespace dl {

constexpr int kTopkMaxN = 4096;

__device__ __forceinline__ uint32_t order_key(float f) {
  // monotone map float -> uint32 (larger float => larger key); NaN sorts as the largest (torch.sort rule)
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
  if (u == 0x80000000u) u = 0;  // -0.0 == +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Radix select: the k-th largest key is found digit by digit (4 passes of 8 bits: LDS histogram of the still-matching keys, suffix
// scan of the 256 bins by one wave), O(n) work per pass instead of the O(n^2) rank counting that kept one CU busy for ~20 us at n = 576.
// kept = key > threshold, or key == threshold and among the first `need` such keys in index order (the pinned tie rule: score
// descending, then index ascending).  Survivors are emitted in index order with wavefront ballots + cross-wave prefixes: no sort,
// no data-dependent atomics in the result => deterministic and bit-exact against the oracle.
template <typename T>
__global__ void topk_select_kernel(const void* __restrict__ score_, int64_t* __restrict__ keep, int n, int k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);             // [n_pad]
  int* hist = reinterpret_cast<int*>(keys + ((n + 3) & ~3));      // [256]
  int* wave_cnt = hist + 256;                                     // [2][16]
  int* sel = wave_cnt + 32;                                       // [2]: chosen bin, keys above it
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < n; i += blockDim.x) keys[i] = order_key(load1<T>(score_, (int64_t)b * n + i));
  uint32_t prefix = 0, mask = 0;
  int need = k;  // keys still to take among those that match `prefix` on the `mask` bits
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
      const uint32_t key = keys[i];
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (wid == 0) {
      // lane l owns bins 4l .. 4l+3; `above` = number of matching keys in bins higher than this lane's (suffix sum over lanes l+1 .. 63)
      const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const int mine = c0 + c1 + c2 + c3;
      int suf = mine;  // inclusive suffix sum
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
      }
      const int above = suf - mine;
      if (above < need && need <= suf) {  // the k-th key falls into one of this lane's bins (exactly one lane)
        int a = above, bin = 4 * lane + 3;
        if (a + c3 >= need) bin = 4 * lane + 3;
        else if ((a += c3) + c2 >= need) bin = 4 * lane + 2;
        else if ((a += c2) + c1 >= need) bin = 4 * lane + 1;
        else { a += c1; bin = 4 * lane; }
        sel[0] = bin;
        sel[1] = a;
      }
    }
    __syncthreads();
    prefix |= (uint32_t)sel[0] << shift;
    mask |= 255u << shift;
    need -= sel[1];
    __syncthreads();
  }
  const uint32_t thr = prefix;  // the k-th largest key; `need` of the keys equal to it are taken, lowest index first
  int base = 0, eq_base = 0;
  const int passes = (n + blockDim.x - 1) / blockDim.x;
  for (int p = 0; p < passes; ++p) {
    const int i = p * blockDim.x + tid;
    const uint32_t me = i < n ? keys[i] : 0u;
    const bool eq = i < n && me == thr;
    const unsigned long long me_q = __ballot(eq);
    if (lane == 0) wave_cnt[16 + wid] = __popcll(me_q);
    __syncthreads();
    int eq_before = eq_base, eq_tot = 0;
    for (int w = 0; w < nw; ++w) {
      const int c = wave_cnt[16 + w];
      if (w < wid) eq_before += c;
      eq_tot += c;
    }
    eq_before += __popcll(me_q & ((1ull << lane) - 1ull));
    const bool kept = i < n && (me > thr || (eq && eq_before < need));
    const unsigned long long m = __ballot(kept);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int off = base, tot = 0;
    for (int w = 0; w < nw; ++w) {
      const int c = wave_cnt[w];
      if (w < wid) off += c;
      tot += c;
    }
    if (kept) keep[(int64_t)b * k + off + __popcll(m & ((1ull << lane) - 1ull))] = (int64_t)i;
    base += tot;
    eq_base += eq_tot;
    __syncthreads();
  }
}

template <typename T>
__device__ __forceinline__ void unpack_u4(const uint4& r, float (&f)[Elem<T>::kVec]) {
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

// one workgroup per OUTPUT token: binary-search its row, map to the source token, copy H elements.
template <typename T>
__global__ __launch_bounds__(256) void compact_tokens_kernel(const void* __restrict__ in_, void* __restrict__ out_,
                                                              const int64_t* __restrict__ keep, const int32_t* __restrict__ cu_in,
                                                              const int32_t* __restrict__ cu_out, const int32_t* __restrict__ img_start,
                                                              int32_t* __restrict__ pos_out, int B, int n_img, int k, int H,
                                                              const void* __restrict__ norm_w, float eps, void* __restrict__ x_out) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ int64_t sh_src;
  __shared__ float red[4];
  const int t = blockIdx.x;
  if (threadIdx.x == 0) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cu_out[mid] <= t) lo = mid; else hi = mid;
    }
    const int j = t - cu_out[lo];
    const int s = img_start[lo];
    int src;  // in-row index in the un-compacted sequence == the token's original position
    if (j < s) src = j;
    else if (j < s + k) src = s + (int)keep[(int64_t)lo * k + (j - s)];
    else src = j + (n_img - k);
    pos_out[t] = src;
    sh_src = (int64_t)cu_in[lo] + src;
  }
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const S*>(in_) + sh_src * H);
  uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<S*>(out_) + (int64_t)t * H);
  if (norm_w == nullptr) {
    for (int v = threadIdx.x; v < H / V; v += 256) dst[v] = src[v];
    return;
  }
  // fused RMSNorm of the compacted row (the input_layernorm of layer `sparse_layer`, DML:134-139): the row is in registers anyway.
  // Same thread -> chunk map, same reduction and rounding order as dl_rmsnorm (elementwise.hip), so the bits are the same.
  constexpr int kMaxVec = 8;  // H <= 256 * 8 * kVec
  const int nvec = H / V;
  uint4 raw[kMaxVec];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      raw[i] = src[v];
      dst[v] = raw[i];
      float x[V];
      unpack_u4<T>(raw[i], x);
#pragma unroll
      for (int j = 0; j < V; ++j) ss += x[j] * x[j];
    }
  }
  const float tot = block_sum<4>(ss, red);
  const float rstd = rsqrtf(tot / (float)H + eps);
  const S* w = reinterpret_cast<const S*>(norm_w);
  S* xo = reinterpret_cast<S*>(x_out) + (int64_t)t * H;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      float x[V], wv[V], o[V];
      unpack_u4<T>(raw[i], x);
      load16<T>(w + v * V, wv);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = wv[j] * Elem<T>::round(x[j] * rstd);  // cast, THEN weight
      store16<T>(xo + v * V, o);
    }
  }
}

}  // namespace dl

using namespace dl;

extern "C" int dl_topk_select(const void* score, int64_t* keep_idx, int B, int n, int k, int dtype, void* stream) {
  DL_REQUIRE(score && keep_idx, "dl_topk_select: NULL pointer");
  DL_REQUIRE(B > 0 && n > 0 && n <= kTopkMaxN && k >= 0 && k <= n, "dl_topk_select: bad shape B=%d n=%d k=%d", B, n, k);
  if (k == 0) return DL_OK;
  int threads = n <= 64 ? 64 : (n <= 256 ? 256 : 1024);
  const size_t smem = (size_t)((n + 3) & ~3) * 4 + (256 + 32 + 2) * 4;
  DL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((topk_select_kernel<T>), dim3((unsigned)B), dim3(threads), smem, as_stream(stream), score, keep_idx, n, k);
  });
  DL_CHECK_LAUNCH("dl_topk_select");
  return DL_OK;
}

extern "C" int dl_compact_tokens(const void* h_in, void* h_out, const int64_t* keep_idx, const int32_t* cu_in, const int32_t* cu_out,
                                 const int32_t* img_start, int32_t* pos_out, int B, int n_img, int k, int total_out, int H, const void* norm_w,
                                 float eps, void* x_out, int dtype, void* stream) {
  DL_REQUIRE(h_in && h_out && cu_in && cu_out && img_start && pos_out, "dl_compact_tokens: NULL pointer");
  DL_REQUIRE((norm_w == nullptr) == (x_out == nullptr), "dl_compact_tokens: norm_w and x_out go together");
  DL_REQUIRE(keep_idx || k == 0, "dl_compact_tokens: keep_idx is NULL");
  DL_REQUIRE(B > 0 && n_img >= 0 && k >= 0 && k <= n_img && total_out >= 0 && H > 0, "dl_compact_tokens: bad shape");
  if (total_out == 0) return DL_OK;
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0, "dl_compact_tokens: H=%d must be a multiple of %d", H, Elem<T>::kVec);
    DL_REQUIRE(norm_w == nullptr || H <= 256 * 8 * Elem<T>::kVec, "dl_compact_tokens: H=%d too wide for the fused norm", H);
    hipLaunchKernelGGL((compact_tokens_kernel<T>), dim3((unsigned)total_out), dim3(256), 0, as_stream(stream), h_in, h_out, keep_idx,
                       cu_in, cu_out, img_start, pos_out, B, n_img, k, H, norm_w, eps, x_out);
  });
  DL_CHECK_LAUNCH("dl_compact_tokens");
  return DL_OK;
}

// ---- rows of ONE packed sequence kept or dropped by a device-side decision over a span (the instruct predictor's prefill compaction,
// DML:2261-2375: tokens [span0, span0 + n_span) of the last instruct turn survive where decision != 0, everything else always).  One workgroup
// per INPUT row: its destination = its index minus the drops in front of it (a block sum over the span's decisions: n_span is a few dozen),
// 16-byte row copy + the original position id.  The kept count never goes to the host: it is written to `counts` = {kept rows, kept rows - 1}
// and `cu_out` = {0, kept rows}, which the following launches (grids sized for the UPPER bound `total`) read from device memory. ----
namespace dl {
template <typename T>
__global__ __launch_bounds__(256) void compact_rows_by_mask_kernel(const void* __restrict__ h_in, const int32_t* __restrict__ pos_in, const int32_t* __restrict__ dec,
                                                                   int span0, int n_span, int total, int H, void* __restrict__ h_out,
                                                                   int32_t* __restrict__ pos_out, int32_t* __restrict__ cu_out, int64_t* __restrict__ counts) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ int red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  // drops in front of row i, and (workgroup 0) in total
  const int upto = i < span0 ? 0 : (i - span0 < n_span ? i - span0 : n_span);
  int before = 0, all = 0;
  for (int j = tid; j < n_span; j += 256) {
    const int drop = dec[j] != 0 ? 0 : 1;
    all += drop;
    before += j < upto ? drop : 0;
  }
  before = (int)wave_sum((float)before);  // exact: counts are far below 2^24
  all = (int)wave_sum((float)all);
  if ((tid & 63) == 0) red[tid >> 6] = before;
  __syncthreads();
  before = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = all;
  __syncthreads();
  all = red[0] + red[1] + red[2] + red[3];
  if (i == 0 && tid == 0) {
    cu_out[0] = 0;
    cu_out[1] = total - all;
    counts[0] = total - all;
    counts[1] = total - all - 1;
  }
  const bool keep = i < span0 || i >= span0 + n_span || dec[i - span0] != 0;
  if (!keep) return;
  const int dst = i - before;
  const S* src = reinterpret_cast<const S*>(h_in) + (int64_t)i * H;
  S* out = reinterpret_cast<S*>(h_out) + (int64_t)dst * H;
  for (int v = tid; v < H / V; v += 256) *reinterpret_cast<uint4*>(out + v * V) = *reinterpret_cast<const uint4*>(src + v * V);
  if (tid == 0) pos_out[dst] = pos_in ? pos_in[i] : i;
}
}  // namespace dl

extern "C" int dl_compact_rows_by_mask(const void* h_in, const int32_t* pos_in, const int32_t* decision, int span0, int n_span, int total, int H, void* h_out,
                                       int32_t* pos_out, int32_t* cu_out, int64_t* counts, int dtype, void* stream) {
  DL_REQUIRE(h_in && h_out && pos_out && cu_out && counts && (decision || n_span == 0), "dl_compact_rows_by_mask: NULL pointer");
  DL_REQUIRE(total > 0 && H > 0 && span0 >= 0 && n_span >= 0 && span0 + n_span <= total, "dl_compact_rows_by_mask: bad span [%d, %d) of %d rows", span0, span0 + n_span, total);
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(H % Elem<T>::kVec == 0, "dl_compact_rows_by_mask: H=%d must be a multiple of %d", H, Elem<T>::kVec);
    hipLaunchKernelGGL((compact_rows_by_mask_kernel<T>), dim3((unsigned)total), dim3(256), 0, as_stream(stream), h_in, pos_in, decision, span0, n_span, total, H, h_out,
                       pos_out, cu_out, counts);
  });
  DL_CHECK_LAUNCH("dl_compact_rows_by_mask");
  return DL_OK;
}

// ---- device-side prompt layout: one workgroup per row ----
namespace dl {
__global__ __launch_bounds__(256) void prompt_layout_kernel(const int64_t* __restrict__ ids, int W, int n_feat, int image_token, int u0, int u1,
                                                             int32_t* __restrict__ seg, int64_t* __restrict__ text_src,
                                                             int64_t* __restrict__ text_dst, int64_t* __restrict__ img_dst,
                                                             int32_t* __restrict__ img_start, int32_t* __restrict__ err,
                                                             const int32_t* __restrict__ w_true, int n_drop, int32_t* __restrict__ cu,
                                                             int32_t* __restrict__ cu2, int32_t* __restrict__ lens, int64_t* __restrict__ last_rows) {
  __shared__ int s_pos, s_cnt, s_user;
  const int b = blockIdx.x, tid = threadIdx.x, B = gridDim.x;
  const int64_t* row = ids + (int64_t)b * W;
  // `w_true` (device scalar): only the first Wt of the W columns are the prompt (the buffer is a width BUCKET shared by prompts of several
  // widths: one captured prefill graph serves them all).  The sequences are packed at their TRUE lengths; the (W - Wt) columns of padding are
  // sent to the unused rows at the end of the packed matrix.
  int Wt = w_true ? w_true[0] : W;
  Wt = Wt < 1 ? 1 : (Wt > W ? W : Wt);
  if (tid == 0) {
    s_pos = Wt;
    s_cnt = 0;
    s_user = -1;
  }
  __syncthreads();
  for (int c = tid; c < Wt; c += 256)
    if (row[c] == (int64_t)image_token) {
      atomicMin(&s_pos, c);
      atomicAdd(&s_cnt, 1);
    }
  __syncthreads();
  const int p = s_pos < Wt ? s_pos : 0;
  // last "USER:" pair inside the instruct span (columns p+1 .. Wt-1), as an offset from its start (ARCH:418-454 with no labels: the
  // span runs to the end of the row)
  for (int c = p + 1 + tid; c + 1 < Wt; c += 256)
    if (row[c] == (int64_t)u0 && row[c + 1] == (int64_t)u1) atomicMax(&s_user, c - (p + 1));
  __syncthreads();
  const int n_row = Wt - 1 + n_feat;  // packed rows of this sequence
  const int64_t base = (int64_t)b * n_row;
  const int64_t trash = (int64_t)B * n_row + (int64_t)b * (W - Wt);  // first unused packed row given to this sequence's padding columns
  for (int c = tid; c < W; c += 256) {
    if (c == p) continue;
    const int j = c < p ? c : c - 1;  // index among the W - 1 non-image columns
    const bool pad = c >= Wt;
    text_src[(int64_t)b * (W - 1) + j] = (int64_t)b * W + (pad ? 0 : c);  // padding gathers column 0's (valid) token
    text_dst[(int64_t)b * (W - 1) + j] = pad ? trash + (c - Wt) : base + (c < p ? c : c - 1 + n_feat);
  }
  for (int i = tid; i < n_feat; i += 256) img_dst[(int64_t)b * n_feat + i] = base + p + i;
  if (tid == 0) {
    seg[b * 8 + 0] = p;
    seg[b * 8 + 1] = s_user < 0 ? 0 : s_user;
    seg[b * 8 + 2] = s_cnt;
    seg[b * 8 + 3] = Wt;
    img_start[b] = p;
    if (s_cnt != 1) atomicMax(err, 1 + b);
    // the prefill plan's device-side metadata at the TRUE lengths (the captured launches are sized for the bucket and read these)
    const int n2 = n_row - n_drop;  // rows per sequence in layers >= sparse_layer
    if (cu) {
      cu[b] = b * n_row;
      if (b == B - 1) cu[B] = B * n_row;
    }
    if (cu2) {
      cu2[b] = b * n2;
      if (b == B - 1) cu2[B] = B * n2;
    }
    if (lens) {
      lens[b] = n_row;
      lens[B + b] = n2;
    }
    if (last_rows) last_rows[b] = (int64_t)(b + 1) * n2 - 1;
  }
}
}  // namespace dl

extern "C" int dl_prompt_layout(const int64_t* input_ids, int B, int W, int n_feat, int image_token, int user_id0, int user_id1, int32_t* seg,
                                int64_t* text_src, int64_t* text_dst, int64_t* img_dst, int32_t* img_start, int32_t* err, const int32_t* w_true,
                                int n_drop, int32_t* cu_seqlens, int32_t* cu_seqlens_sparse, int32_t* lens, int64_t* last_rows, void* stream) {
  DL_REQUIRE(input_ids && seg && text_src && text_dst && img_dst && img_start && err, "dl_prompt_layout: NULL pointer");
  DL_REQUIRE(B > 0 && W > 0 && n_feat >= 0 && n_drop >= 0 && n_drop <= n_feat, "dl_prompt_layout: bad shape");
  hipLaunchKernelGGL(dl::prompt_layout_kernel, dim3((unsigned)B), dim3(256), 0, dl::as_stream(stream), input_ids, W, n_feat, image_token, user_id0,
                     user_id1, seg, text_src, text_dst, img_dst, img_start, err, w_true, n_drop, cu_seqlens, cu_seqlens_sparse, lens, last_rows);
  DL_CHECK_LAUNCH("dl_prompt_layout");
  return DL_OK;
}
