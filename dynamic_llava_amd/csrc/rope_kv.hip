// RoPE (DML:253-285) fused with the KV-slab append (CU:109-268): one pass over the freshly projected
// q|k|v rows -- q,k rotated in place, rotated k and v stored into the per-layer slab at
// slot kv_base[b] + j.  HBM-bound: every element is read once and written once (k, v twice).
#include "dl_common.h"

namespace dl {

constexpr int kRopeThreads = 256;

// PARTS: the projection arrives as `n_parts` fp32 partial sums [part][total][row_w] (dl_linear_packed's k ranges, DL_LP_PARTS): a chunk is their sum in
// part order, rounded once to the model dtype -- exactly what the GEMM would have stored -- and q / k (rotated) and v are WRITTEN to qkv instead of updated.
template <typename T, bool PARTS>
__global__ __launch_bounds__(kRopeThreads) void rope_kv_write_kernel(
    void* __restrict__ qkv_, const void* __restrict__ cos_, const void* __restrict__ sin_, int n_pos,
    const int32_t* __restrict__ cu, const int32_t* __restrict__ pos, const int32_t* __restrict__ pos_base,
    const int32_t* __restrict__ kv_base, void* __restrict__ k_slab_, void* __restrict__ v_slab_, int64_t stride_b,
    int64_t stride_h, int T_cap, int B, int nH, int nKV, int d, const float* __restrict__ parts = nullptr, int n_parts = 0, int64_t part_stride = 0) {
  constexpr int V = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ int sh[4];
  const int t = blockIdx.x;
  if (threadIdx.x == 0) {
    sh[3] = t < cu[B];  // a launch sized for a width bucket: rows past the last sequence are padding (nothing to rotate or append)
    int lo = 0, hi = B;  // find b with cu[b] <= t < cu[b+1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cu[mid] <= t) lo = mid; else hi = mid;
    }
    const int j = t - cu[lo];
    int p = pos ? pos[t] : pos_base[lo] + j;
    p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
    sh[0] = lo;
    sh[1] = p;
    sh[2] = kv_base[lo] + j;
  }
  __syncthreads();
  if (!sh[3]) return;
  const int b = sh[0], p = sh[1], slot = sh[2];
  const int row_w = (nH + 2 * nKV) * d;
  S* row = reinterpret_cast<S*>(qkv_) + (int64_t)t * row_w;
  const S* cosr = reinterpret_cast<const S*>(cos_) + (int64_t)p * d;
  const S* sinr = reinterpret_cast<const S*>(sin_) + (int64_t)p * d;
  S* k_slab = reinterpret_cast<S*>(k_slab_);
  S* v_slab = reinterpret_cast<S*>(v_slab_);
  const bool store_ok = slot >= 0 && slot < T_cap;
  const int half = d / 2;
  const int cph = half / V;                 // rotation chunks per head
  const int n_rot = (nH + nKV) * cph;       // q heads then k heads
  const int cpv = d / V;
  const int n_cpy = nKV * cpv;
  const int items = n_rot + n_cpy;
  const float* prow = PARTS ? parts + (int64_t)t * row_w : nullptr;
  auto load_sum = [&](int col, float (&o)[V]) {  // V model-dtype values at column `col` of this row: sum of the partial sums, part 0 first, rounded once
    static_assert(V % 4 == 0, "16-byte chunks of a 2-byte type");
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      float4 a = *reinterpret_cast<const float4*>(prow + col + q * 4);
      for (int s_ = 1; s_ < n_parts; ++s_) {
        const float4 b_ = *reinterpret_cast<const float4*>(prow + (int64_t)s_ * part_stride + col + q * 4);
        a.x += b_.x; a.y += b_.y; a.z += b_.z; a.w += b_.w;
      }
      o[q * 4] = Elem<T>::round(a.x); o[q * 4 + 1] = Elem<T>::round(a.y); o[q * 4 + 2] = Elem<T>::round(a.z); o[q * 4 + 3] = Elem<T>::round(a.w);
    }
  };
  for (int it = blockIdx.y * kRopeThreads + threadIdx.x; it < items; it += gridDim.y * kRopeThreads) {
    if (it < n_rot) {
      const int h = it / cph, c = (it - h * cph) * V;
      S* x = row + h * d;
      float x1[V], x2[V], cs[V], sn[V], o1[V], o2[V];
      if constexpr (PARTS) {
        load_sum(h * d + c, x1);
        load_sum(h * d + half + c, x2);
      } else {
        load16<T>(x + c, x1);
        load16<T>(x + half + c, x2);
      }
      load16<T>(cosr + c, cs);  // table = cat(freqs, freqs): second half equals the first
      load16<T>(sinr + c, sn);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        // q_embed = (q * cos) + (rotate_half(q) * sin), each op rounded to the model dtype (DML:283-284)
        o1[i] = Elem<T>::round(Elem<T>::round(x1[i] * cs[i]) + Elem<T>::round(-x2[i] * sn[i]));
        o2[i] = Elem<T>::round(Elem<T>::round(x2[i] * cs[i]) + Elem<T>::round(x1[i] * sn[i]));
      }
      const uint4 p1 = pack16<T>(o1), p2 = pack16<T>(o2);
      *reinterpret_cast<uint4*>(x + c) = p1;
      *reinterpret_cast<uint4*>(x + half + c) = p2;
      if (h >= nH && store_ok) {
        S* dst = k_slab + (int64_t)b * stride_b + (int64_t)(h - nH) * stride_h + (int64_t)slot * d;
        *reinterpret_cast<uint4*>(dst + c) = p1;
        *reinterpret_cast<uint4*>(dst + half + c) = p2;
      }
    } else if (PARTS || store_ok) {
      const int iv = it - n_rot;
      const int h = iv / cpv, c = (iv - h * cpv) * V;
      uint4 val;
      if constexpr (PARTS) {
        float vv[V];
        load_sum((nH + nKV + h) * d + c, vv);
        val = pack16<T>(vv);
        *reinterpret_cast<uint4*>(row + (nH + nKV + h) * d + c) = val;  // the prefill attention reads v from the packed projection
      } else {
        val = *reinterpret_cast<const uint4*>(row + (nH + nKV + h) * d + c);
      }
      if (store_ok) {
        S* dst = v_slab + (int64_t)b * stride_b + (int64_t)h * stride_h + (int64_t)slot * d;
        *reinterpret_cast<uint4*>(dst + c) = val;
      }
    }
  }
}

}  // namespace dl

using namespace dl;

static int rope_kv_write_impl(const char* who, void* qkv, const float* parts, int n_parts, const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* cu_seqlens,
                              const int32_t* pos, const int32_t* pos_base, const int32_t* kv_base, void* k_slab, void* v_slab, int64_t slab_stride_b, int64_t slab_stride_h,
                              int T_cap, int B, int total, int n_heads, int n_kv_heads, int head_dim, int dtype, void* stream) {
  DL_REQUIRE(B > 0 && total >= 0 && n_heads > 0 && n_kv_heads > 0 && n_pos > 0, "%s: bad shape", who);
  if (total == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(qkv && cos_tab && sin_tab && cu_seqlens && kv_base && k_slab && v_slab, "%s: NULL pointer", who);
  DL_REQUIRE(pos || pos_base, "%s: one of pos / pos_base is required", who);
  DL_REQUIRE(!parts || (n_parts >= 1 && n_parts <= 8 && ((uintptr_t)parts & 15) == 0 && (dtype == DL_BF16 || dtype == DL_F16) && head_dim % 8 == 0),
             "%s: 1..8 partial sums, 16-byte aligned, 2-byte model dtypes", who);
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(head_dim > 0 && (head_dim / 2) % Elem<T>::kVec == 0, "%s: head_dim=%d unsupported", who, head_dim);
    const int items = (n_heads + n_kv_heads) * (head_dim / 2 / Elem<T>::kVec) + n_kv_heads * (head_dim / Elem<T>::kVec);
    int gy = 1;
    if (total < 256) {  // decode: spread one token's heads over several workgroups
      gy = (items + kRopeThreads - 1) / kRopeThreads;
      if (gy > 8) gy = 8;
    }
    const int64_t part_stride = (int64_t)total * (n_heads + 2 * n_kv_heads) * head_dim;
    if (parts)
      hipLaunchKernelGGL((rope_kv_write_kernel<T, true>), dim3((unsigned)total, (unsigned)gy), dim3(kRopeThreads), 0, as_stream(stream), qkv, cos_tab, sin_tab, n_pos, cu_seqlens, pos,
                         pos_base, kv_base, k_slab, v_slab, slab_stride_b, slab_stride_h, T_cap, B, n_heads, n_kv_heads, head_dim, parts, n_parts, part_stride);
    else
      hipLaunchKernelGGL((rope_kv_write_kernel<T, false>), dim3((unsigned)total, (unsigned)gy), dim3(kRopeThreads), 0, as_stream(stream), qkv,
                       cos_tab, sin_tab, n_pos, cu_seqlens, pos, pos_base, kv_base, k_slab, v_slab, slab_stride_b, slab_stride_h,
                       T_cap, B, n_heads, n_kv_heads, head_dim);
  });
  DL_CHECK_LAUNCH(who);
  return DL_OK;
}

extern "C" int dl_rope_kv_write(void* qkv, const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* cu_seqlens,
                                const int32_t* pos, const int32_t* pos_base, const int32_t* kv_base, void* k_slab, void* v_slab,
                                int64_t slab_stride_b, int64_t slab_stride_h, int T_cap, int B, int total, int n_heads,
                                int n_kv_heads, int head_dim, int dtype, void* stream) {
  return rope_kv_write_impl("dl_rope_kv_write", qkv, nullptr, 0, cos_tab, sin_tab, n_pos, cu_seqlens, pos, pos_base, kv_base, k_slab, v_slab, slab_stride_b, slab_stride_h, T_cap, B, total,
                            n_heads, n_kv_heads, head_dim, dtype, stream);
}

extern "C" int dl_rope_kv_write_parts(void* qkv_out, const float* parts, int n_parts, const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* cu_seqlens,
                                      const int32_t* pos, const int32_t* pos_base, const int32_t* kv_base, void* k_slab, void* v_slab,
                                      int64_t slab_stride_b, int64_t slab_stride_h, int T_cap, int B, int total, int n_heads,
                                      int n_kv_heads, int head_dim, int dtype, void* stream) {
  DL_REQUIRE(parts || total == 0, "dl_rope_kv_write_parts: NULL partial sums");
  return rope_kv_write_impl("dl_rope_kv_write_parts", qkv_out, parts, n_parts, cos_tab, sin_tab, n_pos, cu_seqlens, pos, pos_base, kv_base, k_slab, v_slab, slab_stride_b, slab_stride_h, T_cap, B,
                            total, n_heads, n_kv_heads, head_dim, dtype, stream);
}

// ---- in-place packing of the kept rows of a just-appended chunk (multi-round "new instruct" call: DML:2506-2521 decides per chunk
// token whether its K/V stays in layers >= sparse_layer; the reference then slices / concatenates / zero-pads row by row on the host,
// CU:165-241).  Every (K|V tensor, layer, row, kv head) is one workgroup: the chunk sits at slots [kv_len[b], kv_len[b] + T); the rows
// with keep[b,t] != 0 move, in order, to [kv_len[b], kv_len[b] + n_keep).  dst <= src always, rows are moved in ascending rounds of
// `kPackRows` (all sources of a round are in registers before any destination is written), so the move is safe in place.
namespace dl {
constexpr int kPackRows = 16;  // rows per round: 16 lanes x 16 bytes per row (head_dim 128 x 2 bytes), 256 threads

__global__ __launch_bounds__(256) void kv_pack_rows_kernel(unsigned char* __restrict__ k0, unsigned char* __restrict__ v0, int64_t layer_stride_b,
                                                            int64_t stride_b_b, int64_t stride_h_b, int row_bytes, int T_cap,
                                                            const int32_t* __restrict__ keep, const int32_t* __restrict__ kv_len, int T) {
  extern __shared__ int dst_of[];  // [T]: destination slot offset of chunk token t, or -1
  const int h = blockIdx.x, b = blockIdx.y, lz = blockIdx.z;
  const int layer = lz >> 1;
  unsigned char* base = ((lz & 1) ? v0 : k0) + (int64_t)layer * layer_stride_b + (int64_t)b * stride_b_b + (int64_t)h * stride_h_b;
  const int len = kv_len[b];
  if (threadIdx.x == 0) {
    int n = 0;
    for (int t = 0; t < T; ++t) dst_of[t] = keep[(int64_t)b * T + t] != 0 ? n++ : -1;
  }
  __syncthreads();
  const int cpr = row_bytes / 16;            // 16-byte chunks per row
  const int rows_par = 256 / cpr;            // rows moved per round
  const int r = threadIdx.x / cpr, c = threadIdx.x % cpr;
  for (int t0 = 0; t0 < T; t0 += rows_par) {
    const int t = t0 + r;
    uint4 val = make_uint4(0, 0, 0, 0);
    int d = -1;
    if (r < rows_par && t < T) {
      d = dst_of[t];
      if (d >= 0 && d != t && len + t < T_cap) val = *reinterpret_cast<const uint4*>(base + (int64_t)(len + t) * row_bytes + c * 16);
    }
    __syncthreads();
    if (d >= 0 && d != t && len + t < T_cap) *reinterpret_cast<uint4*>(base + (int64_t)(len + d) * row_bytes + c * 16) = val;
    __syncthreads();
  }
}
}  // namespace dl

extern "C" int dl_kv_pack_rows(void* k_slab0, void* v_slab0, int64_t layer_stride, int n_layers, int64_t slab_stride_b, int64_t slab_stride_h,
                               int T_cap, const int32_t* keep, const int32_t* kv_len, int B, int n_kv_heads, int T, int head_dim, int dtype,
                               void* stream) {
  DL_REQUIRE(k_slab0 && v_slab0 && keep && kv_len, "dl_kv_pack_rows: NULL pointer");
  DL_REQUIRE(B > 0 && n_kv_heads > 0 && T >= 0 && n_layers >= 0 && T_cap > 0, "dl_kv_pack_rows: bad shape");
  if (T == 0 || n_layers == 0) return DL_OK;
  const int es = dtype == DL_F32 ? 4 : 2;
  DL_REQUIRE(dtype == DL_F32 || dtype == DL_F16 || dtype == DL_BF16, "dl_kv_pack_rows: unsupported dtype %d", dtype);
  const int row_bytes = head_dim * es;
  DL_REQUIRE(row_bytes % 16 == 0 && row_bytes / 16 <= 256 && 256 % (row_bytes / 16) == 0, "dl_kv_pack_rows: head_dim=%d unsupported", head_dim);
  DL_REQUIRE((size_t)T * sizeof(int) <= 48 * 1024, "dl_kv_pack_rows: chunk of %d tokens too long", T);
  hipLaunchKernelGGL(dl::kv_pack_rows_kernel, dim3((unsigned)n_kv_heads, (unsigned)B, (unsigned)(2 * n_layers)), dim3(256), (size_t)T * sizeof(int),
                     dl::as_stream(stream), reinterpret_cast<unsigned char*>(k_slab0), reinterpret_cast<unsigned char*>(v_slab0), layer_stride * es,
                     slab_stride_b * es, slab_stride_h * es, row_bytes, T_cap, keep, kv_len, T);
  DL_CHECK_LAUNCH("dl_kv_pack_rows");
  return DL_OK;
}
