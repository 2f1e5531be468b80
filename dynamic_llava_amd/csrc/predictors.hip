// F1: VisionPredictor (DML:1308-1359 + CTL:126-180,276-323) -> keep/drop logits + log-softmax score.
// F6: TextPredictor (DML:1362-1387) + decision (DML:2388-2391).
// Decode-step bookkeeping: greedy argmax + device-side KV length advance (replaces CU:153-164's host sync).
//
// The vision predictor is a fixed pipeline of launches on the caller's stream, built from dl_layernorm,
// dl_linear (MFMA, fused bias/GELU/residual epilogues) and dl_attn_prefill (non-causal) plus three small
// kernels defined here.  Every intermediate is rounded to the model dtype exactly where the eager reference
// materialises a tensor, so scores tie (and break ties) the same way.
#include "dl_common.h"
#include "tp_body.h"

namespace dl {
int linear_launch(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, const void* R, int64_t ldr, int M,
                  int N, int K, int flags, int dtype, hipStream_t st);

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

__global__ void vp_build_index_kernel(const int32_t* __restrict__ cu, const int32_t* __restrict__ img_start, int B, int n,
                                      int32_t* __restrict__ row_index, int32_t* __restrict__ cu_img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * n) {
    const int b = i / n;
    row_index[i] = cu[b] + img_start[b] + (i - b * n);
  }
  if (i <= B) cu_img[i] = i * n;
}

// z[m, :C/2] = x[m, :C/2];  z[m, C/2:] = mean over the image's n tokens of x[., C/2:]   (DML:1353-1357, policy == 1)
// grid (B, ceil(C/2 / 64)); block 1024 = 64 channels x 16 token groups (coalesced 128-byte channel runs).
template <typename T>
__global__ __launch_bounds__(1024) void vp_pool_concat_kernel(const void* __restrict__ x_, void* __restrict__ z_, int n, int C) {
  __shared__ float part[16][65];
  const int b = blockIdx.x;
  const int half = C / 2;
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const bool ok = c < half;
  float s = 0.f;
  if (ok) {
#pragma unroll 4
    for (int i = g; i < n; i += 16) s += load1<T>(x_, ((int64_t)b * n + i) * C + half + c);
  }
  part[g][cl] = s;
  __syncthreads();
  if (!ok) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += part[k][cl];
  const float denom = Elem<T>::round((float)n);  // torch.sum(image_policy) is itself a model-dtype tensor
  const float gm = Elem<T>::round(Elem<T>::round(tot) / denom);
  for (int i = g; i < n; i += 16) {
    const int64_t r = ((int64_t)b * n + i) * C;
    store1<T>(z_, r + half + c, gm);
    store1<T>(z_, r + c, load1<T>(x_, r + c));
  }
}

// logits[m,:] = z[m,:] @ W^T + b (2 outputs); score[m] = log_softmax(logits[m,:])[0]   (DML:1344,1867,1898)
template <typename T>
__global__ __launch_bounds__(256) void vp_head_kernel(const void* __restrict__ z_, const void* __restrict__ w_, const void* __restrict__ b_,
                                                       void* __restrict__ logits_, void* __restrict__ score_, int M, int K) {
  extern __shared__ float wsm[];  // [2][K]
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) wsm[i] = load1<T>(w_, i);
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float a0 = 0.f, a1 = 0.f;
  for (int k = 0; k < K; ++k) {
    const float z = load1<T>(z_, (int64_t)m * K + k);
    a0 = fmaf(z, wsm[k], a0);
    a1 = fmaf(z, wsm[K + k], a1);
  }
  const float l0 = Elem<T>::round(a0 + load1<T>(b_, 0));
  const float l1 = Elem<T>::round(a1 + load1<T>(b_, 1));
  store1<T>(logits_, (int64_t)m * 2, l0);
  store1<T>(logits_, (int64_t)m * 2 + 1, l1);
  const float mx = fmaxf(l0, l1);
  const float lse = logf(expf(l0 - mx) + expf(l1 - mx));
  store1<T>(score_, m, (l0 - mx) - lse);
}

template <typename T>
__global__ __launch_bounds__(256) void tp_stage1_kernel(const void* __restrict__ x_, int64_t x_rs, const void* __restrict__ ln_w,
                                                         const void* __restrict__ ln_b, const void* __restrict__ w1,
                                                         const void* __restrict__ b1, float* __restrict__ h1, int H, int D) {
  tp_stage1_body<T>(x_, x_rs, ln_w, ln_b, w1, b1, h1, H, D, (int)blockIdx.x, (int)blockIdx.y, nullptr, 0u);
}

template <typename T>
__global__ __launch_bounds__(256) void tp_stage2a_kernel(const float* __restrict__ h1, const void* __restrict__ w3, const void* __restrict__ b3,
                                                          float* __restrict__ a1, int D) {
  tp_stage2a_body<T>(h1, w3, b3, a1, D, (int)blockIdx.x, (int)blockIdx.y, nullptr, 0u);
}

template <typename T>
__global__ __launch_bounds__(512) void tp_stage2b_kernel(const float* __restrict__ a1g, const void* w5, const void* b5, const void* w7,
                                                          const void* b7, float* __restrict__ logits, int32_t* __restrict__ decision, int D) {
  extern __shared__ float tp_sm[];  // [D/2] + [D/4] + [2]
  tp_stage2b_body<T>(a1g, w5, b5, w7, b7, logits, decision, D, (int)blockIdx.x, tp_sm);
}

// ---- greedy argmax + device-side bookkeeping.  grid (B).  The row (V logits) is read with 16-byte loads, 4 per thread in flight
// (one round trip for V <= 32768), and thread 0 requests the per-row state it will update before the reduction ----
template <typename T>
__global__ __launch_bounds__(1024) void decode_advance_kernel(const void* __restrict__ logits, int64_t row_stride, int V,
                                                               int64_t* __restrict__ next_ids, int64_t* __restrict__ out_ids, int out_cap,
                                                               int32_t* __restrict__ step, int32_t* __restrict__ finished, int eos_id,
                                                               int eos_id2, int eos_id3, int pad_id, int32_t* __restrict__ kv_len_full,
                                                               int32_t* __restrict__ kv_len_sparse, const int32_t* __restrict__ decision,
                                                               int min_new_tokens) {
  constexpr int VE = Elem<T>::kVec;
  using S = typename Elem<T>::storage;
  __shared__ float smax[16];
  __shared__ int sidx[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int st_fin = 0, st_step = 0, st_dec = 1, st_full = 0, st_sparse = 0;
  if (tid == 0) {
    if (finished) st_fin = finished[b];
    if (step) st_step = step[b];
    if (decision) st_dec = decision[b];
    if (kv_len_full) st_full = kv_len_full[b];
    if (kv_len_sparse) st_sparse = kv_len_sparse[b];
  }
  const S* row = reinterpret_cast<const S*>(logits) + (int64_t)b * row_stride;
  const bool vec_ok = (row_stride % VE == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  // HF MinNewTokensLengthLogitsProcessor: the EOS logit is -inf while fewer than min_new_tokens tokens exist (uniform per row)
  // (all EOS ids of the set: eos_id2 / eos_id3 are -1 when unused, and no vocabulary index is negative)
  const bool ban = eos_id >= 0 && min_new_tokens > 0 && (step ? step[b] : 0) < min_new_tokens;
  const int banned = ban ? eos_id : -1, banned2 = ban ? eos_id2 : -1, banned3 = ban ? eos_id3 : -1;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const int n_chunks = V / VE;
  if (vec_ok) {
    for (int c0 = 0; c0 < n_chunks; c0 += 4 * 1024) {
      float x[4][VE];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 1024 + tid;
        if (c < n_chunks) load16<T>(row + (int64_t)c * VE, x[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 1024 + tid;
        if (c < n_chunks)
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const int v = c * VE + e;
            if (v == banned || v == banned2 || v == banned3) continue;
            if (x[u][e] > best || (x[u][e] == best && v < bi)) {
              best = x[u][e];
              bi = v;
            }
          }
      }
    }
  }
  for (int v = (vec_ok ? n_chunks * VE : 0) + tid; v < V; v += 1024) {  // tail (or the whole row when it is not 16-byte addressable)
    const float xv = load1<T>(logits, (int64_t)b * row_stride + v);
    if (v == banned || v == banned2 || v == banned3) continue;
    if (xv > best || (xv == best && v < bi)) {
      best = xv;
      bi = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (lane == 0) {
    smax[wid] = best;
    sidx[wid] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    const int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w)
      if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) {
        best = smax[w];
        bi = sidx[w];
      }
    if (bi == 0x7fffffff) bi = 0;
    int tok = bi;
    if (finished) {
      if (st_fin) tok = pad_id;
      else if (eos_id >= 0 && (tok == eos_id || tok == eos_id2 || tok == eos_id3)) finished[b] = 1;
    }
    next_ids[b] = tok;
    if (out_ids && step) {
      if (st_step < out_cap) out_ids[(int64_t)b * out_cap + st_step] = tok;
      step[b] = st_step + 1;
    }
    if (kv_len_full) kv_len_full[b] = st_full + 1;
    if (kv_len_sparse) kv_len_sparse[b] = st_sparse + st_dec;
  }
}

}  // namespace dl

using namespace dl;

extern "C" int dl_layernorm(const void*, const int32_t*, const void*, const void*, void*, int64_t, int, float, int, void*);
extern "C" int dl_attn_prefill(const void*, const void*, const void*, int64_t, int64_t, void*, int64_t, const int32_t*, int, int, int,
                               int, int, int, int, void*);

namespace {
struct VpLayout {
  int64_t x0, hs, y, qkv, ff, z, z1, z2, row_index, cu_img, total;
};
VpLayout vp_layout(int B, int n, int H, int D, int FF, int dtype) {
  const int64_t es = dtype == DL_F32 ? 4 : 2;
  const int64_t M = (int64_t)B * n;
  VpLayout L;
  int64_t o = 0;
  L.x0 = o; o += align256(M * H * es);
  L.hs = o; o += align256(M * D * es);
  L.y = o; o += align256(M * D * es);
  L.qkv = o; o += align256(M * 3 * D * es);
  L.ff = o; o += align256(M * FF * es);
  L.z = o; o += align256(M * D * es);
  L.z1 = o; o += align256(M * (D / 2) * es);
  L.z2 = o; o += align256(M * (D / 4) * es);
  L.row_index = o; o += align256(M * 4);
  L.cu_img = o; o += align256((int64_t)(B + 1) * 4);
  L.total = o;
  return L;
}
}  // namespace

extern "C" int64_t dl_vision_predictor_workspace_bytes(int B, int n_img, int H, int d_model, int dim_ff, int dtype) {
  if (B <= 0 || n_img <= 0) return 0;
  return vp_layout(B, n_img, H, d_model, dim_ff, dtype).total;
}

extern "C" int dl_vision_predictor(const void* hidden, const int32_t* cu_seqlens, const int32_t* img_start, int B, int n_img, int H,
                                   int d_model, int nhead, int dim_ff, const dl_vp_weights* w, void* workspace, void* logits_out,
                                   void* score_out, int dtype, void* stream) {
  DL_REQUIRE(hidden && cu_seqlens && img_start && w && workspace && logits_out && score_out, "dl_vision_predictor: NULL pointer");
  DL_REQUIRE(B > 0 && n_img > 0 && H > 0 && d_model > 0 && nhead > 0 && dim_ff > 0, "dl_vision_predictor: bad shape");
  DL_REQUIRE(d_model % nhead == 0 && d_model % 32 == 0, "dl_vision_predictor: d_model=%d nhead=%d unsupported", d_model, nhead);
  DL_REQUIRE(w->num_layers >= 0 && w->num_layers <= 4, "dl_vision_predictor: num_layers=%d unsupported", w->num_layers);
  DL_REQUIRE(dtype == DL_F32 || dtype == DL_F16 || dtype == DL_BF16, "dl_vision_predictor: unsupported dtype %d", dtype);
  const int hd = d_model / nhead;
  DL_REQUIRE(dtype == DL_F32 || hd == 64 || hd == 128, "dl_vision_predictor: head dim %d unsupported", hd);
  hipStream_t st = as_stream(stream);
  const VpLayout L = vp_layout(B, n_img, H, d_model, dim_ff, dtype);
  char* ws = reinterpret_cast<char*>(workspace);
  void *x0 = ws + L.x0, *hs = ws + L.hs, *y = ws + L.y, *qkv = ws + L.qkv, *ff = ws + L.ff, *z = ws + L.z, *z1 = ws + L.z1,
       *z2 = ws + L.z2;
  int32_t* row_index = reinterpret_cast<int32_t*>(ws + L.row_index);
  int32_t* cu_img = reinterpret_cast<int32_t*>(ws + L.cu_img);
  const int M = B * n_img, D = d_model;
  const int64_t es = dtype == DL_F32 ? 4 : 2;

  hipLaunchKernelGGL(vp_build_index_kernel, dim3((unsigned)((M + B + 256) / 256)), dim3(256), 0, st, cu_seqlens, img_start, B, n_img,
                     row_index, cu_img);
  int rc;
  if ((rc = dl_layernorm(hidden, row_index, w->ln_w, w->ln_b, x0, M, H, 1e-5f, dtype, stream))) return rc;
  linear_launch(x0, H, w->down_w, w->down_b, hs, D, nullptr, 0, M, D, H, DL_EPI_GELU, dtype, st);
  for (int j = 0; j < w->num_layers; ++j) {
    const dl_vp_block& k = w->blocks[j];
    if ((rc = dl_layernorm(hs, nullptr, k.norm1_w, k.norm1_b, y, M, D, 1e-5f, dtype, stream))) return rc;
    linear_launch(y, D, k.qkv_w, nullptr, qkv, 3 * D, nullptr, 0, M, 3 * D, D, 0, dtype, st);
    const char* qp = reinterpret_cast<const char*>(qkv);
    if ((rc = dl_attn_prefill(qp, qp + (int64_t)D * es, qp + 2 * (int64_t)D * es, 3 * D, 3 * D, y, D, cu_img, B, n_img, nhead, nhead, hd,
                              0, dtype, stream)))
      return rc;
    linear_launch(y, D, k.proj_w, k.proj_b, hs, D, hs, D, M, D, D, DL_EPI_RESIDUAL, dtype, st);
    if ((rc = dl_layernorm(hs, nullptr, k.norm2_w, k.norm2_b, y, M, D, 1e-5f, dtype, stream))) return rc;
    linear_launch(y, D, k.fc1_w, k.fc1_b, ff, dim_ff, nullptr, 0, M, dim_ff, D, DL_EPI_GELU, dtype, st);
    linear_launch(ff, dim_ff, k.fc2_w, k.fc2_b, hs, D, hs, D, M, D, dim_ff, DL_EPI_RESIDUAL, dtype, st);
  }
  DL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((vp_pool_concat_kernel<T>), dim3((unsigned)B, (unsigned)((D / 2 + 63) / 64)), dim3(1024), 0, st, hs, z, n_img, D);
  });
  linear_launch(z, D, w->out0_w, w->out0_b, z1, D / 2, nullptr, 0, M, D / 2, D, DL_EPI_GELU, dtype, st);
  linear_launch(z1, D / 2, w->out2_w, w->out2_b, z2, D / 4, nullptr, 0, M, D / 4, D / 2, DL_EPI_GELU, dtype, st);
  DL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((vp_head_kernel<T>), dim3((unsigned)((M + 255) / 256)), dim3(256), (size_t)(2 * (D / 4)) * sizeof(float), st, z2,
                       w->out4_w, w->out4_b, logits_out, score_out, M, D / 4);
  });
  DL_CHECK_LAUNCH("dl_vision_predictor");
  return DL_OK;
}

extern "C" int64_t dl_text_predictor_workspace_bytes(int B, int d_model) {
  return B > 0 && d_model > 0 ? (int64_t)B * (d_model + d_model / 2) * (int64_t)sizeof(float) : 0;
}

extern "C" int dl_text_predictor_decide(const void* x, int64_t x_row_stride, int B, int H, int d_model, const dl_tp_weights* w,
                                        void* workspace, float* logits_out, int32_t* decision, int dtype, void* stream) {
  DL_REQUIRE(x && w && workspace && decision, "dl_text_predictor_decide: NULL pointer");
  DL_REQUIRE(B > 0 && H > 0 && d_model > 0 && d_model % 32 == 0 && H % 8 == 0, "dl_text_predictor_decide: bad shape");
  DL_REQUIRE(H <= 12288, "dl_text_predictor_decide: H=%d too large for LDS staging", H);
  hipStream_t st = as_stream(stream);
  float* h1 = reinterpret_cast<float*>(workspace);
  const int D = d_model;
  DL_DISPATCH_DTYPE(dtype, T, {
    DL_REQUIRE(x_row_stride % Elem<T>::kVec == 0 && H % Elem<T>::kVec == 0, "dl_text_predictor_decide: H / row stride must be multiples of %d", Elem<T>::kVec);
    hipLaunchKernelGGL((tp_stage1_kernel<T>), dim3((unsigned)((D + 7) / 8), (unsigned)B), dim3(256), (size_t)H * sizeof(float), st, x,
                       x_row_stride, w->ln_w, w->ln_b, w->l1_w, w->l1_b, h1, H, D);
    float* a1 = h1 + (size_t)B * D;
    hipLaunchKernelGGL((tp_stage2a_kernel<T>), dim3((unsigned)((D / 2 + 7) / 8), (unsigned)B), dim3(256), 0, st, h1, w->l3_w, w->l3_b, a1, D);
    hipLaunchKernelGGL((tp_stage2b_kernel<T>), dim3((unsigned)B), dim3(512), (size_t)(D / 2 + D / 4 + 2) * sizeof(float), st, a1, w->l5_w,
                       w->l5_b, w->l7_w, w->l7_b, logits_out, decision, D);
  });
  DL_CHECK_LAUNCH("dl_text_predictor_decide");
  return DL_OK;
}

extern "C" int dl_decode_advance(const void* logits, int logits_dtype, int64_t logits_row_stride, int V, int B, int64_t* next_ids,
                                 int64_t* out_ids, int out_cap, int32_t* step, int32_t* finished, int eos_id, int eos_id2, int eos_id3, int pad_id,
                                 int32_t* kv_len_full, int32_t* kv_len_sparse, const int32_t* decision, int min_new_tokens,
                                 void* stream) {
  DL_REQUIRE(logits && next_ids, "dl_decode_advance: NULL pointer");
  DL_REQUIRE(B > 0 && V > 0, "dl_decode_advance: bad shape");
  hipStream_t st = as_stream(stream);
  DL_DISPATCH_DTYPE(logits_dtype, T, {
    hipLaunchKernelGGL((decode_advance_kernel<T>), dim3((unsigned)B), dim3(1024), 0, st, logits, logits_row_stride, V, next_ids, out_ids,
                       out_cap, step, finished, eos_id, eos_id >= 0 ? eos_id2 : -1, eos_id >= 0 ? eos_id3 : -1, pad_id, kv_len_full, kv_len_sparse, decision, min_new_tokens);
  });
  DL_CHECK_LAUNCH("dl_decode_advance");
  return DL_OK;
}
