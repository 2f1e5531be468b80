/*
 * dynllava.h -- C ABI of libdynllava_hip.so (gfx950 / MI355X only).
 *
 * The reference (Osilly/dynamic_llava) has NO native code: every entry point below replaces a sequence
 * of eager PyTorch ops on the reference's sparsified prefill+decode hot path.  The reference interface
 * each one replaces is cited as file:line relative to the reference root, with
 *   DML  = llava/model/language_model/dynamic_modeling_llama.py
 *   CU   = llava/model/language_model/cache_utils.py
 *   CTL  = llava/model/language_model/custom_transformer_layer.py
 *
 * Conventions
 *   - plain device pointers + explicit sizes; no torch / HIP types in signatures (stream is a void*
 *     holding a hipStream_t; NULL = the default stream).
 *   - `dtype`: DL_F32 / DL_F16 / DL_BF16 = element type of all "model dtype" buffers of that call.
 *   - every call only ENQUEUES work on `stream`; it never synchronises, allocates or frees, so all
 *     calls are hipGraph-capturable.  The caller owns every buffer including workspaces.
 *   - return value: 0 = enqueued, <0 = DL_ERR_* (nothing was enqueued); dl_last_error() returns a
 *     thread-local message for the last failing call on this host thread.
 *   - empty inputs: the row kernels, dl_linear, dl_rope_kv_write and dl_attn_prefill{,_cached} return DL_OK without enqueueing anything
 *     when their row / token count (rows, n, M, total, max_seqlen) is 0 -- also when the data pointers of such an empty buffer are
 *     NULL (an empty torch tensor has no storage), as the eager ops they replace are no-ops on empty tensors.
 *   - "packed varlen": B sequences concatenated along the token axis, row b = tokens
 *     [cu_seqlens[b], cu_seqlens[b+1]); cu_seqlens is int32[B+1] on the device.
 *   - KV slab layout (one K and one V slab per layer): [B][n_kv_heads][T_cap][head_dim], element
 *     strides given explicitly (slab_stride_b, slab_stride_h; the key-row stride is head_dim).
 */
#ifndef DYNLLAVA_H_
#define DYNLLAVA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_F32 0
#define DL_F16 1
#define DL_BF16 2

#define DL_ABI_VERSION 4

#define DL_OK 0
#define DL_ERR_ARG (-1)     /* bad argument (NULL pointer, unsupported size / dtype) */
#define DL_ERR_LAUNCH (-2)  /* hipLaunchKernel failed; see dl_last_error() */

int dl_version(void);               /* ABI version: 3 (round 5: + dl_pack_weight_tiles / dl_linear_packed; 2 = the round-4 signature changes of
                                       * dl_gemv_qkv_attn / dl_prompt_layout, dl_decode_block* removed).  hip_ops.load_library() refuses any other value. */
const char* dl_last_error(void);    /* thread-local, never NULL */
int dl_device_check(void);          /* 0 if the current HIP device is gfx950, else DL_ERR_ARG */

/* ---- F7: LlamaRMSNorm.forward, DML:134-139 ------------------------------------------------
 * out[r,:] = w * cast(x[r,:] * rsqrt(mean(x[r,:]^2) + eps))   (fp32 statistics; the cast to the
 * model dtype happens BEFORE the weight multiply, as in the reference).  x,out: [rows, H]. */
int dl_rmsnorm(const void* x, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream);

/* residual add (DML:1289 / DML:1295) fused with the following RMSNorm (DML:1273 / DML:1293):
 *   h[r,:] = cast(h[r,:] + delta[r,:])  (written back);  out = rmsnorm(h).  `w`/`out` may be NULL
 *   to perform only the residual add (last layer). */
int dl_add_rmsnorm(void* h, const void* delta, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream);

/* LlamaMLP activation, DML:328: out[r,i] = cast(cast(silu(g)) * u), g = gate_up[r,i], u = gate_up[r,I+i]. */
int dl_silu_mul(const void* gate_up, void* out, int64_t rows, int I, int dtype, void* stream);

/* ---- F8 + F10: apply_rotary_pos_emb DML:260-285 (+rotate_half DML:253-257) fused with the KV-cache
 * append of DynamicCachePlus.update / get_cache, CU:109-268.
 * qkv: packed [total, (n_heads + 2*n_kv_heads) * head_dim]; q and k are rotated IN PLACE, the rotated
 * k and the v row are also written into the slab at key index kv_base[b] + j (j = token offset in row b).
 * Position of token j of row b: pos[cu_seqlens[b] + j] if pos != NULL else pos_base[b] + j.
 * cos/sin: [n_pos, head_dim] tables in the model dtype (DML:181-184 rounds them to the model dtype).
 * Writes beyond T_cap are dropped.  Eviction never copies: a decode token is always written at slot
 * kv_len[b]; whether it stays is decided later by dl_decode_advance (the length simply does not grow). */
int dl_rope_kv_write(void* qkv, const void* cos_tab, const void* sin_tab, int n_pos,
                     const int32_t* cu_seqlens, const int32_t* pos, const int32_t* pos_base, const int32_t* kv_base,
                     void* k_slab, void* v_slab, int64_t slab_stride_b, int64_t slab_stride_h, int T_cap,
                     int B, int total, int n_heads, int n_kv_heads, int head_dim, int dtype, void* stream);

/* The same on a projection that arrives as fp32 PARTIAL SUMS (round 5: dl_linear_packed with DL_LP_PARTS on q|k|v, DML:1011-1013, k ranges not handed over inside
 * the GEMM launch): parts [n_parts][total][(n_heads + 2 n_kv_heads) head_dim] fp32, 16-byte aligned, 1 <= n_parts <= 8; a value is the sum of its partial sums in
 * part order rounded once to the model dtype -- what the GEMM would have stored.  qkv_out (same shape as dl_rope_kv_write's qkv, model dtype) is WRITTEN: rotated q
 * and k, v as is (the prefill attention reads all three from it); slab writes as above.  bf16 / f16. */
int dl_rope_kv_write_parts(void* qkv_out, const float* parts, int n_parts, const void* cos_tab, const void* sin_tab, int n_pos,
                           const int32_t* cu_seqlens, const int32_t* pos, const int32_t* pos_base, const int32_t* kv_base,
                           void* k_slab, void* v_slab, int64_t slab_stride_b, int64_t slab_stride_h, int T_cap,
                           int B, int total, int n_heads, int n_kv_heads, int head_dim, int dtype, void* stream);

/* ---- F9 (prefill): F.scaled_dot_product_attention as called at DML:1114-1122 with is_causal=True, and
 * CTL:164-169 (non-causal, inside VisionPredictor).  Packed varlen self-attention:
 * q/k/v element (token t, head h, dim e) at base[t*row_stride + h*head_dim + e] (k/v use kv head
 * h / (n_heads/n_kv_heads)); out[t*out_row_stride + h*head_dim + e].  scale = 1/sqrt(head_dim).
 * head_dim in {64, 128} (f16/bf16: MFMA path) or any multiple of 4 <= 256 (f32).  * Round 6: head_dim 64, non-causal rows of 257..608 tokens (the CLIP tower's 577, the vision predictor's 576) run on a whole-row kernel -- all keys of a head on
 * chip in fragment order, softmax in registers; same arithmetic (fp32 online softmax, P rounded to the dtype before P V), another summation order: rounding class.
 * Late round 6: head_dim 128, causal rows of 65..256 tokens (the decoder's compacted layers, DML:1061-1122) the same way -- one workgroup per (request, head), its
 * query tiles paired long / short over eight waves; two workgroups per head while there are at most 128 (request, head) pairs.  The head_dim-64 rows of 256 and more
 * (image, head) pairs (a batched CLIP tower) likewise get one 16-wave workgroup per pair. */
int dl_attn_prefill(const void* q, const void* k, const void* v, int64_t q_row_stride, int64_t kv_row_stride,
                    void* out, int64_t out_row_stride, const int32_t* cu_seqlens, int B, int max_seqlen,
                    int n_heads, int n_kv_heads, int head_dim, int causal, int dtype, void* stream);

/* ---- chunk on a cache (multi-round "new instruct", DML:2506-2521 / chunked prefill): causal attention of a packed chunk of
 * queries against the KV slab.  Row b's chunk keys must already sit in the slab at [kv_len[b], kv_len[b] + L_b) (dl_rope_kv_write
 * with kv_base = kv_len); query j of row b attends keys [0, kv_len[b] + j].  max_kv_len >= max_b (kv_len[b] + L_b) (host bound). */
int dl_attn_prefill_cached(const void* q, int64_t q_row_stride, const void* k_slab, const void* v_slab, int64_t slab_stride_b,
                           int64_t slab_stride_h, const int32_t* kv_len, void* out, int64_t out_row_stride,
                           const int32_t* cu_seqlens, int B, int max_seqlen, int max_kv_len, int n_heads, int n_kv_heads,
                           int head_dim, int dtype, void* stream);

/* ---- F9 (decode) + F11: one query token per row against the ragged KV slab (DML:1061-1122 with
 * CU:256-268).  Row b attends keys [0, kv_len[b] + extra) of its slab; q: [B, q_row_stride].
 * Split-KV: `n_splits` workgroups per (row, head); partials in `workspace` (float), merged by a
 * second kernel.  dl_attn_decode_workspace_bytes gives the required size. */
int64_t dl_attn_decode_workspace_bytes(int B, int n_heads, int head_dim, int n_splits);
int dl_attn_decode(const void* q, int64_t q_row_stride, const void* k_slab, const void* v_slab,
                   int64_t slab_stride_b, int64_t slab_stride_h, const int32_t* kv_len, int extra,
                   void* out, int64_t out_row_stride, void* workspace, int n_splits,
                   int B, int n_heads, int n_kv_heads, int head_dim, int dtype, void* stream);

/* ---- F8 + F10 + F9 (decode) in ONE launch: RoPE of q and of the new key (DML:260-285), the KV append of the new token at
 * slot kv_len[b] (CU:109-268) and the ragged attention over keys [0, kv_len[b]] (DML:1061-1122).
 * qkv: [B, qkv_row_stride] UN-rotated projection output (q heads | k heads | v heads), not modified.
 * pos_base[b]: RoPE position of the new token.  Same split-KV scheme as dl_attn_decode.
 * keys_in_flight = 64, 128 or 256: K/V rows a workgroup requests per loop trip (64 / 256: four waves x 4 / 16 rows per lane group; 128: eight
 * waves x 4 -- the form small single-split launches use; the batch-1 decode step is latency-bound).  The rows of trip i + 1 are requested
 * before trip i is consumed; K/V rows are loaded non-temporal.  chunk_keys > 0: split s owns keys [s*chunk_keys, (s+1)*chunk_keys) (the last
 * split also takes any remainder), so the K/V rows are requested before kv_len[b] has been read; 0: the kernel balances
 * ceil((kv_len[b]+1) / n_splits) keys per split itself.  Results are identical either way up to the merge order.
 * call_tag >= 0 (with keys_in_flight = 64, chunk_keys = 0, n_splits > 1 and a grid of <= 1024 workgroups): the splits are merged
 * INSIDE this launch by the workgroup of split 0 (partials travel through `workspace` as self-validating 8-byte granules; same
 * merge order and bits as the two-launch form).  call_tag must differ between consecutive calls that share `workspace` at the same
 * pos_base (the layer index does); -1: always the separate merge launch.  A granule is accepted when its tag equals
 * f(pos_base[b], call_tag): within one sequence positions only grow, so granules of earlier steps never match, but granules left by
 * ANOTHER sequence that visited the same position with the same call_tag would -- the caller must clear `workspace` (all-zero bytes: tag
 * 0 is never expected) whenever the sequence a row belongs to changes (model.py: once per generate(), every eager decode forward()). */
int dl_attn_decode_rope(const void* qkv, int64_t qkv_row_stride, const void* cos_tab, const void* sin_tab, int n_pos,
                        const int32_t* pos_base, const int32_t* kv_len, void* k_slab, void* v_slab,
                        int64_t slab_stride_b, int64_t slab_stride_h, int T_cap, void* out, int64_t out_row_stride,
                        void* workspace, int n_splits, int keys_in_flight, int chunk_keys, int call_tag, int B, int n_heads,
                        int n_kv_heads, int head_dim, int dtype, void* stream);

/* The same launch fed by the q|k|v projection's fp32 PARTIAL SUMS (dl_linear_packed with DL_LP_PARTS: one [B, row_stride] slice per k range, `part_stride` floats
 * apart) instead of its rounded output: every (row, head) workgroup adds the n_parts (1, 2 or 4) ranges of the 3 x head_dim values it reads in range order and rounds once to
 * `dtype` -- the value the projection's store epilogue would have written -- before RoPE (DML:260-285) / append (CU:109-268).  Decode batches of 16..32 rows: the
 * projection's two k ranges then need no hand-over inside its launch (22.2 -> 19.7 us at 32 rows x [12288, 4096]).  bf16 / fp16; 64 keys in flight (the form
 * batched decode steps use); everything else as dl_attn_decode_rope. */
int dl_attn_decode_rope_parts(const float* qkv_parts, int n_parts, int64_t part_stride, int64_t row_stride, const void* cos_tab, const void* sin_tab, int n_pos,
                              const int32_t* pos_base, const int32_t* kv_len, void* k_slab, void* v_slab,
                              int64_t slab_stride_b, int64_t slab_stride_h, int T_cap, void* out, int64_t out_row_stride,
                              void* workspace, int n_splits, int chunk_keys, int call_tag, int B, int n_heads,
                              int n_kv_heads, int head_dim, int dtype, void* stream);

/* ---- F2: top-k select, DML:1867 + 1898-1908.  score [B,n] in the model dtype (= log_softmax(...)[:,:,0]);
 * keep_idx [B,k] int64 ascending = the k largest scores; ties: the LOWER original index wins
 * (= stable descending sort; the reference's argsort is non-stable).  n <= 4096. */
int dl_topk_select(const void* score, int64_t* keep_idx, int B, int n, int k, int dtype, void* stream);

/* ---- F3 + F4: token compaction, DML:1917-1983.  Row b of the packed input has an image span
 * [img_start[b], img_start[b] + n_img); the output row keeps everything outside the span and the
 * k rows keep_idx[b,:] of it, in order.  pos_out[t] = original in-row index of output token t
 * (what the reference builds as position_ids).  h_in [total_in,H] -> h_out [total_out,H].
 * norm_w / x_out (both or neither): the RMSNorm of the compacted rows (the next layer's input_layernorm, DML:134-139) is written to
 * x_out [total_out,H] in the same pass -- the row is in registers anyway, and the result is bit-identical to dl_rmsnorm(h_out). */
int dl_compact_tokens(const void* h_in, void* h_out, const int64_t* keep_idx, const int32_t* cu_in,
                      const int32_t* cu_out, const int32_t* img_start, int32_t* pos_out, int B, int n_img, int k,
                      int total_out, int H, const void* norm_w, float eps, void* x_out, int dtype, void* stream);

/* ---- generic small linear layer used by the predictors: C = epilogue(A @ W^T + bias)
 * A [M,K] (row stride lda), W [N,K] (nn.Linear layout), bias [N] or NULL, C [M,N] (row stride ldc).
 * flags: DL_EPI_GELU -> C = gelu_erf(cast(A W^T + b)); DL_EPI_RESIDUAL -> C = cast(R + cast(A W^T + b))
 * with R [M,N] (row stride ldr; may alias C).  Roundings to the model dtype happen exactly where the
 * eager reference rounds (after the Linear, after the GELU, after the residual add). K % 8 == 0. */
#define DL_EPI_GELU 1
#define DL_EPI_RESIDUAL 2
int dl_linear(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc,
              const void* R, int64_t ldr, int M, int N, int K, int flags, int dtype, void* stream);

/* nn.LayerNorm(eps=1e-5) over the last dim, rows gathered through an optional index:
 * out[r,:] = LN(x[row_index ? row_index[r] : r, :]) * w + b */
int dl_layernorm(const void* x, const int32_t* row_index, const void* w, const void* b, void* out,
                 int64_t rows, int H, float eps, int dtype, void* stream);

/* ---- the CLIP ViT tower's glue between its (library) GEMMs: llava/model/multimodal_encoder/clip_encoder.py:53-71 runs
 * transformers' CLIPVisionModel, whose encoder layer is LN1 -> attention -> residual add -> LN2 -> fc1 -> QuickGELU -> fc2 ->
 * residual add (transformers modeling_clip.py CLIPEncoderLayer.forward / CLIPMLP.forward; a pinned dependency, not under
 * /root/reference).  dl_add_layernorm: h[r,:] = cast(h[r,:] + delta[r,:]) in place, then out[r,:] = LN(h[r,:]) * w + b
 * (w = b = out = NULL: the add only).  dl_quick_gelu: out = cast(x * cast(sigmoid(cast(1.702 * x)))), the three roundings of
 * the eager `input * torch.sigmoid(1.702 * input)`.  The attention itself is dl_attn_prefill (non-causal, head_dim 64). */
int dl_add_layernorm(void* h, const void* delta, const void* w, const void* b, void* out, int64_t rows, int H, float eps,
                     int dtype, void* stream);
int dl_quick_gelu(const void* x, void* out, int64_t n, int dtype, void* stream);

/* ---- F1: VisionPredictor.forward DML:1348-1359 (+ CTL:153-180, 320-323) followed by
 * log_softmax(...)[:, :, 0] DML:1867,1898.  All pointers are device pointers to nn.Module parameters
 * with the reference's state-dict layout (model.image_score_predictor.*). */
typedef struct dl_vp_block {
  const void *norm1_w, *norm1_b, *qkv_w, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} dl_vp_block;
typedef struct dl_vp_weights {
  const void *ln_w, *ln_b, *down_w, *down_b;      /* down_mlp.0 / down_mlp.1 */
  const void *out0_w, *out0_b, *out2_w, *out2_b, *out4_w, *out4_b; /* output_mlp.0/.2/.4 */
  int num_layers;                                  /* <= 4 */
  dl_vp_block blocks[4];
} dl_vp_weights;
int64_t dl_vision_predictor_workspace_bytes(int B, int n_img, int H, int d_model, int dim_ff, int dtype);
/* hidden: packed [total,H]; image rows of sequence b = cu_seqlens[b] + img_start[b] + [0,n_img).
 * logits_out [B,n_img,2] and score_out [B,n_img] in the model dtype. */
int dl_vision_predictor(const void* hidden, const int32_t* cu_seqlens, const int32_t* img_start, int B, int n_img,
                        int H, int d_model, int nhead, int dim_ff, const dl_vp_weights* w, void* workspace,
                        void* logits_out, void* score_out, int dtype, void* stream);

/* ---- F6: TextPredictor.forward DML:1385-1387 + decision DML:2388-2391.
 * x [B,H] (hidden state entering layer `sparse_layer`); logits_out [B,2] float (may be NULL);
 * decision[b] = logit0 > logit1 (strict, raw logits).  workspace: dl_text_predictor_workspace_bytes(B, d_model)
 * (1.5*B*d_model floats).  Three launches: LN + Linear(H -> d) over d/8 workgroups, Linear(d -> d/2) over d/16 workgroups, the
 * d/2 -> d/4 -> 2 tail in one workgroup per row. */
typedef struct dl_tp_weights {
  const void *ln_w, *ln_b, *l1_w, *l1_b, *l3_w, *l3_b, *l5_w, *l5_b, *l7_w, *l7_b; /* output_mlp.0/1/3/5/7 */
} dl_tp_weights;
int64_t dl_text_predictor_workspace_bytes(int B, int d_model);
int dl_text_predictor_decide(const void* x, int64_t x_row_stride, int B, int H, int d_model, const dl_tp_weights* w,
                             void* workspace, float* logits_out, int32_t* decision, int dtype, void* stream);

/* ---- decode-time weight streaming (B <= dl_gemv_max_batch rows): y[b,:] = cast(W @ x[b,:]), W [N,K] nn.Linear layout.
 * Replaces, for small B, the torch GEMMs of DML:1011-1013 (q/k/v_proj, fused), DML:1127 (o_proj), DML:328
 * (gate|up fused, down_proj) and DML:2709 (lm_head); the element-wise op in front of each becomes its prologue:
 *   DL_GEMV_PLAIN   x [B, x_row_stride] as given
 *   DL_GEMV_ADDNORM x = norm_w * cast(hn * rsqrt(mean(hn^2) + eps)), hn = cast(h_in + delta)  (DML:1289/1295 + 134-139);
 *                   hn is written to h_out (a buffer distinct from h_in; h_in, h_out: [B,K] contiguous).
 *                   delta == NULL: hn = h_in, h_out is not written.
 *   DL_GEMV_SILUMUL x = cast(cast(silu(g)) * u), g = x[b, 0:K], u = x[b, K:2K]   (DML:328)
 * mode may be OR-ed with DL_GEMV_OUT_SILU_PAIR: W is a fused gate|up weight [2I,K] and the EPILOGUE emits
 *   y[b,i] = cast(cast(silu(cast(W_i . x))) * cast(W_{I+i} . x)), y [B, I]   (DML:328 without materialising gate|up).
 * y: [B, y_row_stride].  K % 8 == 0. */
#define DL_GEMV_PLAIN 0
#define DL_GEMV_ADDNORM 1
#define DL_GEMV_SILUMUL 2
#define DL_GEMV_OUT_SILU_PAIR 16
int dl_gemv_max_batch(int K, int dtype);
/* grid_cap: workgroup cap of this call (0 = the tuned default, 4 workgroups per CU); there is no process-global tuning state. */
int dl_gemv(int mode, const void* W, int N, int K, const void* x, int64_t x_row_stride, const void* h_in, void* h_out,
            const void* delta, const void* norm_w, float eps, void* y, int64_t y_row_stride, int B, int dtype, int grid_cap,
            void* stream);

/* ---- in-place packing of the kept rows of a just-appended chunk: replaces the per-row slice / cat / zero-pad of CU:165-241 for the
 * multi-round "new instruct" call (DML:2506-2521).  For n_layers consecutive layer slabs starting at k_slab0 / v_slab0 (layer_stride
 * elements apart; each [B, nKV, T_cap, d] with the given strides), the chunk of T tokens sits at slots [kv_len[b], kv_len[b] + T); rows
 * with keep[b,t] != 0 (int32 [B,T]) are moved in order to [kv_len[b], ...).  kv_len itself is NOT modified (the caller adds the kept
 * counts once, on the device). */
int dl_kv_pack_rows(void* k_slab0, void* v_slab0, int64_t layer_stride, int n_layers, int64_t slab_stride_b, int64_t slab_stride_h,
                    int T_cap, const int32_t* keep, const int32_t* kv_len, int B, int n_kv_heads, int T, int head_dim, int dtype,
                    void* stream);

/* ---- prefill compaction of the instruct predictor (DML:2261-2375: `torch.where` on the keep decisions of the last instruct turn + gather/cat)
 * without a device->host copy.  ONE packed sequence of `total` rows [total, H]; rows [span0, span0 + n_span) survive where decision[j] != 0,
 * all other rows always.  h_out [total, H] / pos_out [total] receive the kept rows in order (pos_in NULL: the position is the row index);
 * cu_out int32[2] = {0, kept}; counts int64[2] = {kept, kept - 1} (device-side index of the last row).  Rows past `kept` are left untouched:
 * the caller keeps sizing its launches for `total` rows and passes cu_out to the kernels that need the true length. */
int dl_compact_rows_by_mask(const void* h_in, const int32_t* pos_in, const int32_t* decision, int span0, int n_span, int total, int H, void* h_out,
                            int32_t* pos_out, int32_t* cu_out, int64_t* counts, int dtype, void* stream);

/* ---- device-side prompt layout (replaces the host logic of ARCH:309-490 -- per row `.item()` on the image position ARCH:330-334, the
 * O(n) `torch.equal` scan for "USER:" ARCH:422-428 -- for the common eval case: every row holds exactly ONE image token (-200) and is not
 * padded).  input_ids int64 [B, W].  Outputs (all on the device, nothing is read back):
 *   seg      int32 [B, 8]   : img_pos, last_user (offset inside the instruct span of the last "USER:" id pair, 0 if none), n_images,
 *                             n_valid, reserved...
 *   text_src int64 [B*(W-1)]: flat positions (b*W + col) of the text tokens, row-major
 *   text_dst int64 [B*(W-1)]: their rows in the packed [B*(W-1+n_feat), H] embedding matrix
 *   img_dst  int64 [B*n_feat]: packed rows of the image features
 *   img_start int32 [B]     : = img_pos
 *   err      int32 [1]      : set to 1 + row when a row does not hold exactly one image token (checked by the caller at its next sync)
 * Round 4 -- width buckets (the eval loop of model_vqa_loader.py:123-196 presents a new prompt width nearly every call; one captured prefill
 * graph per WIDTH BUCKET serves them all):
 *   w_true   int32 [1] device scalar or NULL: only the first w_true[0] <= W columns of every row are the prompt; the sequences are packed at
 *            their true length n = w_true - 1 + n_feat, the padding columns gather column 0's token into the unused rows
 *            [B*n, B*(W-1+n_feat)) at the end of the packed matrix (nobody reads them).  NULL: w_true = W.
 *   n_drop   rows every sequence loses at layer `sparse_layer` (n_img - k of DML:1899-1901; 0 when nothing is compacted)
 *   cu_seqlens, cu_seqlens_sparse int32 [B+1], lens int32 [2, B], last_rows int64 [B] (each may be NULL): the prefill's device-side metadata
 *            at the TRUE lengths -- packed offsets before / after the compaction, the KV lengths the prefill leaves in the two length groups
 *            (cache_utils.py:139-149), and the packed row of every sequence's last token after the compaction. */
int dl_prompt_layout(const int64_t* input_ids, int B, int W, int n_feat, int image_token, int user_id0, int user_id1, int32_t* seg,
                     int64_t* text_src, int64_t* text_dst, int64_t* img_dst, int32_t* img_start, int32_t* err, const int32_t* w_true, int n_drop,
                     int32_t* cu_seqlens, int32_t* cu_seqlens_sparse, int32_t* lens, int64_t* last_rows, void* stream);

/* ---- decode-step bookkeeping (replaces HF greedy search's argmax + CU:153-164 / CU:197-199 host syncs):
 * next[b] = argmax_v logits[b,v] (lowest index on ties); finished rows emit pad_id;
 * out_ids[b, step[b]] = next[b]; ++step[b]; kv_len_full[b] += 1; kv_len_sparse[b] += decision ? decision[b] : 1.
 * logits: [B,V] in `logits_dtype` (DL_F32 or the model dtype).  step/finished: int32[B].  All state lives on
 * the device; out_ids / step / finished / kv_len_* / decision may be NULL to skip that piece of bookkeeping.
 * eos_id (-1: none), eos_id2, eos_id3 (-1: unused): the EOS set (HF accepts a list of ids).
 * min_new_tokens > 0: the EOS ids are excluded from the argmax while step[b] < min_new_tokens (HF MinNewTokensLengthLogitsProcessor). */
int dl_decode_advance(const void* logits, int logits_dtype, int64_t logits_row_stride, int V, int B,
                      int64_t* next_ids, int64_t* out_ids, int out_cap, int32_t* step, int32_t* finished,
                      int eos_id, int eos_id2, int eos_id3, int pad_id, int32_t* kv_len_full, int32_t* kv_len_sparse, const int32_t* decision,
                      int min_new_tokens, void* stream);

/* ---- decode GEMM for 5..32 rows: Y[M,N] = X[M,K] @ W[N,K]^T (nn.Linear without bias: DML:1011-1013, 1127, 328, 2709), M <=
 * dl_gemm_smallm_max_m().  Weight-streaming like dl_gemv, products on the matrix cores (X resident in LDS, weights HBM -> MFMA
 * operand registers).  bf16 / f16, fp32 accumulate, one rounding.  K % 256 == 0, N % 4 == 0; ldx / ldy: row strides (elements).
 * n_slices: split-K factor (0 = auto; the kernel may raise it so that the X slice fits LDS); when the effective factor is > 1
 * the fp32 partials go to `workspace` (dl_gemm_smallm_workspace_bytes(M, N, K, n_slices)) and a second launch adds them in slice
 * order.  variant: 1 = weight fragments loaded straight into the MFMA operand registers (wg_waves = 4 / 8, 0 = auto), 2 = coalesced
 * loads transposed through wave-private LDS (8 waves x 256-k chunks), 3 = the same with 16 waves x 128-k chunks, 0 = auto (2). */
int dl_gemm_smallm_max_m(void);
int64_t dl_gemm_smallm_workspace_bytes(int M, int N, int K, int n_slices, int variant);
int dl_gemm_smallm(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, void* workspace, int M, int N, int K,
                   int n_slices, int wg_waves, int variant, int defer_reduce, int dtype, void* stream);
/* defer_reduce != 0: leave the fp32 partials [slices][M][N] in `workspace` (always, even for one slice; Y may be NULL) for a consumer
 * that adds them itself -- dl_add_rmsnorm_parts (o_proj / down_proj -> residual add + RMSNorm) and dl_silu_mul_parts (gate|up ->
 * SiLU*up) -- which saves the reduce launch.  dl_gemm_smallm_slices: the effective slice count of such a call. */
int dl_gemm_smallm_slices(int M, int N, int K, int n_slices, int variant);
int dl_add_rmsnorm_parts(void* h, const float* parts, int n_slices, const void* w, void* out, int64_t rows, int H, float eps,
                         int dtype, void* stream);
int dl_silu_mul_parts(const float* parts, int n_slices, void* out, int64_t rows, int I, int dtype, void* stream);

/* ---- N5: training-time ops.
 * dl_attn_policy_fwd / _bwd replace `scaled_dot_product_attention_with_policy` + `softmax_with_policy` (DML:913-970) and their
 * autograd graph: self-attention whose softmax is gated by a differentiable per-key keep policy,
 *   A_ij = (exp(s_ij - max_j s_ij) * p'_ij + eps / N) / (sum_j exp(..) * p'_ij + eps),  p'_ij = policy[b,j] (1 on the diagonal),
 * without materialising the [B,H,N,N] tensors.  q, k, v: [B,H,L,d] views with element strides qkv_strides = {b, h, l} (d contiguous;
 * the same strides for all three); out / d_out / dq / dk / dv: [B,H,L,d] views with o_strides.  policy: fp32 [B,L].
 * Masking: causal != 0 (is_causal=True, DML:944-950) or `bias` = additive mask in the model dtype, [B or 1, 1, L, L] with element
 * strides bias_stride_b / bias_stride_row (DML:952-957; finfo.min as transformers builds it, or -inf), or neither.
 * row_max / row_denom: fp32 [B,H,L] saved by the forward for the backward.  n_for_eps: the N of eps / N (the padded key count).
 * workspace: dl_attn_policy_workspace_floats(B, H, L, d) floats, the same buffer for both calls.
 * dpolicy_heads: fp32 [B,H,L]; dpolicy[b,j] = sum_h dpolicy_heads[b,h,j] (summed by the caller: deterministic, no atomics).
 * bf16 / f16, head_dim 64 or 128; dropout_p must be 0 (attention_dropout = 0.0 in every shipped config).  The O(eps) gradient through
 * max_j is not propagated. */
int64_t dl_attn_policy_workspace_floats(int B, int H, int L, int head_dim);
int dl_attn_policy_fwd(const void* q, const void* k, const void* v, const int64_t* qkv_strides, void* out, const int64_t* o_strides,
                       const float* policy, const void* bias, int64_t bias_stride_b, int64_t bias_stride_row, float* row_max,
                       float* row_denom, float* workspace, int B, int H, int L, int head_dim, int causal, float scale, float eps,
                       int n_for_eps, int dtype, void* stream);
int dl_attn_policy_bwd(const void* q, const void* k, const void* v, const int64_t* qkv_strides, const void* out, const void* d_out,
                       void* dq, void* dk, void* dv, const int64_t* o_strides, const float* policy, const void* bias,
                       int64_t bias_stride_b, int64_t bias_stride_row, const float* row_max, const float* row_denom,
                       float* dpolicy_heads, float* workspace, int B, int H, int L, int head_dim, int causal, float scale, float eps,
                       int n_for_eps, int dtype, void* stream);
/* Gumbel hard keep mask, DML:1868-1876: keep = F.gumbel_softmax(log_probs, tau, hard=True)[..., 0] * prev_decision with the noise
 * drawn by the caller (`-empty_like(log_probs).exponential_().log()`, torch's generator).  log_probs / gumbels / y_soft /
 * d_log_probs: [n,2]; prev_decision / keep / d_keep / d_prev (may be NULL): [n]; all in `dtype` (f32 / f16 / bf16), rounded where the
 * eager ops round. */
int dl_gumbel_hard_keep_fwd(const void* log_probs, const void* gumbels, const void* prev_decision, void* keep, void* y_soft, int64_t n,
                            float tau, int dtype, void* stream);
int dl_gumbel_hard_keep_bwd(const void* d_keep, const void* prev_decision, const void* y_soft, void* d_log_probs, void* d_prev, int64_t n,
                            float tau, int dtype, void* stream);

/* ---- split-K projection for the prefill's narrow nn.Linear calls (o_proj DML:1127, down_proj DML:328 at M = 100..256 rows):
 * parts[s][m][n] = sum over K slice s of A[m,k] W[n,k], fp32, s < n_slices (<= K / 128); the consumer adds the slices in order
 * (dl_add_rmsnorm_parts: residual add + RMSNorm; dl_gemm_smallm_reduce semantics).  A [M,K] row stride lda, W [N,K] contiguous.
 * bf16 / f16; N % 4 == 0, K % 8 == 0.  parts: n_slices * M * N floats. */
int dl_linear_splitk(const void* A, int64_t lda, const void* W, float* parts, int M, int N, int K, int n_slices, int dtype, void* stream);

/* ---- a batch-1 decode layer's q|k|v projection (DML:1011-1013, with the residual add + input RMSNorm prologue of dl_gemv's ADDNORM mode)
 * AND the attention that consumes it (dl_attn_decode_rope with one split: RoPE DML:260-285, KV append CU:109-268, ragged attention
 * DML:1061-1122) in ONE launch.  The attention workgroups (n_splits per head) are the last blocks of the grid: they request their K/V rows at once
 * and then wait for their 3 x head_dim projection outputs, which the weight-streaming workgroups publish as 8-byte {tag, value} granules
 * beside the ordinary stores to `qkv`.  B = 1 only.
 * W: [(n_heads + 2 n_kv_heads) head_dim, K].  granules: dl_gemv_qkv_attn_workspace_bytes() bytes, zeroed once per request (tags are made of
 * pos_base[0] and call_tag, 0..255: distinct for every (step, layer) of a request).  err_flag (may be NULL): bit 0 is set, and the output poisoned
 * with NaN, if a consumer gave up waiting.  out: [n_heads * head_dim].
 * Round 4: the new token is folded in AFTER the merge of the slab keys' partials (rounding class of dl_attn_decode_rope, not its bits; the
 * projection row, the residual stream and the appended K/V row stay bit-identical).  n_splits (1..4): attention workgroups per head; with more than one,
 * every workgroup takes 128 of the head's slab keys (the last one the rest) -- its whole share is in registers before q arrives -- and the head's
 * first workgroup merges the others' (M, L, O) partials, handed over as granules, in split order before it folds the new token in. */
int64_t dl_gemv_qkv_attn_workspace_bytes(int n_heads, int n_kv_heads, int head_dim);
int dl_gemv_qkv_attn(const void* W, int K, const void* h_in, void* h_out, const void* delta, const void* norm_w, float eps, void* qkv,
                     const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* pos_base, const int32_t* kv_len, void* k_slab,
                     void* v_slab, int64_t slab_stride_b, int64_t slab_stride_h, int T_cap, void* out, void* granules, int call_tag,
                     int32_t* err_flag, int n_splits, int n_heads, int n_kv_heads, int head_dim, int dtype, int grid_cap, void* stream);

/* ---- the gate|up projection of layer `sparse_layer` at decode batch 1 (dl_gemv ADDNORM | OUT_SILU_PAIR: DML:1289 + DML:134-139 + DML:328) AND the
 * text predictor (dl_text_predictor_decide: DML:1385-1387, 2388-2391) on the residual stream entering that layer -- the h_in of this launch --
 * in ONE launch: the predictor's three stages are the first workgroups of the grid and hand their outputs on as granules; nothing in the
 * layer chain waits for them.  Bit-identical to the separate calls.  tp_workspace: dl_text_predictor_workspace_bytes(1, d_model);
 * granules: dl_gemv_gu_tp_workspace_bytes(d_model), zeroed once per request (tag = pos_base[0], call_tag); err_flag (may be NULL): bit 1 is
 * set (and decision[0] = 1, keep) if a stage gave up waiting. */
int64_t dl_gemv_gu_tp_workspace_bytes(int d_model);
int dl_gemv_gu_tp(const void* W, int N, int K, const void* h_in, void* h_out, const void* delta, const void* norm_w, float eps, void* y,
                  const dl_tp_weights* tp, int d_model, void* tp_workspace, float* logits_out, int32_t* decision, const int32_t* pos_base,
                  void* granules, int call_tag, int32_t* err_flag, int dtype, int grid_cap, void* stream);

/* ---- weight-streaming projection on a PRE-PACKED weight copy (round 5): the decoder's nn.Linear calls at a few hundred rows or fewer -- DML:1011-1013
 * (q|k|v), DML:1127 (o_proj), DML:328 (gate / up / down) in the post-compaction prefill layers (M = N' = 117..192) and decode steps of 4..32 rows.
 * dl_pack_weight_tiles writes W [N,K] (nn.Linear layout, contiguous) once in matrix-core operand order: 16-neuron x 32-k fragments of one contiguous
 * KiB each, Wp[((u S + s) 64 + lane) 8 + j] = W[16 u + lane % 16][32 s + 8 (lane / 16) + j], S = K / 32 (u: unit, s: slab); with gate_up_pairs != 0,
 * W = [gate; up] ([2 I, K]) and unit 2 j holds gate neurons [16 j, 16 j + 16), unit 2 j + 1 the matching up neurons.  N % 16 == 0, K % 64 == 0; bf16 / f16;
 * Wp: dl_packed_weight_bytes(N, K, dtype) bytes (= N K 2; -1 for unsupported shapes), 16-byte aligned, not aliasing W.
 * dl_pack_x_tiles: the activations in the same order -- Xp[step][tile][k half][lane][8], step = k / 64, tile = row / 16, lane = 16 ((k % 32) / 8) + row % 16,
 * 4 ceil(ceil(M / 16) / 4) tiles (rows past M repeat row M - 1) -- so that a consumer wave's fragment is one contiguous KiB too; dl_packed_x_bytes(M, K) bytes.
 * dl_linear_packed: Y[M,N] = X[M,K] W^T, fp32 accumulation, M <= 256.  x_packed: 0 = X row-major [M,K] with row stride ldx >= K elements (16-byte aligned
 * rows), 1 = dl_pack_x_tiles layout (ldx ignored).
 *   k_split (1..8): that many workgroups share a set of units, each a contiguous k range; all but the last hand their fp32 tiles to the last through
 *   `workspace` (dl_linear_packed_workspace_bytes bytes, 256-byte aligned, ZEROED ONCE by the caller: the kernel leaves its flag words zero), which adds
 *   them in range order and runs the epilogue -- a fixed summation order for a given (units_per_workgroup, k_split): deterministic, one rounding per
 *   output.  err_flag (may be NULL): bit 3 is set if a reducing wave gave up waiting (never on a healthy launch).
 *   epilogue 0: Y[m][n] = cast(acc); 1 (gate_up_pairs packing): Y[m][i] = cast(cast(silu(cast(gate_i))) * cast(up_i)), Y: [M, N / 2] (DML:328, the
 *   roundings of F.linear followed by dl_silu_mul); 2: Y[m][n] = cast(resid[m][n] + cast(acc)) (DML:1289 / 1295; resid may alias Y).
 *   3 (DL_LP_PARTS): Y is an fp32 buffer [k_split][M][ldy]: range r of k_split writes its partial sums to slice r (no hand-over, no workspace) for a
 *   consumer that adds the slices in order (dl_add_rmsnorm_parts; dl_linear_splitk's contract).
 *   epilogue | DL_LP_Y_PACKED (epilogues 0 and 1): Y is written in dl_pack_x_tiles order (dl_packed_x_bytes(M, n_out) bytes, n_out % 64 == 0) -- the
 *   x_packed input of the next dl_linear_packed (gate|up + SiLU * up feeding down_proj).
 *   units_per_workgroup: 0 = chosen here (one workgroup per CU), else 1, 2, 3, 4, 6, 8. */
#define DL_LP_STORE 0
#define DL_LP_SILU_PAIR 1
#define DL_LP_RESID 2
#define DL_LP_PARTS 3
#define DL_LP_Y_PACKED 16
/* dl_rmsnorm / dl_add_rmsnorm / dl_add_rmsnorm_parts with `out` written in dl_pack_x_tiles order (dl_packed_x_bytes(rows, H) bytes): the producer of a
 * dl_linear_packed(x_packed = 1) call writes its rows where the consumer's fragments expect them -- no separate packing launch.  rows <= 256,
 * H % 64 == 0, bf16 / f16; w and out are required (the residual-only form has no output to pack); same arithmetic, same roundings. */
int dl_rmsnorm_packed(const void* x, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream);
int dl_add_rmsnorm_packed(void* h, const void* delta, const void* w, void* out, int64_t rows, int H, float eps, int dtype, void* stream);
int dl_add_rmsnorm_parts_packed(void* h, const float* parts, int n_slices, const void* w, void* out, int64_t rows, int H, float eps, int dtype,
                                void* stream);
int64_t dl_packed_weight_bytes(int N, int K, int dtype);
int dl_pack_weight_tiles(const void* W, void* Wp, int N, int K, int gate_up_pairs, int dtype, void* stream);
int64_t dl_packed_x_bytes(int M, int K);
int dl_pack_x_tiles(const void* X, int64_t ldx, void* Xp, int M, int K, int dtype, void* stream);
int64_t dl_linear_packed_workspace_bytes(int M, int N, int K, int epilogue, int units_per_workgroup, int k_split);
int dl_linear_packed(const void* X, int64_t ldx, int x_packed, const void* Wp, void* Y, int64_t ldy, const void* resid, int64_t ldr, int M, int N, int K,
                     int epilogue, int units_per_workgroup, int k_split, void* workspace, int32_t* err_flag, int dtype, void* stream);
/* The same launch with a per-wave timeline (tools/lp_timeline.py): every wave writes s_memtime stamps -- entry, first ring step landed, k loop done, hand-over done,
 * stores done -- to stamps[(workgroup * 10 + wave) * 8 + k]; stamps: 80 int64 per workgroup, zeroed by the caller.  Results are unchanged. */
int dl_linear_packed_stamped(const void* X, int64_t ldx, int x_packed, const void* Wp, void* Y, int64_t ldy, const void* resid, int64_t ldr, int M, int N,
                             int K, int epilogue, int units_per_workgroup, int k_split, void* workspace, int32_t* err_flag, int64_t* stamps, int dtype,
                             void* stream);

/* ---- tiled MFMA GEMM for the vision side's nn.Linear calls at M = 577 B rows (round 6): the CLIP ViT-L/14-336 encoder layer's projections
 * (llava/model/multimodal_encoder/clip_encoder.py:53-71 runs transformers' CLIPEncoderLayer: q|k|v, out_proj, fc1 + QuickGELU, fc2 -- a pinned
 * dependency, not under /root/reference), the mlp2x_gelu projector (llava/model/multimodal_projector/builder.py:172-179) and the vision predictor's
 * linears (DML:1348-1359, CTL:153-180).  Y[M,N] = X[M,K] W^T on dl_pack_weight_tiles' copy of W [N,K] (gate_up_pairs = 0), fp32 accumulation.
 *   x_packed: 0 = X row-major (row stride ldx >= K elements, 16-byte aligned rows); 1 = fragment order Xp[step = k / 64][tile = row / 16 of ceil(M / 16)]
 *   [k half][lane = 16 ((k % 32) / 8) + row % 16][8] (dl_pack_x_rows, dl_layernorm_rows / dl_add_layernorm_rows, or a previous call's y_packed output;
 *   dl_tiles_x_bytes(M, K) bytes; rows past M inside the last tile are never read into a stored result).  NOTE: the tile count is ceil(M / 16), not
 *   dl_pack_x_tiles' multiple of four.
 *   epilogue 0: Y = cast(acc + bias) (bias may be NULL); 1: QuickGELU on the rounded sum, dl_quick_gelu's three roundings; 2: nn.GELU() (erf) on the rounded
 *   sum; 3 (DL_LT_PARTS): Y is an fp32 buffer [k_split][M][ldy] of partial sums, range r of the k_split ranges in slice r, bias must be NULL (the consumer
 *   -- dl_add_layernorm_parts -- adds the slices in order, then the bias).  k_split > 1 only with epilogue 3.
 *   y_packed (epilogues 0..2, N % 64 == 0): Y is written in fragment order for the next call.
 *   tile_shape: 0 = chosen here (one round of workgroups over the 256 CUs where the shape allows), else 100 TM + 10 WN + NUW: TM row tiles of 16 rows x
 *   WN consumer waves x NUW units of 16 neurons per workgroup (built: 542, 532, 522, 512, 521, 541, and for two to four images 1042, 1032, 1041; + 20000 = five instead of three steps of weight
 *   fragments in flight -- 542, 532, 521, 541; a library built with -DDL_LT_MEASURE also holds + 10000 = the weights through the LDS ring as well -- 542, 532, 521); with epilogue 3 only:
 *   142, 242, 342, 442, 642, 742, 842 (+ 20142, 20242, 20842) -- the decoder's o_proj (DML:1127) at <= 256 rows as TM row tiles x 8 units x k ranges.
 *   Deterministic: one fp32 accumulation per output in k order per range; row-position invariant. */
#define DL_LT_BIAS 0
#define DL_LT_QGELU 1
#define DL_LT_GELU 2
#define DL_LT_PARTS 3
int64_t dl_tiles_x_bytes(int M, int K);
/* LayerNorm launches around dl_linear_tiles (transformers CLIPEncoderLayer.forward: LN1 -> attention -> residual add -> LN2 -> MLP -> residual add), a wave
 * per row, bf16 / f16: dl_layernorm_rows = dl_layernorm without the row gather; dl_add_layernorm_rows = dl_add_layernorm; dl_add_layernorm_parts:
 * h[r,:] = cast(h[r,:] + cast(sum_s parts[s][r][:] + bias)) in place -- parts: the fp32 [n_slices][rows][H] output of dl_linear_tiles(DL_LT_PARTS), added in
 * slice order; bias (may be NULL): the Linear's bias; one rounding of the sum = what F.linear returns -- then out[r,:] = LN(h[r,:]) * w + b
 * (w = b = out = NULL: the residual add only).  out_packed != 0: out is written in dl_linear_tiles' fragment order (dl_tiles_x_bytes(rows, H) bytes,
 * H % 64 == 0) -- the x_packed input of the next projection.  H / 8 <= 256 or H == 4096.  Same arithmetic as dl_layernorm (fp32 statistics, two passes over
 * the registers), another summation tree: rounding class, not bits. */
int dl_layernorm_rows(const void* x, const void* w, const void* b, void* out, int64_t rows, int H, float eps, int out_packed, int dtype, void* stream);
int dl_add_layernorm_rows(void* h, const void* delta, const void* w, const void* b, void* out, int64_t rows, int H, float eps, int out_packed, int dtype,
                          void* stream);
int dl_add_layernorm_parts(void* h, const float* parts, int n_slices, const void* bias, const void* w, const void* b, void* out, int64_t rows, int H, float eps,
                           int out_packed, int dtype, void* stream);
int dl_pack_x_rows(const void* X, int64_t ldx, void* Xp, int M, int K, int dtype, void* stream);
int dl_linear_tiles(const void* X, int64_t ldx, int x_packed, const void* Wp, const void* bias, void* Y, int64_t ldy, int y_packed, int M, int N, int K,
                    int epilogue, int tile_shape, int k_split, int dtype, void* stream);
/* The same launch with a per-wave timeline: stamps[(workgroup * 8 + wave) * 8 + k], k = entry / first step landed / k loop done / stores done / half of the
 * k loop done; zeroed by the caller.  Results are unchanged.  Measurement only: epilogue | (w << 8) makes step t read the operands of step t % w (the buffers
 * then only need w steps of K): the second pass over a short K is served by L2 (tools/bench_linear_tiles.py --stamps). */
int dl_linear_tiles_stamped(const void* X, int64_t ldx, int x_packed, const void* Wp, const void* bias, void* Y, int64_t ldy, int y_packed, int M, int N,
                            int K, int epilogue, int tile_shape, int k_split, int64_t* stamps, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNLLAVA_H_ */
