/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the INTEGER part of the reference path (the checker for
 * dl_topk_select / dl_compact_tokens / dl_decode_advance's bookkeeping).  Never linked into the product.
 *
 *   orc_topk_keep      dynamic_modeling_llama.py:1898-1908  keep_index = sort(argsort(score, desc)[:k]), with the pinned
 *                      tie rule (equal scores: lower index first == stable descending sort; NaN sorts largest, -0 == +0)
 *   orc_compact_map    dynamic_modeling_llama.py:1917-1983  source row + position id of every surviving token
 *   orc_cache_advance  cache_utils.py:139-164               true_cache_length bookkeeping for one decoded token
 *   orc_get_chunk      llava/dynamic_eval/model_vqa_loader.py:30-38  contiguous data-parallel chunk [begin,end)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int gt(float a, float b) { /* "a sorts before b" in descending order; NaN is the largest */
  int an = isnan(a), bn = isnan(b);
  if (an || bn) return an && !bn;
  return a > b;
}

/* insertion sort of indices: stable by construction */
int orc_topk_keep(const float* score, int n, int k, int64_t* keep) {
  if (k < 0 || k > n) return -1;
  int* idx = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    int j = i;
    while (j > 0 && gt(score[i], score[idx[j - 1]])) { idx[j] = idx[j - 1]; --j; }
    idx[j] = i;
  }
  /* first k in descending order, then ascending index order (the second sort of DML:1902-1908) */
  char* sel = (char*)calloc((size_t)(n > 0 ? n : 1), 1);
  for (int i = 0; i < k; ++i) sel[idx[i]] = 1;
  int o = 0;
  for (int i = 0; i < n; ++i) if (sel[i]) keep[o++] = i;
  free(sel);
  free(idx);
  return 0;
}

/* one row: n_in tokens with an image span [img_start, img_start+n_img); output n_in-(n_img-k) tokens */
int orc_compact_map(int n_in, int img_start, int n_img, int k, const int64_t* keep, int32_t* src, int32_t* pos) {
  if (img_start < 0 || img_start + n_img > n_in || k > n_img) return -1;
  int o = 0;
  for (int j = 0; j < img_start; ++j, ++o) src[o] = pos[o] = j;                 /* left: arange(0, s)            */
  for (int j = 0; j < k; ++j, ++o) src[o] = pos[o] = img_start + (int)keep[j]; /* image: keep_index + s         */
  for (int j = img_start + n_img; j < n_in; ++j, ++o) src[o] = pos[o] = j;      /* right: arange(s+n_img, N)     */
  return o;
}

/* layers < sparse_layer always append; layers >= sparse_layer append iff decision (or no predictor: decision < 0) */
void orc_cache_advance(int64_t* len_full, int64_t* len_sparse, const int32_t* decision, int B) {
  for (int b = 0; b < B; ++b) {
    len_full[b] += 1;
    len_sparse[b] += decision ? (decision[b] != 0) : 1;
  }
}

void orc_get_chunk(int64_t n_items, int n_chunks, int k, int64_t* begin, int64_t* end) {
  int64_t chunk = (n_items + n_chunks - 1) / n_chunks; /* math.ceil(len/n) */
  int64_t b = chunk * k, e = b + chunk;
  if (chunk == 0 || b >= n_items) { *begin = *end = n_items; return; }
  *begin = b;
  *end = e < n_items ? e : n_items;
}
