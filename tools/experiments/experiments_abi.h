/* Declarations of the two experiment entry points that were built, measured and NOT shipped (they are not part of include/dynllava.h).
 * Kept so that the archived sources still compile:
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I dynamic_llava_amd/csrc -I include -I tools/experiments \
 *         -c tools/experiments/decode_block_r03.hip      (or linear_stream_r04.hip)
 * Results: profiles/r03_block_timeline.txt, profiles/r03_lds_dma_raw.txt (decode block); profiles/r04_prefill_stream_gemm_experiment.txt (linear stream). */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- the weight-streaming part of a batch-1 decode layer as ONE launch on the LDS-DMA engine (csrc/decode_block.hip): up to 4 chained
 * GEMV phases, y_i = W_i x_i, where phase 0 reads its input vector from memory (x_in: the attention output) and phase i > 0 consumes
 * phase i-1's output inside the launch (8-byte {tag, bf16 pair} granules in `sync_buf`); the last phase writes `out` to memory.  The
 * decode layer is o_proj (DML:1127) -> gate|up with residual add + RMSNorm prologue and SiLU*up epilogue (DML:1289-1295, 134-139, 328)
 * -> down_proj (DML:328) -> the next layer's q|k|v with add + norm (DML:1011-1013; lm_head DML:2709 after the last layer).  Arithmetic
 * and order are dl_gemv's: the result is bit-identical to the chain of dl_gemv launches it replaces.
 *   flags: DL_BLK_ADDNORM   x = norm_w * rmsnorm(h + delta): delta = the phase's input vector, h = h_in (memory) for the first such
 *                           phase of a block and the block's own running residual afterwards; h_out (may be NULL): the updated
 *                           residual stream is also written to memory (by one workgroup)
 *          DL_BLK_SILU_PAIR W = gate|up [2 I, K]: out[o] = cast(cast(silu(y_o)) * y_{I+o}), I = N / 2 outputs
 * N even, K % 8 == 0, bf16 / f16, batch 1.  pos_base[0] (the new token's position) and call_tag (0..255, e.g. the layer) make the
 * granule tags of this call unique among calls that reuse `sync_buf` (dl_decode_block_sync_bytes(max K); clear it once per request).
 * One 256-thread workgroup per CU (n_workgroups: 0 = all CUs), all of which must be resident: every in-kernel wait is bounded
 * (spin_limit, 0 = default) and a give-up ORs a code into *err_flag (may be NULL).  debug_stamps: NULL; debug_mode: 0 (measurement modes of
 * tools/bench_block.py: 1 = no arithmetic, 2 = loader wave alone; the outputs are then meaningless). */
#define DL_BLK_ADDNORM 1
#define DL_BLK_SILU_PAIR 2
typedef struct dl_block_phase {
  const void* W;      /* [N, K] row-major */
  const void* norm_w; /* [K] (DL_BLK_ADDNORM) */
  void* out;          /* last phase: [N] (or [N/2] with DL_BLK_SILU_PAIR); NULL otherwise */
  const void* x_in;   /* phase 0: input vector [K]; NULL otherwise */
  const void* h_in;   /* first DL_BLK_ADDNORM phase: residual stream [K] */
  void* h_out;        /* DL_BLK_ADDNORM: updated residual stream [K], or NULL */
  int32_t N, K, flags, reserved;
} dl_block_phase;
int64_t dl_decode_block_sync_bytes(int max_k);
int dl_decode_block(const dl_block_phase* phases, int n_phases, void* sync_buf, int64_t sync_bytes, const int32_t* pos_base, int call_tag,
                    float eps, int32_t* err_flag, int n_workgroups, int spin_limit, void* debug_stamps, int debug_mode, int dtype, void* stream);


/* round 4: Y[M <= 192, N] = X W^T as a weight stream with an MFMA consumer (three designs; this file holds the third) */
#define DL_STREAM_SILU_PAIR 1
int dl_linear_stream(const void* A, int64_t lda, const void* W, void* out, int64_t ldc, int M, int N, int K, int n_slices, int flags, int variant,
                     int dtype, void* stream);

#ifdef __cplusplus
}
#endif
