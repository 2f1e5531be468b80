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
#include <mutex>

#include "attn_decode_body.h"
#include "granule.h"

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

// keys_in_flight 128 = eight waves x U = 4 (16 waves were measured slower: 10.7 vs 8.9 us at T = 226).
// U = key rows each lane group requests per loop trip.  U = 16 puts 256 keys (NW = 4) in flight per workgroup in one round trip:
// at decode batch 1 the kernel's time is the number of dependent HBM round trips, not bytes.

// FUSED: the RoPE of q and of the new key (DML:260-285) and the KV-slab append (CU:109-268) happen inside the attention
// kernel: q|k|v are read un-rotated from the projection output, the new token's rotated key / value are used from
// registers by the one lane group that owns key index kv_len[b] and written to slab slot kv_len[b] for later steps.
// The body lives in attn_decode_body.h (shared with the persistent decode step).
// INK ("in-kernel combine", FUSED only): the partials travel as 8-byte {tag, value} granules (granule.h) and the workgroup of split 0
// merges them itself -- same merge code, same order, same bits as attn_decode_combine_kernel -- instead of a second launch (4.7 us +
// a ~1.2 us boundary per layer at batch 1).  tag = f(position of the new token, call_tag): the writes that precede a launch in the same
// slot come from the previous layer / step, so a stale granule never carries the expected tag.  Needs every workgroup of the grid
// resident (the host selects INK only for grids the device can hold at once: ink_resident_capacity); the wait is bounded and poisons the
// output with NaN on give-up.
template <typename T, int D, int NW, bool FUSED, int U, bool INK = false, int NP = 0>
__global__ __launch_bounds__(NW * 64) void attn_decode_split_kernel(
    const void* __restrict__ q_, int64_t q_row_stride, const void* k_slab_, const void* v_slab_, int64_t stride_b, int64_t stride_h,
    const int32_t* __restrict__ kv_len, int extra, float* __restrict__ ws, void* __restrict__ out_, int64_t out_row_stride, int n_rep,
    float scale, const void* __restrict__ cos_, const void* __restrict__ sin_, int n_pos, const int32_t* __restrict__ pos_base, int T_cap,
    int n_kv_heads, int chunk_keys, int call_tag = 0, int64_t part_stride = 0) {
  using St = AttnSplitState<T, D, NW, U>;
  using S = typename Elem<T>::storage;
  constexpr int NG = St::NG;
  __shared__ float sm_m[NG], sm_l[NG];
  __shared__ float sm_o[NG * D];

  DL_STAMP(0, false);
  const int split = blockIdx.x, n_splits = gridDim.x, h = blockIdx.y;
  const int n_heads = gridDim.y;
  const int tid = threadIdx.x;
  // Ragged batches, longest row first (round 6): workgroups are dispatched in blockIdx order (x, then y, then z), so slice z of the grid takes the row with
  // the z-th LARGEST length (ties: lower row index) instead of row z -- the long rows start first and the short ones fill the tail, whatever order the caller's
  // batch is in (a batch of 32 requests with 200..900 keys otherwise ends with a few CUs finishing their 900-key rows alone).  Every wave ranks the <= 64
  // lengths itself (B readlanes); which slice computes a row does not change a bit of the result.
  int b = blockIdx.z;
  int T_ranked = -1;  // the chosen row's length, when the ranking below has it in a register already (saves the dependent kv_len[b] load: one memory round trip
  if (gridDim.z > 1 && gridDim.z <= 64) {  // in front of the first K/V request of every workgroup)
    const int lane = tid & 63, Bn = (int)gridDim.z;
    const int my = lane < Bn ? kv_len[lane] : -1;
    int rank = 0;
    for (int j = 0; j < Bn; ++j) {
      const int lj = __builtin_amdgcn_readlane(my, j);
      rank += (lj > my || (lj == my && j < lane)) ? 1 : 0;
    }
    const unsigned long long m = __ballot(lane < Bn && rank == (int)blockIdx.z);
    b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
    T_ranked = __builtin_amdgcn_readlane(my, b);
  }
  const int kvh = h / n_rep;
  // the new token's q|k|v row: elements of T, or (dl_attn_decode_rope_parts) the fp32 partial sums of the projection's k ranges
  constexpr int64_t esz = NP > 0 ? 4 : (int64_t)sizeof(S);
  const char* row = reinterpret_cast<const char*>(q_) + (int64_t)b * q_row_stride * esz;
  const int T_old = T_ranked >= 0 ? T_ranked : kv_len[b];
  St st;
  attn_split_issue<T, D, NW, FUSED, U>(st, tid, k_slab_, v_slab_, stride_b, stride_h, T_old, extra, b, kvh, split, n_splits, T_cap, chunk_keys);
  float M, L, O;
  attn_split_finish<T, D, NW, FUSED, U, AttnNoWait, NP>(st, tid, row + (int64_t)h * D * esz, row + (int64_t)(n_heads + kvh) * D * esz,
                                        row + (int64_t)(n_heads + n_kv_heads + kvh) * D * esz, cos_, sin_, n_pos, FUSED ? pos_base[b] : 0, scale,
                                        h % n_rep == 0, T_cap, sm_m, sm_l, sm_o, M, L, O, AttnNoWait(), part_stride);
  if constexpr (INK) {
    extern __shared__ __attribute__((aligned(16))) float comb[];  // [n_splits][D + kAttnPartPad]: the layout attn_split_merge reads
    constexpr int PG = D + 2;                                      // granules of one partial: M, L, O[D]
    const uint32_t tag = ((((uint32_t)pos_base[b] & 0x7fffffu) << 8) | ((uint32_t)call_tag & 0xffu)) + 1u;
    u64_t* gws = reinterpret_cast<u64_t*>(ws) + ((int64_t)b * n_heads + h) * n_splits * PG;
    if (n_splits == 1) {
      if (tid < D) store1<T>(out_, (int64_t)b * out_row_stride + (int64_t)h * D + tid, L > 0.f ? O / L : 0.f);
      return;
    }
    if (split != 0) {
      if (tid < D) {
        u64_t* pr = gws + (int64_t)split * PG;
        gr_store(pr + 2 + tid, tag, __float_as_uint(O));
        if (tid == 0) {
          gr_store(pr, tag, __float_as_uint(M));
          gr_store(pr + 1, tag, __float_as_uint(L));
        }
      }
      return;
    }
    // split 0: own partial straight into the staging area, the others as they arrive
    if (tid < D) {
      comb[kAttnPartPad + tid] = O;
      if (tid == 0) {
        comb[0] = M;
        comb[1] = L;
      }
    }
    bool bad = false;
    for (int i = PG + tid; i < n_splits * PG; i += NW * 64) {
      u64_t v = 0;
      int spins = 0;
      for (;; ++spins) {
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
      store1<T>(out_, (int64_t)b * out_row_stride + (int64_t)h * D + tid, any_bad ? __uint_as_float(0x7fc00000u) : o1[0]);
    }
    return;
  } else {
  if (tid < D) {
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

// workgroups of `kfn` (block threads, dynamic LDS bytes) the current device holds at once, at most 4 per CU (the regime the spin-wait of the
// in-kernel combine was validated in); 0 when the device cannot be queried.  Cached per (kernel, device).
static int64_t ink_resident_capacity(const void* kfn, int threads, size_t smem) {
  struct Entry { const void* k; int dev; int64_t cap; };
  static std::mutex mu;
  static Entry cache[16];
  static int n_cache = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n_cache; ++i)
    if (cache[i].k == kfn && cache[i].dev == dev) return cache[i].cap;
  int n_cu = 0, per_cu = 0;
  int64_t cap = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, threads, smem > 4096 ? smem : 4096) == hipSuccess)
    cap = (int64_t)n_cu * (per_cu < 4 ? per_cu : 4);
  else
    (void)hipGetLastError();
  if (n_cache < 16) cache[n_cache++] = Entry{kfn, dev, cap};
  return cap;
}

template <typename T, int D, int NW, bool FUSED, int U, int NP = 0>
static void launch_split(const void* q, int64_t q_row_stride, const void* k_slab, const void* v_slab, int64_t stride_b, int64_t stride_h,
                         const int32_t* kv_len, int extra, void* out, int64_t out_row_stride, void* workspace, int n_splits, int B,
                         int n_heads, int n_kv_heads, const void* cos_tab, const void* sin_tab, int n_pos, const int32_t* pos_base,
                         int T_cap, int chunk_keys, hipStream_t st, int call_tag = -1, int64_t part_stride = 0) {
  const float scale = 1.0f / sqrtf((float)D);
  if constexpr (FUSED) {
    // in-kernel combine: only when every workgroup of the grid is certainly resident -- what THIS device (CU count of the current
    // partition mode, occupancy of this kernel with its merge buffer) can hold at once, capped at 4 per CU -- and the caller gave a tag
    // (occupancy is queried ONCE per kernel and device, with the merge buffer of the LARGEST split count the ABI accepts for this path: a later
    // launch with more splits than the first one must not inherit a capacity computed for a smaller LDS footprint -- it bounds a spin-wait)
    if (call_tag >= 0 && n_splits > 1 && n_splits <= 32 &&
        (int64_t)n_splits * n_heads * B <= ink_resident_capacity((const void*)attn_decode_split_kernel<T, D, NW, true, U, true, NP>, NW * 64,
                                                                  (size_t)32 * (D + kAttnPartPad) * sizeof(float))) {
      const size_t smem = (size_t)n_splits * (D + kAttnPartPad) * sizeof(float);
      hipLaunchKernelGGL((attn_decode_split_kernel<T, D, NW, true, U, true, NP>), dim3((unsigned)n_splits, (unsigned)n_heads, (unsigned)B), dim3(NW * 64),
                         smem, st, q, q_row_stride, k_slab, v_slab, stride_b, stride_h, kv_len, extra, reinterpret_cast<float*>(workspace), out,
                         out_row_stride, n_heads / n_kv_heads, scale, cos_tab, sin_tab, n_pos, pos_base, T_cap, n_kv_heads, chunk_keys, call_tag, part_stride);
      return;
    }
  }
  hipLaunchKernelGGL((attn_decode_split_kernel<T, D, NW, FUSED, U, false, NP>), dim3((unsigned)n_splits, (unsigned)n_heads, (unsigned)B), dim3(NW * 64), 0,
                     st, q, q_row_stride, k_slab, v_slab, stride_b, stride_h, kv_len, extra, reinterpret_cast<float*>(workspace), out,
                     out_row_stride, n_heads / n_kv_heads, scale, cos_tab, sin_tab, n_pos, pos_base, T_cap, n_kv_heads, chunk_keys, 0, part_stride);
  if (n_splits > 1)
    hipLaunchKernelGGL((attn_decode_combine_kernel<T, D>), dim3((unsigned)n_heads, (unsigned)B), dim3(D), 0, st,
                       reinterpret_cast<const float*>(workspace), out, out_row_stride, n_splits);
}

}  // namespace dl

using namespace dl;

extern "C" int64_t dl_attn_decode_workspace_bytes(int B, int n_heads, int head_dim, int n_splits) {
  if (n_splits <= 1) return 0;
  // float partials [D + 4] (separate combine launch) or 8-byte granules [D + 2] (in-kernel combine): sized for the larger
  return (int64_t)B * n_heads * n_splits * (head_dim + kAttnPartPad) * (int64_t)sizeof(u64_t);
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

static int attn_decode_rope_impl(const void* qkv, int64_t qkv_row_stride, int n_parts, int64_t part_stride, const void* cos_tab, const void* sin_tab, int n_pos,
                                 const int32_t* pos_base, const int32_t* kv_len, void* k_slab, void* v_slab, int64_t slab_stride_b,
                                 int64_t slab_stride_h, int T_cap, void* out, int64_t out_row_stride, void* workspace, int n_splits,
                                 int keys_in_flight, int chunk_keys, int call_tag, int B, int n_heads, int n_kv_heads, int head_dim, int dtype,
                                 void* stream) {
  DL_REQUIRE(keys_in_flight == 64 || keys_in_flight == 128 || keys_in_flight == 256, "dl_attn_decode_rope: keys_in_flight must be 64, 128 (eight waves) or 256");
  DL_REQUIRE(chunk_keys >= 0, "dl_attn_decode_rope: chunk_keys must be >= 0");
  DL_REQUIRE(qkv && cos_tab && sin_tab && pos_base && kv_len && k_slab && v_slab && out, "dl_attn_decode_rope: NULL pointer");
  DL_REQUIRE(B > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && n_pos > 0 && T_cap > 0, "dl_attn_decode_rope: bad shape");
  DL_REQUIRE(n_splits >= 1 && n_splits <= kMaxSplits, "dl_attn_decode_rope: n_splits=%d must be in [1, %d]", n_splits, kMaxSplits);
  DL_REQUIRE(n_splits == 1 || workspace, "dl_attn_decode_rope: workspace required when n_splits > 1");
  DL_REQUIRE(head_dim == 128 || head_dim == 64, "dl_attn_decode_rope: head_dim=%d unsupported (64 or 128)", head_dim);
  hipStream_t st = as_stream(stream);
  // a 16-byte-per-lane row covers head_dim with D/kVec lanes, so one wave-wide load is 64/(D/kVec) keys: U is chosen so that
  // NW * keys-per-load * U = keys_in_flight for the 16-bit dtypes at head_dim 128 (the production shape)
  // the in-kernel combine is built for the production shape only (U = 4); other variants keep the separate combine launch
  const int tag4 = (keys_in_flight == 64 && chunk_keys == 0) ? call_tag : -1;
#define DL_FUSED_ARGS qkv, qkv_row_stride, k_slab, v_slab, slab_stride_b, slab_stride_h, kv_len, 1, out, out_row_stride, workspace, n_splits, B, n_heads, n_kv_heads, cos_tab, sin_tab, n_pos, pos_base, T_cap, chunk_keys, st
  if (n_parts > 0) {  // fp32 partial sums of the projection's k ranges (16-bit cache types, the four-wave form): the range count is a template argument
#define DL_PARTS_CASE(NP_)                                                                        \
  case NP_:                                                                                       \
    if (dtype == DL_BF16) {                                                                       \
      if (head_dim == 128) launch_split<bf16_t, 128, 4, true, 4, NP_>(DL_FUSED_ARGS, tag4, part_stride); \
      else launch_split<bf16_t, 64, 4, true, 4, NP_>(DL_FUSED_ARGS, tag4, part_stride);           \
    } else {                                                                                      \
      if (head_dim == 128) launch_split<f16_t, 128, 4, true, 4, NP_>(DL_FUSED_ARGS, tag4, part_stride);  \
      else launch_split<f16_t, 64, 4, true, 4, NP_>(DL_FUSED_ARGS, tag4, part_stride);            \
    }                                                                                             \
    break
    switch (n_parts) {
      DL_PARTS_CASE(1);
      DL_PARTS_CASE(2);
      DL_PARTS_CASE(4);
    }
#undef DL_PARTS_CASE
    DL_CHECK_LAUNCH("dl_attn_decode_rope_parts");
    return DL_OK;
  }
  DL_DISPATCH_DTYPE(dtype, T, {
    if (head_dim == 128) {
      if (keys_in_flight == 256) launch_split<T, 128, 4, true, 16>(DL_FUSED_ARGS);
      else if (keys_in_flight == 128) launch_split<T, 128, 8, true, 4>(DL_FUSED_ARGS);
      else launch_split<T, 128, 4, true, 4>(DL_FUSED_ARGS, tag4);
    } else {
      if (keys_in_flight == 256) launch_split<T, 64, 4, true, 16>(DL_FUSED_ARGS);
      else if (keys_in_flight == 128) launch_split<T, 64, 8, true, 4>(DL_FUSED_ARGS);
      else launch_split<T, 64, 4, true, 4>(DL_FUSED_ARGS, tag4);
    }
  });
#undef DL_FUSED_ARGS
  DL_CHECK_LAUNCH("dl_attn_decode_rope");
  return DL_OK;
}

extern "C" int dl_attn_decode_rope(const void* qkv, int64_t qkv_row_stride, const void* cos_tab, const void* sin_tab, int n_pos,
                                   const int32_t* pos_base, const int32_t* kv_len, void* k_slab, void* v_slab, int64_t slab_stride_b,
                                   int64_t slab_stride_h, int T_cap, void* out, int64_t out_row_stride, void* workspace, int n_splits,
                                   int keys_in_flight, int chunk_keys, int call_tag, int B, int n_heads, int n_kv_heads, int head_dim, int dtype,
                                   void* stream) {
  return attn_decode_rope_impl(qkv, qkv_row_stride, 0, 0, cos_tab, sin_tab, n_pos, pos_base, kv_len, k_slab, v_slab, slab_stride_b, slab_stride_h, T_cap, out, out_row_stride,
                               workspace, n_splits, keys_in_flight, chunk_keys, call_tag, B, n_heads, n_kv_heads, head_dim, dtype, stream);
}

extern "C" int dl_attn_decode_rope_parts(const float* qkv_parts, int n_parts, int64_t part_stride, int64_t row_stride, const void* cos_tab, const void* sin_tab, int n_pos,
                                         const int32_t* pos_base, const int32_t* kv_len, void* k_slab, void* v_slab, int64_t slab_stride_b,
                                         int64_t slab_stride_h, int T_cap, void* out, int64_t out_row_stride, void* workspace, int n_splits,
                                         int chunk_keys, int call_tag, int B, int n_heads, int n_kv_heads, int head_dim, int dtype, void* stream) {
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_attn_decode_rope_parts: bf16 / fp16 only (dtype %d): the partial sums are rounded to the cache's 16-bit type", dtype);
  DL_REQUIRE(n_parts == 1 || n_parts == 2 || n_parts == 4, "dl_attn_decode_rope_parts: n_parts=%d must be 1, 2 or 4", n_parts);
  DL_REQUIRE(((uintptr_t)qkv_parts & 15) == 0 && row_stride % 4 == 0 && part_stride % 4 == 0 && (n_parts == 1 || part_stride >= (int64_t)B * row_stride),
             "dl_attn_decode_rope_parts: partial sums must be 16-byte aligned, the ranges at least B rows apart");
  return attn_decode_rope_impl(qkv_parts, row_stride, n_parts, part_stride, cos_tab, sin_tab, n_pos, pos_base, kv_len, k_slab, v_slab, slab_stride_b, slab_stride_h, T_cap, out,
                               out_row_stride, workspace, n_splits, 64, chunk_keys, call_tag, B, n_heads, n_kv_heads, head_dim, dtype, stream);
}
