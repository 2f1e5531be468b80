// dl_linear_tiles: Y[M, N] = X[M, K] Wp^T (+ bias, activation) for the nn.Linear calls of the vision side at M = 577 rows (one image):
// the CLIP ViT-L/14-336 encoder layer's four projections (llava/model/multimodal_encoder/clip_encoder.py:53-71 runs transformers' CLIPEncoderLayer:
// q|k|v [3072, 1024], out_proj [1024, 1024], fc1 [4096, 1024] + QuickGELU, fc2 [1024, 4096]) and the mlp2x_gelu projector
// (llava/model/multimodal_projector/builder.py:172-179).
//
// Why not the library and not dl_linear_packed.  These GEMMs are SMALL (1.2-4.8 GFLOP, 2-8 MB of weights that the 23 layers stream once each): the
// launch is a latency / tiling problem.  hipBLASLt picks 64x160 / 64x64 / 128x64 macro tiles: 260, 160 and 240 workgroups on 256 CUs (fc1: two rounds
// for four workgroups; fc2 / out_proj: 96 CUs idle) and 12.5-21.9 us per GEMM = 0.10 of the MFMA peak (profiles/r05_prefill_mfma_util.txt).
// dl_linear_packed keeps ALL rows in one tile (M <= 256) and streams weights past them; at 577 rows the accumulators do not fit.
// Here the output is cut into (80 rows x 16 NU neurons) tiles chosen so that ONE round covers the chip (fc1: 8 x 32 = 256 workgroups), the k range of a
// long-K GEMM (fc2, out_proj) is split over workgroups as fp32 partial sums for the residual-add / LayerNorm launch that reads them anyway, and bias /
// activation are the epilogue.
//
// One workgroup: 4 LOADER waves + WN CONSUMER waves.
//   loaders   : X -- the operand all consumers share -- by LDS-DMA (one 16-row x 32-k fragment = one KiB per wave instruction; fragment-order X: one contiguous
//               KiB, row-major X: 16 rows x 64 B) into a ring of RD 64-k steps, pieces dealt round-robin to the four loaders; they never touch a register or look
//               at data; loader h waits with a counted vmcnt so that step t + 1 has landed at barrier B(t).  The ring fills while the first steps multiply.
//   consumer c: all TM row tiles x units [c NUW, (c + 1) NUW) of the workgroup.  Its weight fragments (operand-order copy: one contiguous KiB each) go STRAIGHT to
//               registers DW steps ahead -- no other wave multiplies them -- (template WDIR = 0 sends them through the ring too: measurement); per 32-k half step
//               TM ds_read_b128 (lane-linear: conflict-free) feed TM x NUW v_mfma_f32_16x16x32; the next half's fragments are requested before the current
//               half multiplies (order pinned with sched_barrier).
//   one s_barrier per step, as in dl_linear_packed.
// Workgroup b -> XCD b % 8 (observed placement; speed only): the grid is renumbered so that one XCD's workgroups share a k range and a neuron range, i.e.
// W is read from HBM once and X lives in the XCD's L2.
// What bounds the k loop is neither MFMA (340 of 750-1000 cycles per step) nor the path (DMA or registers) but the ~100 outstanding 128-byte requests one CU's
// L1 keeps at L2 at ~470 cycles each: ~26-30 B/clk per CU (profiles/r06_linear_tiles_{timelines,counters}.txt).
// Result: a fixed function of the k order: one fp32 accumulation per output in k order per range (every tile shape returns the same bits), ranges added in
// order by the consumer launch; a row's result does not depend on its position.
#include <mutex>
#include <type_traits>

#include "dl_common.h"

namespace dl {

#define DL_GLOBAL __attribute__((address_space(1)))
#define DL_LDS __attribute__((address_space(3)))

typedef uint32_t lt_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lt_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lt_f16x8 __attribute__((ext_vector_type(8)));
typedef float lt_f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ lt_f32x4 lt_mfma(const lt_u32x4& a, const lt_u32x4& b, lt_f32x4 c);
template <>
__device__ __forceinline__ lt_f32x4 lt_mfma<bf16_t>(const lt_u32x4& a, const lt_u32x4& b, lt_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lt_bf16x8, a), __builtin_bit_cast(lt_bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ lt_f32x4 lt_mfma<f16_t>(const lt_u32x4& a, const lt_u32x4& b, lt_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lt_f16x8, a), __builtin_bit_cast(lt_f16x8, b), c, 0, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void lt_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    lt_static_for<I + 1, N>(f);
  }
}

// one KiB, global -> LDS, no register pass: LDS address = m0 + 16 * lane, global address = s_base + v_off.  Default cache policy: every piece is
// re-read by the other workgroups of the XCD (the M blocks share W, the N blocks share X), so the lines should stay in L2.
__device__ __forceinline__ void lt_dma_piece(const DL_GLOBAL void* s_base, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(v_off), "s"(s_base), "s"(lds_dst)
               : "memory");
}

template <int N>
__device__ __forceinline__ void lt_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

enum { LT_EPI_BIAS = 0, LT_EPI_QGELU = 1, LT_EPI_GELU = 2, LT_EPI_PARTS = 3 };
constexpr int kLtLoaders = 4;

struct LtParams {
  const void* X;
  int64_t ldx;       // row-major X: elements between rows
  int x_packed;      // X is Xp[step][x_tiles][k half][lane][8] (fragment order)
  int x_tiles;       // ceil(M / 16): tiles of the fragment-order X
  const void* Wp;    // dl_pack_weight_tiles output
  const void* bias;  // [N] or NULL
  void* Y;
  int64_t ldy;
  int y_packed;  // Y is written in fragment order with y_tiles = ceil(M / 16) tiles
  int M, n_units, S;  // S = K / 32 slabs
  int n_mb, n_nb, k_split;
  int xcd_remap;
  long long* stamps;  // measurement: [workgroup][8 waves][8] s_memtime stamps (NULL in the product)
  int wrap;           // measurement (stamped entry only): step t reads the operands of step t % wrap -- the second pass over a short K is all L2 hits
};

// 16-bit stores of the epilogue: gfx950 converts fp32 -> bf16 in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even: the same bits as Elem<bf16_t>::from_f for
// every finite value -- tests/test_linear_tiles_gpu.py holds the two against each other); the software rounding costs ~6 VALU per value on the four waves that
// own a workgroup's whole epilogue.
template <typename T>
__device__ __forceinline__ uint32_t lt_pack2(float a, float b) {
  if constexpr (Elem<T>::kBf16) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
  } else {
    return (uint32_t)Elem<T>::from_f(a) | ((uint32_t)Elem<T>::from_f(b) << 16);
  }
}
template <typename T>
__device__ __forceinline__ float lt_round(float a) {
  if constexpr (Elem<T>::kBf16)
    return __uint_as_float(lt_pack2<T>(a, 0.f) << 16);
  else
    return Elem<T>::round(a);
}

// cast(sigmoid(t)) = cast(1 / (1 + expf(-t))) with the bits of the exact expression (dl_quick_gelu) at a quarter of its cost: v_exp_f32 / v_rcp_f32 are within a
// few fp32 ulps of expf / the IEEE quotient, so the ROUNDED value can only differ when the fp32 result sits within `kGuard` fp32 ulps of a rounding boundary of
// the 16-bit type (1e-3 of the values for bf16) or in the range where the fast forms flush -- those lanes take the exact expression.
template <typename T>
__device__ __forceinline__ float lt_sigmoid_rounded(float t) {
  constexpr int kDrop = Elem<T>::kBf16 ? 16 : 13;  // fp32 mantissa bits the 16-bit type drops (fp16 normals)
  constexpr uint32_t kGuard = 64;
  const float fast = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
  const uint32_t low = __float_as_uint(fast) & ((1u << kDrop) - 1u);
  const uint32_t half = 1u << (kDrop - 1);
  const bool near_tie = (low > half ? low - half : half - low) <= kGuard;
  const bool in_range = fabsf(t) < (Elem<T>::kBf16 ? 30.0f : 8.0f);  // fp16: sigmoid below 2^-14 is subnormal in the type (another boundary grid)
  float s = fast;
  if (near_tie || !in_range) s = 1.0f / (1.0f + expf(-t));
  return lt_round<T>(s);
}

// WDIR: the weight fragments do not pass through LDS -- consumer c is the only wave of the workgroup that multiplies units [c NUW, (c + 1) NUW), so it loads their
// fragments (one contiguous KiB each) straight into registers, DW steps ahead, and only X (shared by the WN consumers) travels by LDS-DMA.
template <typename T, int TM, int WN, int NUW, int RD, int EPI, int WDIR, int DW>
__global__ __launch_bounds__(64 * (kLtLoaders + WN)) void linear_tiles_kernel(const LtParams p) {
  using S_ = uint16_t;
  constexpr int NU = WN * NUW;
  constexpr int P = WDIR ? 2 * TM : 2 * (TM + NU);  // DMA pieces per step: X (half 0: TM tiles, half 1: TM tiles)[, then W (half 0: NU units, half 1: NU units)]
  constexpr int kStepBytes = P * 1024;
  // DW (WDIR): steps of weight fragments in registers / in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (k range, neuron block, row block), k range outermost: the 32 workgroups of an XCD (b % 8) then share one k range of X and of a few
  // neuron blocks of W
  const int G = gridDim.x;
  const int v = p.xcd_remap ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int per_ks = p.n_nb * p.n_mb;
  const int ks = v / per_ks, nb = (v % per_ks) / p.n_mb, mb = v % p.n_mb;
  const int all_steps = p.S >> 1;
  const int t_begin = (int)((int64_t)all_steps * ks / p.k_split), t_end = (int)((int64_t)all_steps * (ks + 1) / p.k_split);
  const int steps = t_end - t_begin;
  const int tile0 = mb * TM, u0 = nb * NU;
  DL_LDS unsigned char* ring = (DL_LDS unsigned char*)lt_smem;
#define LT_STAMP(k_)                                                                                              \
  do {                                                                                                            \
    if (p.stamps && lane == 0) p.stamps[((int64_t)blockIdx.x * 8 + w) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  LT_STAMP(0);

  if (w < kLtLoaders) {
    // ---------------- loader h: pieces h, h + 4, h + 8, ... of every step ----------------
    auto run = [&](auto hc) {
      constexpr int H = decltype(hc)::value;
      constexpr int PH = (P - H + kLtLoaders - 1) / kLtLoaders;  // (0 for the last loaders of a one-row-tile workgroup: they only keep the barriers)
      static_assert((RD - 3) * PH <= 63, "ring too deep for the 6-bit vmcnt");
      const DL_GLOBAL char* base[PH > 0 ? PH : 1];
      uint32_t voff[PH > 0 ? PH : 1];
      const int n_real = (p.M + 15) >> 4;
      const uint32_t x_step = p.x_packed ? (uint32_t)p.x_tiles * 2048u : 128u;
      lt_static_for<0, PH>([&](auto qc) {
        constexpr int q = decltype(qc)::value, pc = H + q * kLtLoaders;
        if constexpr (pc < 2 * TM) {
          constexpr int hf = pc / TM, j = pc % TM;
          if (p.x_packed) {
            int tile = tile0 + j;
            tile = tile < n_real ? tile : n_real - 1;  // a tile wholly past row M re-reads the last real one (nothing of it is stored)
            base[q] = (const DL_GLOBAL char*)p.X + ((int64_t)tile * 2 + hf) * 1024;
            voff[q] = (uint32_t)lane * 16u + (uint32_t)t_begin * x_step;
          } else {
            int row = (tile0 + j) * 16 + (lane & 15);
            row = row < p.M ? row : p.M - 1;
            base[q] = (const DL_GLOBAL char*)p.X + hf * 64;
            voff[q] = (uint32_t)row * (uint32_t)p.ldx * 2u + (uint32_t)(lane >> 4) * 16u + (uint32_t)t_begin * x_step;
          }
        } else {
          constexpr int r = pc - 2 * TM, hf = r / NU, i = r % NU;
          const int u = u0 + i < p.n_units ? u0 + i : p.n_units - 1;
          base[q] = (const DL_GLOBAL char*)p.Wp + ((int64_t)u * p.S + hf) * 1024;
          voff[q] = (uint32_t)lane * 16u + (uint32_t)t_begin * 2048u;
        }
      });
      const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
      int slot_issue = 0, issued = 0;
      auto issue = [&]() {
        lt_static_for<0, PH>([&](auto qc) {
          constexpr int q = decltype(qc)::value, pc = H + q * kLtLoaders;
          lt_dma_piece(base[q], voff[q], ring_lds + (uint32_t)(slot_issue * kStepBytes + pc * 1024));
          voff[q] += pc < 2 * TM ? x_step : 2048u;
          if (p.wrap && (issued + 1) % p.wrap == 0) voff[q] -= (uint32_t)p.wrap * (pc < 2 * TM ? x_step : 2048u);
        });
        slot_issue = slot_issue + 1 == RD ? 0 : slot_issue + 1;
        ++issued;
      };
      // the ring fills while the first steps are already being multiplied: two steps before B(0), then two per step until RD - 1 are ahead (a prologue
      // that issues the whole ring first spends (RD - 1) PH issue slots -- over a microsecond -- before anybody may start)
      for (int n = 0; n < 2 && issued < steps; ++n) issue();
      for (int t = 0; t < steps; ++t) {
        // B(t) promises the consumers step t + 1: it has landed once at most the pieces of the steps issued AFTER it are outstanding
        const int ahead = issued - t - 2;
        lt_static_for<0, RD - 2>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if (ahead == k || (k == 0 && ahead < 0)) lt_wait_vmcnt<k * PH>();
        });
        if (t == 0) LT_STAMP(1);
        __syncthreads();  // B(t): steps <= t + 1 are in the ring, step t - 1 is consumed -> its slot is free
        for (int n = 0; n < 2 && issued < steps && issued - t < RD; ++n) issue();
      }
      __syncthreads();  // B(steps)
      LT_STAMP(2);
    };
    if (w == 0)
      run(std::integral_constant<int, 0>{});
    else if (w == 1)
      run(std::integral_constant<int, 1>{});
    else if (w == 2)
      run(std::integral_constant<int, 2>{});
    else
      run(std::integral_constant<int, 3>{});
    return;
  }

  // ---------------- consumer c: TM row tiles x units [c NUW, (c + 1) NUW) ----------------
  const int c = w - kLtLoaders;
  const int lr = lane & 15, lg = lane >> 4;
  lt_f32x4 acc[NUW][TM];
#pragma unroll
  for (int i = 0; i < NUW; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = lt_f32x4{0.f, 0.f, 0.f, 0.f};
  lt_u32x4 x0[TM], x1[TM];
  auto read_x = [&](lt_u32x4(&xd)[TM], int slot, int hf) {
    const DL_LDS unsigned char* s = ring + slot * kStepBytes + lane * 16;
#pragma unroll
    for (int j = 0; j < TM; ++j) xd[j] = *(const DL_LDS lt_u32x4*)(s + (hf * TM + j) * 1024);
  };
  auto mma_half = [&](const lt_u32x4(&xs)[TM], const lt_u32x4(&ws)[NUW]) {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < NUW; ++i) acc[i][j] = lt_mfma<T>(ws[i], xs[j], acc[i][j]);
  };
  if constexpr (WDIR) {
    const lt_u32x4* wsrc[NUW];
#pragma unroll
    for (int i = 0; i < NUW; ++i) {
      const int u = u0 + c * NUW + i < p.n_units ? u0 + c * NUW + i : p.n_units - 1;
      wsrc[i] = reinterpret_cast<const lt_u32x4*>(reinterpret_cast<const char*>(p.Wp) + ((int64_t)u * p.S + 2 * t_begin) * 1024) + lane;
    }
    lt_u32x4 wr[DW][2][NUW];
    auto load_w = [&](auto d, int t) {
#pragma unroll
      for (int i = 0; i < NUW; ++i) {
        const int tw = p.wrap ? t % p.wrap : t;
        wr[d][0][i] = wsrc[i][(int64_t)tw * 128];  // a step = two fragments of 64 x 16 bytes
        wr[d][1][i] = wsrc[i][(int64_t)tw * 128 + 64];
      }
    };
    // always DW - 1 steps in flight, nothing conditional in the steady state (hipcc's s_waitcnt counts are exact only in straight-line code): the last steps
    // re-load the final step
    lt_static_for<0, DW - 1>([&](auto d) { load_w(d, d < steps ? (int)d : (steps > 0 ? steps - 1 : 0)); });
    __syncthreads();  // B(0)
    LT_STAMP(1);
    read_x(x0, 0, 0);
    int slot = 0;
    auto step_body = [&](auto d, int t) {
      constexpr int dn = (decltype(d)::value + DW - 1) % DW;
      // The order below is pinned (sched_barrier): left to itself hipcc sinks every ds_read next to its first use to save registers, which puts an LDS
      // round trip in front of each group of MFMAs -- 1000 cycles per step instead of the 340 the matrix pipe needs (first build; with one consumer
      // wave per SIMD nothing else hides it).
      load_w(std::integral_constant<int, dn>{}, t + DW - 1 < steps ? t + DW - 1 : steps - 1);
      read_x(x1, slot, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(x0, wr[d][0]);
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == RD ? 0 : slot + 1;
      read_x(x0, slot, 0);  // B(t) promised step t + 1
      __builtin_amdgcn_sched_barrier(0);
      mma_half(x1, wr[d][1]);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // B(t + 1)
      if (p.stamps && t + 1 == steps / 2) LT_STAMP(4);
    };
    int t0 = 0;
    for (; t0 + DW <= steps; t0 += DW) lt_static_for<0, DW>([&](auto d) { step_body(d, t0 + decltype(d)::value); });
    lt_static_for<0, DW - 1>([&](auto d) {
      if (t0 + decltype(d)::value < steps) step_body(d, t0 + decltype(d)::value);
    });
  } else {
    lt_u32x4 w0[NUW], w1[NUW];
    auto read_w = [&](lt_u32x4(&wd)[NUW], int slot, int hf) {
      const DL_LDS unsigned char* s = ring + slot * kStepBytes + lane * 16;
#pragma unroll
      for (int i = 0; i < NUW; ++i) wd[i] = *(const DL_LDS lt_u32x4*)(s + (2 * TM + hf * NU + c * NUW + i) * 1024);
    };
    __syncthreads();  // B(0)
    LT_STAMP(1);
    read_w(w0, 0, 0);
    read_x(x0, 0, 0);
    int slot = 0;
    for (int t = 0; t < steps; ++t) {
      read_w(w1, slot, 1);
      read_x(x1, slot, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(x0, w0);
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == RD ? 0 : slot + 1;
      read_w(w0, slot, 0);  // B(t) promised step t + 1 (past the last step: a slot nobody writes any more; the values are not used)
      read_x(x0, slot, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(x1, w1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // B(t + 1)
      if (p.stamps && t + 1 == steps / 2) LT_STAMP(4);
    }
  }
  LT_STAMP(2);

  // ---------------- epilogue: lane (lr, lg) of tile (i, j) holds row 16 (tile0 + j) + lr, neurons 16 (u0 + c NUW + i) + 4 lg + 0..3 ----------------
  if constexpr (EPI == LT_EPI_PARTS) {
    float* Pp = reinterpret_cast<float*>(p.Y) + (int64_t)ks * p.M * p.ldy;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int row = (tile0 + j) * 16 + lr;
      if (row >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NUW; ++i) {
        const int u = u0 + c * NUW + i;
        if (u >= p.n_units) continue;
        // non-temporal: the slices are read once, by another launch on (mostly) other XCDs -- written through as they are produced instead of sitting dirty
        // in this XCD's L2 until the end-of-kernel write-back
        __builtin_nontemporal_store(acc[i][j], reinterpret_cast<lt_f32x4*>(Pp + (int64_t)row * p.ldy + u * 16 + lg * 4));
      }
    }
  } else {
    S_* Y = reinterpret_cast<S_*>(p.Y);
    float bv[NUW][4];
#pragma unroll
    for (int i = 0; i < NUW; ++i) {
      const int u = u0 + c * NUW + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[i][r] = 0.f;
      if (p.bias && u < p.n_units) {
        const uint2 bw = *reinterpret_cast<const uint2*>(reinterpret_cast<const S_*>(p.bias) + u * 16 + lg * 4);
        bv[i][0] = Elem<T>::to_f((S_)(bw.x & 0xffffu));
        bv[i][1] = Elem<T>::to_f((S_)(bw.x >> 16));
        bv[i][2] = Elem<T>::to_f((S_)(bw.y & 0xffffu));
        bv[i][3] = Elem<T>::to_f((S_)(bw.y >> 16));
      }
    }
    const int y_tiles = (p.M + 15) >> 4;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int row = (tile0 + j) * 16 + lr;
      if (row >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NUW; ++i) {
        const int u = u0 + c * NUW + i;
        if (u >= p.n_units) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = acc[i][j][r] + bv[i][r];  // F.linear: fp32 accumulate + bias, one rounding to the dtype
          if constexpr (EPI == LT_EPI_QGELU) {  // dl_quick_gelu's three roundings on the rounded Linear output (HF QuickGELUActivation)
            a = lt_round<T>(a);
            const float t = lt_round<T>(1.702f * a);
            a = a * lt_sigmoid_rounded<T>(t);
          } else if constexpr (EPI == LT_EPI_GELU) {  // nn.GELU() (erf) on the rounded Linear output
            a = gelu_erf(lt_round<T>(a));
          }
          o[r] = a;
        }
        const int col = u * 16 + lg * 4;
        S_* dst = p.y_packed ? Y + lp_x_chunk_offset(row, col >> 3, y_tiles) + (col & 7) : Y + (int64_t)row * p.ldy + col;
        *reinterpret_cast<uint2*>(dst) = make_uint2(lt_pack2<T>(o[0], o[1]), lt_pack2<T>(o[2], o[3]));
      }
    }
  }
  LT_STAMP(3);
#undef LT_STAMP
}

// ---- X in fragment order for this kernel family: Xp[step][tile][k half][lane][8], tile = row / 16 of ceil(M / 16) tiles (rows past M: left unwritten) ----
__global__ __launch_bounds__(256) void lt_pack_x_kernel(const uint16_t* __restrict__ X, int64_t ldx, uint16_t* __restrict__ Xp, int M, int K, int n_tiles) {
  const int steps = K >> 6;
  const int64_t n_chunks = (int64_t)steps * n_tiles * 2 * 64;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_chunks; idx += (int64_t)gridDim.x * 256) {
    const int lane = (int)(idx & 63);
    const int h = (int)((idx >> 6) & 1);
    const int64_t st_ = idx >> 7;
    const int tile = (int)(st_ % n_tiles);
    const int step = (int)(st_ / n_tiles);
    int row = tile * 16 + (lane & 15);
    row = row < M ? row : M - 1;
    const int k = step * 64 + h * 32 + (lane >> 4) * 8;
    *reinterpret_cast<uint4*>(Xp + idx * 8) = *reinterpret_cast<const uint4*>(X + (int64_t)row * ldx + k);
  }
}

struct LtShape {
  int tm, wn, nuw, wdir, dw;
};

template <typename T, int TM, int WN, int NUW, int EPI, int WDIR, int DW>
static int lt_launch(const LtParams& p, hipStream_t st) {
  constexpr int P = WDIR ? 2 * TM : 2 * (TM + WN * NUW);
  constexpr int PH = (P + kLtLoaders - 1) / kLtLoaders;
  constexpr int rd_lds = 156 / P;                  // KiB of LDS / KiB per step
  constexpr int rd_cnt = 63 / PH + 3;              // (RD - 3) PH <= 63
  constexpr int RD = rd_lds < rd_cnt ? (rd_lds < 8 ? rd_lds : 8) : (rd_cnt < 8 ? rd_cnt : 8);
  static_assert(RD >= 4, "tile too large for a ring of four steps");
  auto kfn = linear_tiles_kernel<T, TM, WN, NUW, RD, EPI, WDIR, DW>;
  const size_t smem = (size_t)RD * P * 1024;
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [&] { attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
  if (attr_err != hipSuccess) {
    (void)hipGetLastError();
    set_error("dl_linear_tiles: cannot raise the dynamic LDS limit to %zu bytes", smem);
    return DL_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)(p.n_mb * p.n_nb * p.k_split)), dim3(64 * (kLtLoaders + WN)), smem, st, p);
  return DL_OK;
}

template <typename T, int TM, int WN, int NUW, int WDIR, int DW>
static int lt_epi(const LtParams& p, int epilogue, hipStream_t st) {
  switch (epilogue) {
    case LT_EPI_BIAS: return lt_launch<T, TM, WN, NUW, LT_EPI_BIAS, WDIR, DW>(p, st);
    case LT_EPI_QGELU: return lt_launch<T, TM, WN, NUW, LT_EPI_QGELU, WDIR, DW>(p, st);
    case LT_EPI_GELU: return lt_launch<T, TM, WN, NUW, LT_EPI_GELU, WDIR, DW>(p, st);
    case LT_EPI_PARTS: return lt_launch<T, TM, WN, NUW, LT_EPI_PARTS, WDIR, DW>(p, st);
  }
  set_error("dl_linear_tiles: epilogue %d", epilogue);
  return DL_ERR_ARG;
}

template <typename T>
static int lt_shape(const LtParams& p, const LtShape& s, int epilogue, hipStream_t st) {
#define LT_CASE(tm_, wn_, nuw_, wdir_, dw_) \
  if (s.tm == tm_ && s.wn == wn_ && s.nuw == nuw_ && s.wdir == wdir_ && (!wdir_ || s.dw == dw_)) return lt_epi<T, tm_, wn_, nuw_, wdir_, dw_>(p, epilogue, st)
  LT_CASE(5, 4, 2, 1, 3);
  LT_CASE(5, 4, 2, 1, 5);
  LT_CASE(5, 3, 2, 1, 3);
  LT_CASE(5, 3, 2, 1, 5);
  LT_CASE(5, 2, 2, 1, 3);
  LT_CASE(5, 1, 2, 1, 3);
  LT_CASE(5, 2, 1, 1, 3);
  LT_CASE(5, 2, 1, 1, 5);
  LT_CASE(5, 4, 1, 1, 3);
  LT_CASE(5, 4, 1, 1, 5);
  // two to four images (M = 1154 .. 2308): 160-row tiles -- 18 instead of 2 x 13 KiB-pairs per step and MAC pair, one round of workgroups at two images
  LT_CASE(10, 4, 2, 1, 3);
  LT_CASE(10, 3, 2, 1, 3);
  LT_CASE(10, 4, 1, 1, 3);
#ifdef DL_LT_MEASURE  // measurement builds only (HIPCC_EXTRA=-DDL_LT_MEASURE): the weights through the LDS ring as well (tile_shape + 10000), DESIGN.md section 4d
  LT_CASE(5, 4, 2, 0, 3);
  LT_CASE(5, 3, 2, 0, 3);
  LT_CASE(5, 2, 1, 0, 3);
#endif
#undef LT_CASE
  // partial-sum form only (round 6, late): the decoder's o_proj [H, H] at <= 256 rows -- the post-compaction prefill layers (M = 117..192: 8..12 row tiles as one or two
  // row blocks) and decode batches of 16..32 rows (1..2 row tiles) -- as TM row tiles x 8 units x k ranges, the slices added by dl_add_rmsnorm_parts
#define LT_CASE_PARTS(tm_, dw_)                                                                                      \
  if (s.tm == tm_ && s.wn == 4 && s.nuw == 2 && s.wdir == 1 && s.dw == dw_ && epilogue == LT_EPI_PARTS)            \
  return lt_launch<T, tm_, 4, 2, LT_EPI_PARTS, 1, dw_>(p, st)
  LT_CASE_PARTS(1, 3);
  LT_CASE_PARTS(1, 5);
  LT_CASE_PARTS(2, 3);
  LT_CASE_PARTS(2, 5);
  LT_CASE_PARTS(3, 3);
  LT_CASE_PARTS(4, 3);
  LT_CASE_PARTS(6, 3);
  LT_CASE_PARTS(7, 3);
  LT_CASE_PARTS(8, 3);
  LT_CASE_PARTS(8, 5);
#undef LT_CASE_PARTS
  set_error("dl_linear_tiles: tile shape %d row tiles x %d waves x %d units (weights %s, %d steps ahead) is not built", s.tm, s.wn, s.nuw, s.wdir ? "direct" : "through LDS", s.dw);
  return DL_ERR_ARG;
}

// one round over the chip where the shape allows it: the fewest workgroups <= 256 among the built shapes, the largest tile first
static LtShape lt_pick(int row_tiles, int n_units, int k_split) {
  static const LtShape kShapes[] = {{5, 4, 2, 1, 3}, {5, 3, 2, 1, 3}, {5, 4, 1, 1, 3}, {5, 2, 1, 1, 3}};  // (4 units as 4 waves x 1 beat 2 x 2: tools/bench_linear_tiles.py)
  static const LtShape kShapes10[] = {{10, 4, 2, 1, 3}, {10, 3, 2, 1, 3}, {10, 4, 1, 1, 3}};
  if (row_tiles > 40) {  // more than one image: 160-row tiles halve the bytes per MAC that enter a CU, and two images still fit one round of workgroups
    for (const LtShape& s : kShapes10) {
      const int64_t g = (int64_t)((row_tiles + s.tm - 1) / s.tm) * ((n_units + s.wn * s.nuw - 1) / (s.wn * s.nuw)) * k_split;
      if (g >= 224) return s;
    }
    return kShapes10[2];
  }
  for (const LtShape& s : kShapes) {
    const int64_t g = (int64_t)((row_tiles + s.tm - 1) / s.tm) * ((n_units + s.wn * s.nuw - 1) / (s.wn * s.nuw)) * k_split;
    if (g >= 224) return s;  // at least 7/8 of the chip; smaller tiles only add traffic
  }
  return kShapes[3];
}

}  // namespace dl

extern "C" int64_t dl_tiles_x_bytes(int M, int K) {
  if (M <= 0 || K <= 0 || K % 64) return -1;
  return (int64_t)((M + 15) / 16) * 16 * K * 2;
}

extern "C" int dl_pack_x_rows(const void* X, int64_t ldx, void* Xp, int M, int K, int dtype, void* stream) {
  using namespace dl;
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_pack_x_rows: bf16 / fp16 only (dtype %d)", dtype);
  DL_REQUIRE(M >= 0 && K > 0 && K % 64 == 0 && ldx >= K && ldx % 8 == 0, "dl_pack_x_rows: M=%d, K=%d (multiple of 64), ldx=%lld", M, K, (long long)ldx);
  if (M == 0) return DL_OK;
  DL_REQUIRE(X && Xp && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Xp & 15) == 0, "dl_pack_x_rows: NULL / unaligned pointers");
  const int n_tiles = (M + 15) / 16;
  const int64_t n_chunks = (int64_t)n_tiles * 16 * K / 8;
  const int64_t blocks = (n_chunks + 255) / 256;
  hipLaunchKernelGGL(lt_pack_x_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), (const uint16_t*)X, ldx, (uint16_t*)Xp, M, K, n_tiles);
  DL_CHECK_LAUNCH("dl_pack_x_rows");
  return DL_OK;
}

static int lt_entry(const void* X, int64_t ldx, int x_packed, const void* Wp, const void* bias, void* Y, int64_t ldy, int y_packed, int M, int N, int K,
                    int epilogue, int tile_shape, int k_split, long long* stamps, int dtype, void* stream) {
  using namespace dl;
  const int wrap = stamps ? (epilogue >> 8) : 0;  // measurement (stamped entry only)
  epilogue &= 0xff;
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_linear_tiles: bf16 / fp16 only (dtype %d)", dtype);
  DL_REQUIRE(M >= 0 && N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "dl_linear_tiles: M=%d, N=%d (multiple of 16), K=%d (multiple of 64)", M, N, K);
  if (M == 0) return DL_OK;
  DL_REQUIRE(epilogue >= 0 && epilogue <= LT_EPI_PARTS, "dl_linear_tiles: epilogue %d", epilogue);
  DL_REQUIRE(X && Wp && Y && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Wp & 15) == 0 && ((uintptr_t)Y & 7) == 0 && ldy % 4 == 0,
             "dl_linear_tiles: NULL / unaligned pointers or strides (ldy=%lld)", (long long)ldy);
  DL_REQUIRE(x_packed || (ldx % 8 == 0 && ldx >= K && (int64_t)M * ldx * 2 < ((int64_t)1 << 31)), "dl_linear_tiles: ldx=%lld (multiple of 8, >= K, M ldx < 2^30)", (long long)ldx);
  DL_REQUIRE(!x_packed || (wrap || (int64_t)((M + 15) / 16) * 16 * K * 2 < ((int64_t)1 << 31)), "dl_linear_tiles: packed X past 2 GiB");
  DL_REQUIRE(!y_packed || (epilogue != LT_EPI_PARTS && N % 64 == 0 && ((uintptr_t)Y & 15) == 0), "dl_linear_tiles: a fragment-order output needs N=%d %% 64 == 0 and a 16-bit epilogue", N);
  DL_REQUIRE(y_packed || ldy >= N, "dl_linear_tiles: ldy=%lld < N=%d", (long long)ldy, N);
  DL_REQUIRE(epilogue != LT_EPI_PARTS || (bias == nullptr && ((uintptr_t)Y & 15) == 0), "dl_linear_tiles: partial sums take no bias (the consumer adds it) and a 16-byte aligned fp32 buffer");
  DL_REQUIRE(bias == nullptr || ((uintptr_t)bias & 7) == 0, "dl_linear_tiles: unaligned bias");
  if (k_split <= 0) k_split = 1;
  DL_REQUIRE(k_split <= 8 && K / 64 >= k_split && (k_split == 1 || epilogue == LT_EPI_PARTS), "dl_linear_tiles: k_split=%d (1..8, <= K / 64; > 1 only as partial sums)", k_split);
  LtParams p;
  p.X = X;
  p.ldx = ldx;
  p.x_packed = x_packed;
  p.x_tiles = (M + 15) / 16;
  p.Wp = Wp;
  p.bias = bias;
  p.Y = Y;
  p.ldy = ldy;
  p.y_packed = y_packed;
  p.M = M;
  p.n_units = N / 16;
  p.S = K / 32;
  p.k_split = k_split;
  LtShape s;
  if (tile_shape > 0) {
    s.wdir = tile_shape / 10000 == 1 ? 0 : 1;  // + 10000: the weights through the LDS ring too (measurement)
    s.dw = tile_shape / 10000 == 2 ? 5 : 3;     // + 20000: five steps of weight fragments in flight instead of three
    tile_shape %= 10000;
    s.tm = tile_shape / 100;
    s.wn = (tile_shape / 10) % 10;
    s.nuw = tile_shape % 10;
  } else {
    s = lt_pick(p.x_tiles, p.n_units, k_split);
  }
  p.n_mb = (p.x_tiles + s.tm - 1) / s.tm;
  p.n_nb = (p.n_units + s.wn * s.nuw - 1) / (s.wn * s.nuw);
  p.xcd_remap = ((int64_t)p.n_mb * p.n_nb * k_split) % 8 == 0;
  p.stamps = stamps;
  p.wrap = wrap;
  int rc;
  if (dtype == DL_BF16)
    rc = lt_shape<bf16_t>(p, s, epilogue, as_stream(stream));
  else
    rc = lt_shape<f16_t>(p, s, epilogue, as_stream(stream));
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_linear_tiles");
  return DL_OK;
}

extern "C" int dl_linear_tiles(const void* X, int64_t ldx, int x_packed, const void* Wp, const void* bias, void* Y, int64_t ldy, int y_packed, int M, int N,
                               int K, int epilogue, int tile_shape, int k_split, int dtype, void* stream) {
  return lt_entry(X, ldx, x_packed, Wp, bias, Y, ldy, y_packed, M, N, K, epilogue, tile_shape, k_split, nullptr, dtype, stream);
}

extern "C" int dl_linear_tiles_stamped(const void* X, int64_t ldx, int x_packed, const void* Wp, const void* bias, void* Y, int64_t ldy, int y_packed, int M,
                                       int N, int K, int epilogue, int tile_shape, int k_split, int64_t* stamps, int dtype, void* stream) {
  DL_REQUIRE(stamps, "dl_linear_tiles_stamped: NULL stamp buffer");
  return lt_entry(X, ldx, x_packed, Wp, bias, Y, ldy, y_packed, M, N, K, epilogue, tile_shape, k_split, reinterpret_cast<long long*>(stamps), dtype, stream);
}
