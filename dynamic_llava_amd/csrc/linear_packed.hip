// dl_linear_packed: Y[M <= 256, N] = X[M, K] Wp^T for the decoder projections at a FEW HUNDRED rows or fewer -- DML:1011-1013 (q|k|v), DML:1127 (o_proj),
// DML:328 (gate / up / down) in the post-compaction prefill layers (M = N' = 117..192 packed rows: 30 of the 32 layers of a B=1 request) and in
// decode steps of 25..32 rows -- on a weight copy that dl_pack_weight_tiles wrote ONCE in matrix-core operand order.
//
// At these row counts the GEMM is a weight stream with a matrix-core consumer (7B gate|up at M = 170: 180 MB of weights = 28.6 us at 6.3 TB/s
// against 12.3 us of MFMA at peak).  Rounds 2-4 built four kernels for it on the nn.Linear layout [N, K]; all ended at the library's 2.5-2.9 TB/s
// (profiles/r04_prefill_stream_gemm_experiment.txt).  Two things were wrong with them, and both are structural:
//   * one wave instruction of a [N, K] operand-layout load is 16 rows x 64 bytes, 8 KiB apart.  Here a 16-neuron x 32-k operand fragment is ONE
//     contiguous KiB (`unit` u = neurons [16 u, 16 u + 16), `slab` s = k [32 s, 32 s + 32): Wp[((u S + s) 64 + lane) 8 + j] =
//     W[16 u + lane % 16][32 s + 8 (lane / 16) + j], S = K / 32), a unit's K range one contiguous stream;
//   * a wave that streams weights AND stages X counts both on one in-order vmcnt: the wait for the X tile it must write to LDS also waits for
//     every weight load issued before it, so a "ring of D slabs in flight" is really one or two.  Here the two streams belong to different waves.
//
// One workgroup per CU: 2 LOADER waves + 4 CONSUMER waves.
//   loaders   : global_load_lds_dwordx4 (LDS-DMA, non-temporal) of whole fragments into a ring of RD steps x NU units x 2 KiB (a step = 64 k = two
//               slabs of every unit; loader h moves the h-th slab).  They never touch a register or look at data; their vmcnt counts DMA pieces only,
//               so the ring really is RD - 2 steps of HBM latency deep (tools/experiments/lds_dma_r03.h measured 7.3 TB/s from one such wave per CU).
//   consumers : consumer c owns rows [16 TPW c, 16 TPW (c + 1)) x ALL NU units of the workgroup.  Its X fragments come straight from global memory
//               (L2-resident: X is <= 1.5 MB) in B-operand order -- 16 rows x 64 contiguous bytes per load, the two k halves of a step back to
//               back so every 128-byte line is used whole -- through a register ring DX steps deep; nothing of X passes through LDS, and its
//               vmcnt counts X loads only.  Weight fragments are read from the ring (ds_read_b128, lane-linear: conflict-free), each feeding TPW MFMAs.
//   one s_barrier per step: "step t has landed" (the loaders waited for it) and "step t - 1 is consumed" (its ring slot is re-filled next).
// Every weight byte is read from HBM exactly once.  What bounds the launch is not that stream (the loaders alone run at 5.0-5.4 TB/s) but the bytes every CU pulls
// through its L1: its own weights AND all of X for its k range, ~30 B/clk per CU beside an HBM stream.  Hence k ranges: k_split workgroups
// share a unit set, each a contiguous range of steps -- X per CU shrinks by k_split, the weights do not grow.  The ranges' fp32 tiles meet either inside the launch
// (wave c of a partner -> wave c of the last range, fence-free sc1 stores / loads and a flag; the reducer adds in range order: deterministic, one rounding per
// output) or in the consumer kernel (LP_EPI_PARTS: [k range][M][N] fp32 for dl_add_rmsnorm_parts).
// Epilogues: plain store, SiLU(gate) * up on a gate / up INTERLEAVED packing (unit 2 j = gate tile j, unit 2 j + 1 = up tile j: both values of a neuron sit in
// the same lane), residual add, partial sums; the first two can write Y in the fragment order the next call's X wants.
#include <mutex>
#include <type_traits>

// -DDL_LP_ABLATIONS (measurement builds only: HIPCC_EXTRA=-DDL_LP_ABLATIONS python -m dynamic_llava_amd.build_ext --force) also builds the ablated variants that
// tools/bench_linear_packed.py --ablate times (no MFMA / no X loads / loaders alone); the product library does not contain them.

#include "act_round.h"

namespace dl {

#define DL_GLOBAL __attribute__((address_space(1)))
#define DL_LDS __attribute__((address_space(3)))

typedef uint32_t lp_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lp_f16x8 __attribute__((ext_vector_type(8)));
typedef float lp_f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ lp_f32x4 lp_mfma(const lp_u32x4& a, const lp_u32x4& b, lp_f32x4 c);
template <>
__device__ __forceinline__ lp_f32x4 lp_mfma<bf16_t>(const lp_u32x4& a, const lp_u32x4& b, lp_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lp_bf16x8, a), __builtin_bit_cast(lp_bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ lp_f32x4 lp_mfma<f16_t>(const lp_u32x4& a, const lp_u32x4& b, lp_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lp_f16x8, a), __builtin_bit_cast(lp_f16x8, b), c, 0, 0, 0);
}

// compile-time loop: register rings must be indexed by constants (a loop the optimiser declines to unroll turns them into scratch memory)
template <int I, int N, typename F>
__device__ __forceinline__ void lp_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    lp_static_for<I + 1, N>(f);
  }
}

// one KiB, global -> LDS, no register pass: LDS address = m0 + 16 * lane, global address = s_base + v_off (v_off = 16 * lane + stream offset)
__device__ __forceinline__ void lp_dma_piece(const DL_GLOBAL void* s_base, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(v_off), "s"(s_base), "s"(lds_dst)
               : "memory");
}

template <int N>
__device__ __forceinline__ void lp_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

enum { LP_EPI_STORE = 0, LP_EPI_SILU_PAIR = 1, LP_EPI_RESID = 2, LP_EPI_PARTS = 3 };
constexpr int kLpLoaders = 2, kLpConsumers = 4, kLpThreads = 64 * (kLpLoaders + kLpConsumers);

struct LpParams {
  const void* X;   // [M, K] row-major, ldx elements between rows
  int64_t ldx;
  const void* Wp;  // dl_pack_weight_tiles output
  void* Y;
  int64_t ldy;
  const void* R;   // LP_EPI_RESID: residual rows [M, n_out], may alias Y
  int64_t ldr;
  int M, n_units, S;  // S = K / 32 slabs
  int y_packed;       // Y is written in the same fragment order (the next dl_linear_packed's x_packed input): LP_EPI_STORE / LP_EPI_SILU_PAIR
  int x_packed;       // X is Xp[step][tile][k half][lane][8] (dl_pack_x_tiles / a producer's packed output): every fragment one contiguous KiB
  int n_sets, k_split;  // workgroup b: unit set b % n_sets (NU units), k range b / n_sets of k_split
  int trim256;          // see the k-range bounds in the kernel (0 for LP_EPI_PARTS / one range)
  int* flags;           // [n_sets][k_split - 1][4 consumers]: 1 once that wave's partial tiles are in `parts`; zero before and after every launch
  float* parts;         // [n_sets][k_split - 1][4 consumers][NU x TPW tiles][64 lanes x 4]
  int32_t* err;         // may be NULL: bit 3 is set if a reducing wave gave up waiting for a partner
  long long* stamps;    // measurement: [workgroup][6 waves][8] s_memtime stamps (NULL in the product)
};

// ABL (measurement only): bit 0 = no MFMA, bit 1 = no X loads, bit 2 = no LDS fragment reads
template <typename T, int NU, int TPW, int RD, int DX, int EPI, int ABL>
__global__ __launch_bounds__(kLpThreads) void linear_packed_kernel(const LpParams p) {
  using S_ = uint16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char lp_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // k_split workgroups share a unit set, each a contiguous range of 64-wide k steps; the one with the LAST range (and the highest block index: it is
  // dispatched after the partners it waits for, whatever the grid size) adds the partners' fp32 tiles to its own, in range order, and runs the epilogue
  const int set = blockIdx.x % p.n_sets, ks = blockIdx.x / p.n_sets;
  const int all_steps = p.S >> 1;  // 64 k per step
  // In-kernel hand-over: the partners' ranges are `trim256` / 256 shorter than an even share and the reducer's range longer, so that a partner's
  // tiles are written through and acknowledged (5-6 us) while the reducer still multiplies -- the reducer then pays only for reading them.
  auto bound = [&](int r) { return r >= p.k_split ? all_steps : (int)((int64_t)all_steps * r * (256 - p.trim256) / (256 * p.k_split)); };
  const int t_begin = bound(ks), t_end = bound(ks + 1);
  const int steps = t_end - t_begin;
  const int u0 = set * NU;
  DL_LDS unsigned char* ring = (DL_LDS unsigned char*)lp_smem;
  constexpr int kStepBytes = NU * 2048;
#define LP_STAMP(k_)                                                                                               \
  do {                                                                                                             \
    if (p.stamps && lane == 0) p.stamps[((int64_t)blockIdx.x * 10 + w) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  LP_STAMP(0);

  if (w < kLpLoaders) {
    // ---------------- loader h: slab 2 t + h of every unit, step after step ----------------
    const int h = w;
    const DL_GLOBAL char* base[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = u0 + i < p.n_units ? u0 + i : p.n_units - 1;  // (a workgroup past the last unit re-streams it; nothing is stored)
      base[i] = (const DL_GLOBAL char*)p.Wp + ((int64_t)u * p.S + h) * 1024;
    }
    const uint32_t ring_lds = (uint32_t)(uintptr_t)ring + (uint32_t)h * 1024u;
    uint32_t voff = (uint32_t)lane * 16u + (uint32_t)t_begin * 2048u;  // + 2048 per step
    auto issue = [&](int slot) {
#pragma unroll
      for (int i = 0; i < NU; ++i) lp_dma_piece(base[i], voff, ring_lds + (uint32_t)(slot * kStepBytes + i * 2048));
      voff += 2048u;
    };
    int slot_issue = 0;
    int issued = 0;
    for (; issued < RD - 1 && issued < steps; ++issued) {
      issue(slot_issue);
      slot_issue = slot_issue + 1 == RD ? 0 : slot_issue + 1;
    }
    for (int t = 0; t < steps; ++t) {
      // B(t) promises the consumers step t + 1 (they read its first half before B(t + 1)): it has landed once at most the pieces of the steps
      // issued AFTER it are outstanding
      if (issued - 2 - t >= RD - 3)
        lp_wait_vmcnt<(RD - 3) * NU>();
      else
        lp_wait_vmcnt<0>();
      if (t == 0) LP_STAMP(1);
      __syncthreads();  // B(t): steps <= t + 1 are in the ring, step t - 1 is consumed -> its slot is free
      if (issued < steps) {
        issue(slot_issue);
        slot_issue = slot_issue + 1 == RD ? 0 : slot_issue + 1;
        ++issued;
      }
    }
    __syncthreads();  // B(steps): the consumers' last step ends with a barrier like every other
    LP_STAMP(2);
    return;
  }

  // ---------------- consumer c: rows [16 TPW c, 16 TPW (c + 1)) x NU units ----------------
  const int c = w - kLpLoaders;
  const int lr = lane & 15, lg = lane >> 4;
  const S_* X = reinterpret_cast<const S_*>(p.X);
  const S_* xp[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int row = (c * TPW + j) * 16 + lr;
    row = row < p.M ? row : p.M - 1;  // rows past M compute values that are never stored
    xp[j] = X + (int64_t)row * p.ldx + lg * 8;
  }
  const bool x_packed = p.x_packed != 0;
  if (x_packed) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      // a tile wholly past row M (M = 170: the twelfth) re-reads the last real tile instead of its own padding: the same lines the wave asked for a
      // moment ago, so nothing more leaves L2 for rows nobody stores
      const int n_real = (p.M + 15) >> 4, tile = c * TPW + j;
      xp[j] = X + ((int64_t)(tile < n_real ? tile : n_real - 1) * 2) * 512 + lane * 8;
    }
  }
  const int64_t x_step = x_packed ? (int64_t)kLpConsumers * TPW * 1024 : 64;
  const int x_half = x_packed ? 512 : 32;
  lp_u32x4 xr[DX][TPW][2];
  auto load_x = [&](auto d, int t_) {
    if constexpr (!(ABL & 2)) {
      const int t = t_ + t_begin;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        xr[d][j][0] = *reinterpret_cast<const lp_u32x4*>(xp[j] + (int64_t)t * x_step);
        xr[d][j][1] = *reinterpret_cast<const lp_u32x4*>(xp[j] + (int64_t)t * x_step + x_half);
      }
    }
  };
  if constexpr (ABL & 2) {
#pragma unroll
    for (int d = 0; d < DX; ++d)
#pragma unroll
      for (int j = 0; j < TPW; ++j) xr[d][j][0] = xr[d][j][1] = lp_u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  }
  // always DX - 1 steps in flight: the waits count loads.  (A trimmed partner range of a very short K can be EMPTY -- K = 128 in two ranges: 0 + 2 steps --
  // and must not ask for step -1: it re-loads step t_begin, which exists because the last range never is empty, and hands over zeros.)
  lp_static_for<0, DX - 1>([&](auto d) { load_x(d, d < steps ? (int)d : (steps > 0 ? steps - 1 : 0)); });

  lp_f32x4 acc[NU][TPW];
#pragma unroll
  for (int i = 0; i < NU; ++i)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = lp_f32x4{0.f, 0.f, 0.f, 0.f};

  // weight fragments: half a step (NU fragments) in registers while the other half multiplies, so an LDS read has NU x TPW MFMAs to land behind
  lp_u32x4 w0[NU], w1[NU];
  auto read_half = [&](lp_u32x4(&dst)[NU], int slot, int h) {
    const DL_LDS unsigned char* fr = ring + slot * kStepBytes + h * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if constexpr (ABL & 4)
        dst[i] = lp_u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      else
        dst[i] = *(const DL_LDS lp_u32x4*)(fr + i * 2048);
    }
  };
  auto mma_half = [&](const lp_u32x4(&wf)[NU], auto d, auto h) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if constexpr (!(ABL & 1)) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[i][j] = lp_mfma<T>(wf[i], xr[d][j][h], acc[i][j]);
      } else {
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[i][j].x += __uint_as_float(wf[i].x ^ xr[d][j][h].x);
      }
    }
  };
  // Nothing in a steady-state step is conditional: hipcc's s_waitcnt insertion takes the SMALLEST possible count of younger loads at a control-flow
  // join, so a prefetch behind `if (t + DX - 1 < steps)` makes every wait for step t's fragments also wait for the loads just issued for step
  // t + DX - 1 (first build: vmcnt(5) where 17 was meant -- one L2 round trip exposed per step).  The last steps re-load the final step's X and
  // re-read a ring slot that nobody writes any more; both are harmless.
  __syncthreads();  // B(0)
  LP_STAMP(1);
  read_half(w0, 0, 0);
  int slot = 0;
  auto step_body = [&](auto d, int t) {
    constexpr int dn = (d + DX - 1) % DX;
    load_x(std::integral_constant<int, dn>{}, t + DX - 1 < steps ? t + DX - 1 : steps - 1);
    read_half(w1, slot, 1);
    mma_half(w0, d, std::integral_constant<int, 0>{});
    slot = slot + 1 == RD ? 0 : slot + 1;
    read_half(w0, slot, 0);  // B(t) promised step t + 1
    mma_half(w1, d, std::integral_constant<int, 1>{});
    __syncthreads();  // B(t + 1)
  };
  int t0 = 0;
  for (; t0 + DX <= steps; t0 += DX) lp_static_for<0, DX>([&](auto d) { step_body(d, t0 + d); });
  lp_static_for<0, DX - 1>([&](auto d) {
    if (t0 + d < steps) step_body(d, t0 + d);
  });

  LP_STAMP(2);
  // ---------------- k_split > 1: hand the fp32 tiles over / take the partners' ----------------
  // Wave c of a partner and wave c of the reducer own the same tiles, so the hand-over is wave to wave: no workgroup-wide step on either side.
  // No fences: an agent-scope release / acquire pair here is buffer_wbl2 / buffer_inv sc1 -- a write-back and an invalidate of the XCD's WHOLE L2,
  // issued by 4 waves of 256 workgroups while everybody streams X out of that L2 (first build: qkv 39 -> 46 us with two k ranges instead of
  // 39 -> 25).  The tiles travel as 8-byte agent-scope stores (global_store_dwordx2 ... sc1: written through, past the non-coherent L2s),
  // the wave waits for their acknowledgements, then publishes its flag the same way; the reducer polls the flag and reads the tiles with agent-scope
  // loads (sc1: never a stale line of its own L2, which still holds the previous launch's tiles at these addresses).
  if constexpr (EPI == LP_EPI_PARTS) {
    // fp32 partial sums of this workgroup's k range, [k range][M][N], for a consumer that adds the ranges in order (dl_add_rmsnorm_parts: o_proj /
    // down_proj -> residual add + RMSNorm): no hand-over, no flags, plain 16-byte stores
    float* P = reinterpret_cast<float*>(p.Y) + (int64_t)ks * p.M * p.ldy;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int row = (c * TPW + j) * 16 + lr;
      if (row >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (u0 + i >= p.n_units) continue;
        *reinterpret_cast<float4*>(P + (int64_t)row * p.ldy + (u0 + i) * 16 + lg * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    LP_STAMP(4);
    return;
  }
  if (p.k_split > 1) {
    constexpr int kTiles = NU * TPW;
    typedef unsigned long long u64;
    typedef __attribute__((address_space(1))) u64 gu64;
    typedef __attribute__((address_space(1))) int gi32;
    if (ks + 1 < p.k_split) {
      const int slot_ = (set * (p.k_split - 1) + ks) * kLpConsumers + c;
      u64* dst = reinterpret_cast<u64*>(p.parts) + ((int64_t)slot_ * kTiles * 64 + lane) * 2;
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          u64* d = dst + (i * TPW + j) * 128;
          __hip_atomic_store((gu64*)d, (u64)__float_as_uint(acc[i][j][0]) | ((u64)__float_as_uint(acc[i][j][1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store((gu64*)(d + 1), (u64)__float_as_uint(acc[i][j][2]) | ((u64)__float_as_uint(acc[i][j][3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every tile store of this wave is acknowledged
      if (lane == 0) __hip_atomic_store((gi32*)(p.flags + slot_), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      LP_STAMP(3);
      return;
    }
    for (int q = 0; q + 1 < p.k_split; ++q) {
      const int slot_ = (set * (p.k_split - 1) + q) * kLpConsumers + c;
      int spins = 0;
      while (__hip_atomic_load((const gi32*)(p.flags + slot_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {  // seconds: the partner never ran (never on a healthy launch: it was dispatched before this workgroup)
          if (p.err && lane == 0) atomicOr(p.err, 8);
          break;
        }
      }
      const u64* src = reinterpret_cast<const u64*>(p.parts) + ((int64_t)slot_ * kTiles * 64 + lane) * 2;
#pragma unroll
      for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const u64* sp = src + (i * TPW + j) * 128;
          const u64 a0 = __hip_atomic_load((const gu64*)sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const u64 a1 = __hip_atomic_load((const gu64*)(sp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][0] += __uint_as_float((uint32_t)a0);
          acc[i][j][1] += __uint_as_float((uint32_t)(a0 >> 32));
          acc[i][j][2] += __uint_as_float((uint32_t)a1);
          acc[i][j][3] += __uint_as_float((uint32_t)(a1 >> 32));
        }
      if (lane == 0) __hip_atomic_store((gi32*)(p.flags + slot_), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
    }
  }

  LP_STAMP(3);
  // ---------------- epilogue: lane (lr, lg) of tile (i, j) holds row 16 (c TPW + j) + lr, neurons 16 (u0 + i) + 4 lg + 0..3 ----------------
  S_* Y = reinterpret_cast<S_*>(p.Y);
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int row = (c * TPW + j) * 16 + lr;
    if (row >= p.M) continue;
    if constexpr (EPI == LP_EPI_SILU_PAIR) {
      static_assert(EPI != LP_EPI_SILU_PAIR || NU % 2 == 0, "gate / up tiles come in pairs");
#pragma unroll
      for (int i = 0; i < NU; i += 2) {
        if (u0 + i >= p.n_units) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = hw_round<T>(acc[i][j][r]), u = hw_round<T>(acc[i + 1][j][r]);
          const float sg = silu_rounded<T>(g);  // dl_silu_mul's two roundings (DML:328), the bits of its exact expression (act_round.h)
          o[r] = sg * u;
        }
        const uint32_t lo = hw_pack2<T>(o[0], o[1]);
        const uint32_t hi = hw_pack2<T>(o[2], o[3]);
        const int col = ((u0 + i) >> 1) * 16 + lg * 4;
        S_* dst = p.y_packed ? Y + lp_x_chunk_offset(row, col >> 3, kLpConsumers * TPW) + (col & 7) : Y + (int64_t)row * p.ldy + col;
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (u0 + i >= p.n_units) continue;
        const int col = (u0 + i) * 16 + lg * 4;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[i][j][r];
        if constexpr (EPI == LP_EPI_RESID) {
          const uint2 rv = *reinterpret_cast<const uint2*>(reinterpret_cast<const S_*>(p.R) + (int64_t)row * p.ldr + col);
          const uint32_t rw[2] = {rv.x, rv.y};
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = Elem<T>::to_f((S_)((rw[r >> 1] >> ((r & 1) * 16)) & 0xffffu)) + Elem<T>::round(o[r]);  // the projection is rounded first (DML:1289/1295)
        }
        const uint32_t lo = (uint32_t)Elem<T>::from_f(o[0]) | ((uint32_t)Elem<T>::from_f(o[1]) << 16);
        const uint32_t hi = (uint32_t)Elem<T>::from_f(o[2]) | ((uint32_t)Elem<T>::from_f(o[3]) << 16);
        S_* dst = (EPI == LP_EPI_STORE && p.y_packed) ? Y + lp_x_chunk_offset(row, col >> 3, kLpConsumers * TPW) + (col & 7) : Y + (int64_t)row * p.ldy + col;
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LP_STAMP(4);
#undef LP_STAMP
}

// ---- packing: one thread per 16-byte operand chunk ----
template <int PAIR>
__global__ __launch_bounds__(256) void pack_weight_tiles_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wp, int N, int K) {
  const int S = K >> 5;
  const int64_t n_chunks = (int64_t)(N >> 4) * S * 64;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n_chunks; idx += (int64_t)gridDim.x * 256) {
    const int lane = (int)(idx & 63);
    const int64_t us = idx >> 6;
    const int s = (int)(us % S);
    const int u = (int)(us / S);
    int tile = u;
    if (PAIR) tile = (u & 1) * (N >> 5) + (u >> 1);  // unit 2 j = gate tile j, unit 2 j + 1 = up tile j (rows [N / 2, N) are `up`)
    const int n = tile * 16 + (lane & 15);
    const int k = s * 32 + (lane >> 4) * 8;
    *reinterpret_cast<uint4*>(Wp + idx * 8) = *reinterpret_cast<const uint4*>(W + (int64_t)n * K + k);
  }
}

template <typename T, int NU, int TPW, int EPI, int ABL>
static int lp_launch(const LpParams& p, int rd, hipStream_t st) {
  // ring depth: (RD - 3) NU pieces per loader must fit the 6-bit vmcnt, RD NU 2 KiB the LDS; 50-60 KiB in flight per CU cover HBM's latency at 25 GB/s per CU
  constexpr int RD = NU == 1 ? 24 : NU == 2 ? 16 : NU == 3 ? 12 : NU == 4 ? 10 : NU == 12 ? 6 : 8;
  constexpr int DX = (TPW >= 4 || NU >= 12) ? 2 : 3;
  static_assert((RD - 3) * NU <= 63 && RD * NU * 2 <= 152, "ring too deep for vmcnt / LDS");
  (void)rd;
  auto kfn = linear_packed_kernel<T, NU, TPW, RD, DX, EPI, ABL>;
  const size_t smem = (size_t)RD * NU * 2048;
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [&] { attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
  if (attr_err != hipSuccess) {
    (void)hipGetLastError();
    set_error("dl_linear_packed: cannot raise the dynamic LDS limit to %zu bytes", smem);
    return DL_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)(p.n_sets * p.k_split)), dim3(kLpThreads), smem, st, p);
  return DL_OK;
}

template <typename T, int NU, int TPW, int ABL>
static int lp_epi(const LpParams& p, int epilogue, hipStream_t st) {
  if (epilogue == LP_EPI_STORE) return lp_launch<T, NU, TPW, LP_EPI_STORE, ABL>(p, 0, st);
  if constexpr (NU % 2 == 0 && ABL == 0) {
    if (epilogue == LP_EPI_SILU_PAIR) return lp_launch<T, NU, TPW, LP_EPI_SILU_PAIR, 0>(p, 0, st);
  }
  if constexpr (ABL == 0) {
    if (epilogue == LP_EPI_RESID) return lp_launch<T, NU, TPW, LP_EPI_RESID, 0>(p, 0, st);
    if (epilogue == LP_EPI_PARTS) return lp_launch<T, NU, TPW, LP_EPI_PARTS, 0>(p, 0, st);
  }
  set_error("dl_linear_packed: epilogue %d is not built for %d units per workgroup", epilogue, NU);
  return DL_ERR_ARG;
}

template <typename T, int NU, int ABL>
static int lp_tpw(const LpParams& p, int epilogue, hipStream_t st) {
  const int tiles = (p.M + 15) / 16;
  const int tpw = (tiles + kLpConsumers - 1) / kLpConsumers;
  if constexpr (ABL != 0) {  // measurement builds: the prefill shape only
    if (tpw == 3) return lp_epi<T, NU, 3, ABL>(p, epilogue, st);
    set_error("dl_linear_packed: ablations are built for 129..192 rows");
    return DL_ERR_ARG;
  } else {
    switch (tpw) {
      case 1: return lp_epi<T, NU, 1, ABL>(p, epilogue, st);
      case 2: return lp_epi<T, NU, 2, ABL>(p, epilogue, st);
      case 3: return lp_epi<T, NU, 3, ABL>(p, epilogue, st);
      case 4: return lp_epi<T, NU, 4, ABL>(p, epilogue, st);
    }
    set_error("dl_linear_packed: M=%d is past the 256 rows this kernel holds in one tile", p.M);
    return DL_ERR_ARG;
  }
}

template <typename T, int ABL>
static int lp_nu(const LpParams& p, int nu, int epilogue, hipStream_t st) {
  if constexpr (ABL != 0) {
    switch (nu) {
      case 3: return lp_tpw<T, 3, ABL>(p, epilogue, st);
      case 6: return lp_tpw<T, 6, ABL>(p, epilogue, st);
    }
  } else {
    switch (nu) {
      case 1: return lp_tpw<T, 1, ABL>(p, epilogue, st);
      case 2: return lp_tpw<T, 2, ABL>(p, epilogue, st);
      case 3: return lp_tpw<T, 3, ABL>(p, epilogue, st);
      case 4: return lp_tpw<T, 4, ABL>(p, epilogue, st);
      case 6: return lp_tpw<T, 6, ABL>(p, epilogue, st);
      case 8: return lp_tpw<T, 8, ABL>(p, epilogue, st);
#ifdef DL_LP_MEASURE_12U  // round 6 experiment (verdict r5 item 5; HIPCC_EXTRA=-DDL_LP_MEASURE_12U, tools/bench_gate_up_12units.py): gate|up at 129..192 rows as 12 units
      case 12:              // x 2 k ranges of partial sums for dl_silu_mul_parts -- 58.4 us + 8.7 against the shipped launch's 55.3 (profiles/r06_gate_up_12units.txt)
        if (epilogue == LP_EPI_PARTS && (p.M + 15) / 16 > 2 * kLpConsumers && (p.M + 15) / 16 <= 3 * kLpConsumers) return lp_launch<T, 12, 3, LP_EPI_PARTS, 0>(p, 0, st);
        set_error("dl_linear_packed: 12 units per workgroup are built for partial sums at 129..192 rows only");
        return DL_ERR_ARG;
#endif
    }
  }
  set_error("dl_linear_packed: units_per_workgroup=%d is not built", nu);
  return DL_ERR_ARG;
}

}  // namespace dl

extern "C" int64_t dl_packed_weight_bytes(int N, int K, int dtype) {
  if (N <= 0 || K <= 0 || N % 16 || K % 64 || dtype == DL_F32) return -1;
  return (int64_t)N * K * 2;
}

extern "C" int dl_pack_weight_tiles(const void* W, void* Wp, int N, int K, int gate_up_pairs, int dtype, void* stream) {
  using namespace dl;
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_pack_weight_tiles: bf16 / fp16 only (dtype %d)", dtype);
  DL_REQUIRE(N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "dl_pack_weight_tiles: N=%d must be a multiple of 16, K=%d of 64", N, K);
  DL_REQUIRE(!gate_up_pairs || N % 32 == 0, "dl_pack_weight_tiles: gate|up pairs need N=%d to be a multiple of 32", N);
  DL_REQUIRE(W && Wp && W != Wp && ((uintptr_t)W & 15) == 0 && ((uintptr_t)Wp & 15) == 0, "dl_pack_weight_tiles: NULL / unaligned / aliased pointers");
  const int64_t n_chunks = (int64_t)N * K / 8;
  const int64_t blocks = (n_chunks + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
  if (gate_up_pairs)
    hipLaunchKernelGGL((pack_weight_tiles_kernel<1>), dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)W, (uint16_t*)Wp, N, K);
  else
    hipLaunchKernelGGL((pack_weight_tiles_kernel<0>), dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)W, (uint16_t*)Wp, N, K);
  DL_CHECK_LAUNCH("dl_pack_weight_tiles");
  return DL_OK;
}

namespace dl {
// ---- X in fragment order: Xp[step][tile][k half][lane][8] with tile = row / 16, lane = 16 * ((k % 32) / 8) + row % 16; rows past M repeat row M - 1 ----
template <int DUMMY>
__global__ __launch_bounds__(256) void pack_x_tiles_kernel(const uint16_t* __restrict__ X, int64_t ldx, uint16_t* __restrict__ Xp, int M, int K, int n_tiles) {
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

// The flag words sit in a FIXED region at the start of the workspace, whatever the call's (units per workgroup, k_split): one workspace shared by calls of
// different shapes (model._lp_ws) then never has one call's fp32 tiles on top of another call's flag words, and "the flag words are zero before and after
// every launch" holds literally.  256 unit sets x 7 partner ranges x 4 consumer waves x 4 bytes = 28 KiB at most.
constexpr int64_t kLpFlagBytes = 64 * 1024;

static int lp_tiles_per_wave(int M) { return ((M + 15) / 16 + dl::kLpConsumers - 1) / dl::kLpConsumers; }

}  // namespace dl

extern "C" int64_t dl_packed_x_bytes(int M, int K) {
  if (M <= 0 || K <= 0 || K % 64 || M > 256) return -1;
  return (int64_t)dl::lp_tiles_per_wave(M) * dl::kLpConsumers * 16 * K * 2;
}

extern "C" int dl_pack_x_tiles(const void* X, int64_t ldx, void* Xp, int M, int K, int dtype, void* stream) {
  using namespace dl;
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_pack_x_tiles: bf16 / fp16 only (dtype %d)", dtype);
  DL_REQUIRE(M >= 0 && M <= 256 && K > 0 && K % 64 == 0 && ldx >= K && ldx % 8 == 0, "dl_pack_x_tiles: M=%d (<= 256), K=%d (multiple of 64), ldx=%lld", M, K, (long long)ldx);
  if (M == 0) return DL_OK;
  DL_REQUIRE(X && Xp && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Xp & 15) == 0, "dl_pack_x_tiles: NULL / unaligned pointers");
  const int n_tiles = lp_tiles_per_wave(M) * kLpConsumers;
  const int64_t n_chunks = (int64_t)n_tiles * 16 * K / 8;
  const int64_t blocks = (n_chunks + 255) / 256;
  hipLaunchKernelGGL((pack_x_tiles_kernel<0>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), (const uint16_t*)X, ldx, (uint16_t*)Xp, M, K, n_tiles);
  DL_CHECK_LAUNCH("dl_pack_x_tiles");
  return DL_OK;
}

static int lp_pick_units(int n_units, int epilogue, int k_split) {
  static const int kChoices[] = {1, 2, 3, 4, 6, 8};
  int nu = 8;
  for (int c : kChoices) {
    if (epilogue == dl::LP_EPI_SILU_PAIR && (c & 1)) continue;
    if ((int64_t)((n_units + c - 1) / c) * k_split <= 256) {  // one workgroup per CU, the fewest units per workgroup that fits
      nu = c;
      break;
    }
  }
  return nu;
}

extern "C" int64_t dl_linear_packed_workspace_bytes(int M, int N, int K, int epilogue, int units_per_workgroup, int k_split) {
  if (M <= 0 || M > 256 || N <= 0 || N % 16 || K <= 0 || K % 64 || k_split < 1 || k_split > 8) return -1;
  if (k_split == 1) return 0;
  const int nu = units_per_workgroup > 0 ? units_per_workgroup : lp_pick_units(N / 16, epilogue, k_split);
  const int64_t n_sets = (N / 16 + nu - 1) / nu;
  const int64_t n_slots = n_sets * (k_split - 1) * dl::kLpConsumers;
  return dl::kLpFlagBytes + n_slots * nu * dl::lp_tiles_per_wave(M) * 1024;
}

static int lp_entry(const void* X, int64_t ldx, int x_packed, const void* Wp, void* Y, int64_t ldy, const void* resid, int64_t ldr, int M, int N, int K, int epilogue,
                    int units_per_workgroup, int k_split, void* workspace, int32_t* err_flag, long long* stamps, int dtype, void* stream) {
  using namespace dl;
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_linear_packed: bf16 / fp16 only (dtype %d)", dtype);
  DL_REQUIRE(M >= 0 && N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "dl_linear_packed: M=%d, N=%d (multiple of 16), K=%d (multiple of 64)", M, N, K);
  if (M == 0) return DL_OK;
  const int abl = (epilogue >> 8) & 0xff;  // measurement builds only (tools/bench_linear_packed.py)
  const int y_packed = (epilogue >> 4) & 1;  // DL_LP_Y_PACKED
  const int trim_arg = (epilogue >> 16) & 0xff;  // measurement: trim256 + 1 (0 = the default below)
  epilogue &= 0xf;
  DL_REQUIRE(epilogue >= 0 && epilogue <= LP_EPI_PARTS, "dl_linear_packed: epilogue %d", epilogue);
  DL_REQUIRE(!y_packed || epilogue == LP_EPI_STORE || epilogue == LP_EPI_SILU_PAIR, "dl_linear_packed: a fragment-order output goes with epilogue 0 or 1");
  DL_REQUIRE(X && Wp && Y && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Wp & 15) == 0 && ((uintptr_t)Y & 7) == 0 && ldy % 4 == 0,
             "dl_linear_packed: NULL / unaligned pointers or strides (ldy=%lld)", (long long)ldy);
  DL_REQUIRE(x_packed || (ldx % 8 == 0 && ldx >= K), "dl_linear_packed: ldx=%lld (multiple of 8, >= K)", (long long)ldx);
  const int n_out = epilogue == LP_EPI_SILU_PAIR ? N / 2 : N;
  DL_REQUIRE(y_packed || ldy >= n_out, "dl_linear_packed: ldy=%lld < %d output columns", (long long)ldy, n_out);
  DL_REQUIRE(!y_packed || (n_out % 64 == 0 && ((uintptr_t)Y & 15) == 0), "dl_linear_packed: a fragment-order output needs %d output columns to be a multiple of 64", n_out);
  DL_REQUIRE(epilogue != LP_EPI_PARTS || (((uintptr_t)Y & 15) == 0 && ldy % 4 == 0), "dl_linear_packed: partial sums need a 16-byte aligned fp32 buffer");
  DL_REQUIRE(epilogue != LP_EPI_SILU_PAIR || N % 32 == 0, "dl_linear_packed: gate|up pairs need N=%d to be a multiple of 32", N);
  DL_REQUIRE(epilogue != LP_EPI_RESID || (resid && ((uintptr_t)resid & 7) == 0 && ldr % 4 == 0 && ldr >= N), "dl_linear_packed: residual pointer / stride");
  if (k_split <= 0) k_split = 1;
  DL_REQUIRE(k_split <= 8 && K / 64 >= k_split, "dl_linear_packed: k_split=%d (1..8, <= K / 64)", k_split);
  DL_REQUIRE(k_split == 1 || epilogue == LP_EPI_PARTS || (workspace && ((uintptr_t)workspace & 255) == 0),
             "dl_linear_packed: k_split > 1 needs a 256-byte aligned workspace (dl_linear_packed_workspace_bytes, zeroed once)");
  LpParams p;
  p.X = X;
  p.ldx = ldx;
  p.Wp = Wp;
  p.Y = Y;
  p.ldy = ldy;
  p.R = resid;
  p.ldr = ldr;
  p.M = M;
  p.n_units = N / 16;
  p.S = K / 32;
  p.x_packed = x_packed;
  p.y_packed = y_packed;
  const int nu = units_per_workgroup > 0 ? units_per_workgroup : lp_pick_units(p.n_units, epilogue, k_split);
  p.n_sets = (p.n_units + nu - 1) / nu;
  p.k_split = k_split;
  p.trim256 = (k_split > 1 && epilogue != LP_EPI_PARTS) ? (trim_arg ? trim_arg - 1 : 24) : 0;  // default: partners 9 % short of an even share (tools/bench_lp_trim.py)
  const int64_t n_slots = (int64_t)p.n_sets * (k_split - 1) * kLpConsumers;
  p.flags = reinterpret_cast<int*>(workspace);
  DL_REQUIRE(n_slots * 4 <= kLpFlagBytes, "dl_linear_packed: %lld hand-over slots exceed the flag region", (long long)n_slots);
  p.parts = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kLpFlagBytes);
  p.err = err_flag;
  p.stamps = stamps;
  int rc = DL_ERR_ARG;
  if (dtype == DL_BF16) {
    switch (abl) {
      case 0: rc = lp_nu<bf16_t, 0>(p, nu, epilogue, as_stream(stream)); break;
#ifdef DL_LP_ABLATIONS
      case 1: rc = lp_nu<bf16_t, 1>(p, nu, epilogue, as_stream(stream)); break;
      case 2: rc = lp_nu<bf16_t, 2>(p, nu, epilogue, as_stream(stream)); break;
      case 3: rc = lp_nu<bf16_t, 3>(p, nu, epilogue, as_stream(stream)); break;
      case 7: rc = lp_nu<bf16_t, 7>(p, nu, epilogue, as_stream(stream)); break;
#endif
      default: set_error("dl_linear_packed: ablation %d is not built", abl); return DL_ERR_ARG;
    }
  } else {
    DL_REQUIRE(abl == 0, "dl_linear_packed: ablations are bf16 only");
    rc = lp_nu<f16_t, 0>(p, nu, epilogue, as_stream(stream));
  }
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_linear_packed");
  return DL_OK;
}

extern "C" int dl_linear_packed(const void* X, int64_t ldx, int x_packed, const void* Wp, void* Y, int64_t ldy, const void* resid, int64_t ldr, int M, int N,
                                int K, int epilogue, int units_per_workgroup, int k_split, void* workspace, int32_t* err_flag, int dtype, void* stream) {
  return lp_entry(X, ldx, x_packed, Wp, Y, ldy, resid, ldr, M, N, K, epilogue, units_per_workgroup, k_split, workspace, err_flag, nullptr, dtype, stream);
}

extern "C" int dl_linear_packed_stamped(const void* X, int64_t ldx, int x_packed, const void* Wp, void* Y, int64_t ldy, const void* resid, int64_t ldr, int M,
                                        int N, int K, int epilogue, int units_per_workgroup, int k_split, void* workspace, int32_t* err_flag, int64_t* stamps,
                                        int dtype, void* stream) {
  DL_REQUIRE(stamps, "dl_linear_packed_stamped: NULL stamp buffer");
  return lp_entry(X, ldx, x_packed, Wp, Y, ldy, resid, ldr, M, N, K, epilogue, units_per_workgroup, k_split, workspace, err_flag, reinterpret_cast<long long*>(stamps),
                  dtype, stream);
}
