// F9 (prefill): packed-varlen self-attention, causal (decoder, DML:1114-1122) or full (VisionPredictor,
// CTL:164-169).  No [T,T] score matrix, no padding, no mask tensor: raggedness comes from cu_seqlens.
//
// f16/bf16: flash-style MFMA kernel.  One 256-thread workgroup = 64 query rows of one (sequence, head),
// 16 rows per wave; K/V tiles of 64 keys are staged through LDS (K row-major, V transposed so that both
// MFMA B-operands are 16-byte LDS reads); S = Q K^T and O += P V on v_mfma_f32_16x16x32_{bf16,f16};
// online softmax in fp32 on the MFMA C layout (row = (lane>>4)*4 + reg, col = lane&15).
// f32: simple wave-per-query kernel (parity/debug path, exact fp32 arithmetic).
#include <stdlib.h>

#include <type_traits>

#include "dl_common.h"

namespace dl {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma16<bf16_t>(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma16<f16_t>(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// two fp32 -> one packed pair of the 16-bit type, round-to-nearest-even (bf16: v_cvt_pk_bf16_f32, the same bits as Elem<bf16_t>::from_f for finite values)
template <typename T>
__device__ __forceinline__ uint32_t pf_pack2(float a, float b) {
  if constexpr (Elem<T>::kBf16) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
  } else {
    return (uint32_t)Elem<T>::from_f(a) | ((uint32_t)Elem<T>::from_f(b) << 16);
  }
}

constexpr int kBN = 64, kPad = 8;

// tools/pf_timing.hip compiles this file with -DDL_PF_TIMING to stamp the phases of the LAST query tile of head 0 (100 MHz wall clock).
#ifdef DL_PF_TIMING
__device__ long long g_pf_stamps[16];
#define DL_PSTAMP(i)                                                                                          \
  do {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                               \
    if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_pf_stamps[i] = wall_clock64(); \
  } while (0)
#else
#define DL_PSTAMP(i)
#endif

// BN keys per K/V tile (64; 128 halves the tile count of the head_dim-64 towers).
// NW waves per workgroup, 16 query rows per wave (BM = 16 * NW): small prompts use fewer waves per workgroup so that the
// grid still covers the chip (T=170, 32 heads: NW=4 -> 96 workgroups, NW=1 -> 352).
template <typename T, int D, bool CAUSAL, int NW, int BN>
__global__ __launch_bounds__(NW * 64) void attn_prefill_mfma_kernel(const void* __restrict__ q_, const void* __restrict__ k_,
                                                                 const void* __restrict__ v_, int64_t q_rs, int64_t kv_rs,
                                                                 void* __restrict__ out_, int64_t out_rs,
                                                                 const int32_t* __restrict__ cu, int n_rep, float scale,
                                                                 const int32_t* __restrict__ kv_len, int64_t kv_sb, int64_t kv_sh) {
  using S = uint16_t;
  constexpr int KS = D / 32;   // MFMA k-steps over the head dim
  constexpr int DT = D / 16;   // 16-wide output column tiles
  constexpr int NT = BN / 16; // 16-key tiles per KV tile
  constexpr int LDK = D + kPad;
  constexpr int LDV = BN + kPad;
  constexpr int LDP = BN + kPad;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Ks = reinterpret_cast<S*>(smem);   // [BN][LDK]
  S* Vt = Ks + BN * LDK;               // [D][LDV]
  S* Ps = Vt + D * LDV;                 // [NW][16][LDP]
  constexpr int kBM = 16 * NW;
  constexpr int NT_ = NW * 64;

  DL_PSTAMP(0);
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  const int q0 = qt * kBM;
  if (q0 >= L) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int kvh = h / n_rep;
  const S* qb = reinterpret_cast<const S*>(q_) + (int64_t)tok0 * q_rs + (int64_t)h * D;
  // K/V either come from the same packed projection output as q (fresh prefill) or from the KV slab (chunk on a cache:
  // keys [0, off) were cached before, this chunk's keys sit at [off, off + L); query j sees keys <= off + j)
  const int off = kv_len ? kv_len[b] : 0;
  const int Lk = off + L;
  if (kv_len) kv_rs = D;
  const S* kb = kv_len ? reinterpret_cast<const S*>(k_) + (int64_t)b * kv_sb + (int64_t)kvh * kv_sh
                       : reinterpret_cast<const S*>(k_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const S* vb = kv_len ? reinterpret_cast<const S*>(v_) + (int64_t)b * kv_sb + (int64_t)kvh * kv_sh
                       : reinterpret_cast<const S*>(v_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;

  // Q fragments (A operand): row = lane&15, k = (lane>>4)*8 .. +8
  uint4 qf[KS];
  {
    const int qrow = q0 + w * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = make_uint4(0, 0, 0, 0);
      if (qrow < L) qf[ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * q_rs + ks * 32 + lg * 8);
    }
  }
  f32x4_t acc_o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc_o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
  }
  S* Pw = Ps + w * 16 * LDP;

  const int n_tiles = CAUSAL ? min((Lk + BN - 1) / BN, (q0 + kBM - 1 + off) / BN + 1) : (Lk + BN - 1) / BN;
  for (int jt = 0; jt < n_tiles; ++jt) {
    const int key0 = jt * BN;
    // ---- stage K (row-major) and V (transposed) tiles ----
    constexpr int CPR = D / 8;  // 16-byte chunks per row
#pragma unroll
    for (int it = 0; it < (BN * CPR) / NT_; ++it) {
      const int idx = it * NT_ + tid;
      const int key = idx / CPR, ch = idx % CPR;
      uint4 kv4 = make_uint4(0, 0, 0, 0), vv4 = make_uint4(0, 0, 0, 0);
      if (key0 + key < Lk) {
        kv4 = *reinterpret_cast<const uint4*>(kb + (int64_t)(key0 + key) * kv_rs + ch * 8);
        vv4 = *reinterpret_cast<const uint4*>(vb + (int64_t)(key0 + key) * kv_rs + ch * 8);
      }
      *reinterpret_cast<uint4*>(Ks + key * LDK + ch * 8) = kv4;
      const uint32_t vw[4] = {vv4.x, vv4.y, vv4.z, vv4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Vt[(ch * 8 + 2 * e) * LDV + key] = (S)(vw[e] & 0xffffu);
        Vt[(ch * 8 + 2 * e + 1) * LDV + key] = (S)(vw[e] >> 16);
      }
    }
    __syncthreads();
    if (jt == 0) DL_PSTAMP(1);  // Q fragments + first K/V tile staged

    // ---- S = Q K^T ----
    f32x4_t acc_s[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc_s[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (nt * 16 + lr) * LDK + ks * 32 + lg * 8);
        acc_s[nt] = mfma16<T>(qf[ks], kf, acc_s[nt]);
      }
    }
    if (jt == 0) DL_PSTAMP(2);  // S done
    // ---- online softmax on the C layout: row = lg*4 + r, col = nt*16 + lr ----
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + w * 16 + lg * 4 + r;
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ki = key0 + nt * 16 + lr;
        float s = acc_s[nt][r] * scale;
        if (ki >= Lk || (CAUSAL && ki > qi + off)) s = -INFINITY;
        acc_s[nt][r] = s;
        mx = fmaxf(mx, s);
      }
      mx = row16_max(mx);
      const float mn = fmaxf(m[r], mx);
      const float ms = mn == -INFINITY ? 0.f : mn;
      alpha[r] = __expf(m[r] - ms);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float p = __expf(acc_s[nt][r] - ms);
        acc_s[nt][r] = p;
        rs += p;
      }
      rs = row16_sum(rs);
      l[r] = l[r] * alpha[r] + rs;
      m[r] = mn;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha[r];
    if (jt == 0) DL_PSTAMP(3);  // softmax done
    // ---- P (C layout) -> LDS -> A layout ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Pw[(lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(acc_s[nt][r]);
    __syncthreads();
    if (jt == 0) DL_PSTAMP(4);  // P in LDS (barrier)
    // ---- O += P V ----
#pragma unroll
    for (int ks = 0; ks < BN / 32; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(Pw + lr * LDP + ks * 32 + lg * 8);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const uint4 vf = *reinterpret_cast<const uint4*>(Vt + (dt * 16 + lr) * LDV + ks * 32 + lg * 8);
        acc_o[dt] = mfma16<T>(pf, vf, acc_o[dt]);
      }
    }
    __syncthreads();
    if (jt == 0) DL_PSTAMP(5);  // PV done, tile 0 complete
    if (jt == n_tiles - 1) DL_PSTAMP(6);  // all tiles
  }
  // ---- epilogue ----
  S* ob = reinterpret_cast<S*>(out_) + (int64_t)tok0 * out_rs + (int64_t)h * D;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + w * 16 + lg * 4 + r;
    if (qi < L) {
      const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) ob[(int64_t)qi * out_rs + dt * 16 + lr] = Elem<T>::from_f(acc_o[dt][r] * inv);
    }
  }
  DL_PSTAMP(7);
}

// ---- software-pipelined variant for rows that span many K/V tiles (max_seqlen > 256).
// K/V tiles are double-buffered in LDS: while the MFMAs of tile j run, the global loads of tile j+1 are in flight in registers and
// are written to the other buffer before the single barrier of the iteration (the plain kernel above has three barriers and an
// exposed global-load round trip per tile).  V is transposed with 8-byte LDS writes of 4 consecutive keys (conflict-free: lanes of
// a quarter-wave write 128 contiguous bytes) instead of 2-byte scatters.  P is wave-private: no workgroup barrier around it.
template <typename T, int D, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn_prefill_mfma_pipe_kernel(const void* __restrict__ q_, const void* __restrict__ k_,
                                                                         const void* __restrict__ v_, int64_t q_rs, int64_t kv_rs,
                                                                         void* __restrict__ out_, int64_t out_rs,
                                                                         const int32_t* __restrict__ cu, int n_rep, float scale,
                                                                         const int32_t* __restrict__ kv_len, int64_t kv_sb, int64_t kv_sh) {
  using S = uint16_t;
  constexpr int KS = D / 32, DT = D / 16, NT = kBN / 16;
  constexpr int LDK = D + kPad, LDV = kBN + kPad, LDP = kBN + kPad;
  constexpr int kBM = 16 * NW, NT_ = NW * 64;
  constexpr int CPR = D / 8;                          // 16-byte chunks per K/V row
  constexpr int KIT = (kBN * CPR) / NT_;              // K chunks per thread per tile
  constexpr int VITEMS = (kBN / 4) * CPR;             // V items (4 keys x one chunk) per tile
  constexpr int VIT = (VITEMS + NT_ - 1) / NT_;       // V items per thread per tile
  constexpr int kTile = kBN * LDK + D * LDV;          // elements per K|V^T buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* bufs = reinterpret_cast<S*>(smem);               // [2][kTile]
  S* Ps = bufs + 2 * kTile;                           // [NW][16][LDP]

  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  const int q0 = qt * kBM;
  if (q0 >= L) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int kvh = h / n_rep;
  const S* qb = reinterpret_cast<const S*>(q_) + (int64_t)tok0 * q_rs + (int64_t)h * D;
  const int off = kv_len ? kv_len[b] : 0;
  const int Lk = off + L;
  if (kv_len) kv_rs = D;
  const S* kb = kv_len ? reinterpret_cast<const S*>(k_) + (int64_t)b * kv_sb + (int64_t)kvh * kv_sh
                       : reinterpret_cast<const S*>(k_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const S* vb = kv_len ? reinterpret_cast<const S*>(v_) + (int64_t)b * kv_sb + (int64_t)kvh * kv_sh
                       : reinterpret_cast<const S*>(v_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;

  uint4 kreg[KIT], vreg[VIT][4];
  auto fetch = [&](int key0) {  // tile -> registers (zeros beyond the sequence)
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = it * NT_ + tid;
      const int key = key0 + idx / CPR, ch = idx % CPR;
      kreg[it] = make_uint4(0, 0, 0, 0);
      if (key < Lk) kreg[it] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * kv_rs + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int item = it * NT_ + tid;  // key group fastest: lanes of a quarter-wave own 16 consecutive key groups of one chunk
      const int kg = item % (kBN / 4), ch = item / (kBN / 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = key0 + kg * 4 + j;
        vreg[it][j] = make_uint4(0, 0, 0, 0);
        if (item < VITEMS && key < Lk) vreg[it][j] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * kv_rs + ch * 8);
      }
    }
  };
  auto stash = [&](S* buf) {  // registers -> K (row-major) | V^T
    S* Ks = buf;
    S* Vt = buf + kBN * LDK;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = it * NT_ + tid;
      *reinterpret_cast<uint4*>(Ks + (idx / CPR) * LDK + (idx % CPR) * 8) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int item = it * NT_ + tid;
      if (item < VITEMS) {
        const int kg = item % (kBN / 4), ch = item / (kBN / 4);
        const uint32_t w0[4] = {vreg[it][0].x, vreg[it][0].y, vreg[it][0].z, vreg[it][0].w};
        const uint32_t w1[4] = {vreg[it][1].x, vreg[it][1].y, vreg[it][1].z, vreg[it][1].w};
        const uint32_t w2[4] = {vreg[it][2].x, vreg[it][2].y, vreg[it][2].z, vreg[it][2].w};
        const uint32_t w3[4] = {vreg[it][3].x, vreg[it][3].y, vreg[it][3].z, vreg[it][3].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // dims 2e, 2e+1 of the chunk: 4 keys each -> one 8-byte write per dim
          uint2 lo, hi;
          lo.x = (w0[e] & 0xffffu) | (w1[e] << 16);
          lo.y = (w2[e] & 0xffffu) | (w3[e] << 16);
          hi.x = (w0[e] >> 16) | (w1[e] & 0xffff0000u);
          hi.y = (w2[e] >> 16) | (w3[e] & 0xffff0000u);
          *reinterpret_cast<uint2*>(Vt + (ch * 8 + 2 * e) * LDV + kg * 4) = lo;
          *reinterpret_cast<uint2*>(Vt + (ch * 8 + 2 * e + 1) * LDV + kg * 4) = hi;
        }
      }
    }
  };

  const int n_tiles = CAUSAL ? min((Lk + kBN - 1) / kBN, (q0 + kBM - 1 + off) / kBN + 1) : (Lk + kBN - 1) / kBN;
  fetch(0);  // in flight while the Q fragments are loaded
  uint4 qf[KS];
  {
    const int qrow = q0 + w * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = make_uint4(0, 0, 0, 0);
      if (qrow < L) qf[ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * q_rs + ks * 32 + lg * 8);
    }
  }
  f32x4_t acc_o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc_o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
  }
  S* Pw = Ps + w * 16 * LDP;
  stash(bufs);
  __syncthreads();

  for (int jt = 0; jt < n_tiles; ++jt) {
    const int key0 = jt * kBN;
    const S* Ks = bufs + (jt & 1) * kTile;
    const S* Vt = Ks + kBN * LDK;
    const bool more = jt + 1 < n_tiles;
    if (more) fetch(key0 + kBN);
    // ---- S = Q K^T ----
    f32x4_t acc_s[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc_s[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (nt * 16 + lr) * LDK + ks * 32 + lg * 8);
        acc_s[nt] = mfma16<T>(qf[ks], kf, acc_s[nt]);
      }
    }
    // ---- online softmax on the C layout: row = lg*4 + r, col = nt*16 + lr ----
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + w * 16 + lg * 4 + r;
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ki = key0 + nt * 16 + lr;
        float sv = acc_s[nt][r] * scale;
        if (ki >= Lk || (CAUSAL && ki > qi + off)) sv = -INFINITY;
        acc_s[nt][r] = sv;
        mx = fmaxf(mx, sv);
      }
      mx = row16_max(mx);
      const float mn = fmaxf(m[r], mx);
      const float ms = mn == -INFINITY ? 0.f : mn;
      alpha[r] = __expf(m[r] - ms);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float p = __expf(acc_s[nt][r] - ms);
        acc_s[nt][r] = p;
        rs += p;
      }
      rs = row16_sum(rs);
      l[r] = l[r] * alpha[r] + rs;
      m[r] = mn;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha[r];
    // ---- P (C layout) -> wave-private LDS -> A layout: only this wave touches Pw, LDS is in-order per wave ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Pw[(lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(acc_s[nt][r]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- O += P V ----
#pragma unroll
    for (int ks = 0; ks < kBN / 32; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(Pw + lr * LDP + ks * 32 + lg * 8);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const uint4 vf = *reinterpret_cast<const uint4*>(Vt + (dt * 16 + lr) * LDV + ks * 32 + lg * 8);
        acc_o[dt] = mfma16<T>(pf, vf, acc_o[dt]);
      }
    }
    if (more) stash(bufs + ((jt + 1) & 1) * kTile);  // the other buffer: last read two iterations ago, before the previous barrier
    __syncthreads();
  }
  S* ob = reinterpret_cast<S*>(out_) + (int64_t)tok0 * out_rs + (int64_t)h * D;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + w * 16 + lg * 4 + r;
    if (qi < L) {
      const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) ob[(int64_t)qi * out_rs + dt * 16 + lr] = Elem<T>::from_f(acc_o[dt][r] * inv);
    }
  }
}

// ---- key-split variant for rows with FEW K/V tiles (fresh prefill: the 170 / 117 tokens a decoder layer >= 2 sees at head_dim 128;
// the 577 / 576 tokens of the CLIP tower / vision predictor at head_dim 64).  With 3-5 K/V tiles per row the plain kernel is one long
// dependent chain per workgroup (tools/pf_timing.hip, T=170: per tile 3.5 us of exposed staging + 2.1 us of S -> softmax -> P -> PV,
// three times, + 1.6 us of 2-byte stores).  Here a workgroup = RW row tiles x KW key tiles of waves: a ROUND stages KW*64 keys at
// once (one round trip; the next round's rows are already in flight in registers), each wave does S / online softmax / PV for its
// own 64 keys of the round, and at the end the KW partial (m, l, O) of a row tile are merged through LDS by the kw == 0 wave in key
// order (deterministic).  V is read coalesced like K and transposed with 8-byte LDS writes into an XOR-swizzled V^T image
// (conflict-free on both sides).  Output rows leave as 16-byte stores (transposed through LDS).
template <typename T, int D, bool CAUSAL, int RW, int KW>
__global__ __launch_bounds__(RW * KW * 64) void attn_prefill_keysplit_kernel(const void* __restrict__ q_, const void* __restrict__ k_,
                                                                             const void* __restrict__ v_, int64_t q_rs, int64_t kv_rs,
                                                                             void* __restrict__ out_, int64_t out_rs,
                                                                             const int32_t* __restrict__ cu, int n_rep, float scale) {
  using S = uint16_t;
  constexpr int KS = D / 32, DT = D / 16, NT = kBN / 16;
  constexpr int NKEY = KW * kBN;                 // keys per round
  constexpr int LDK = D + kPad, LDV = NKEY + kPad, LDP = kBN + kPad, LDO = D + kPad;
  constexpr int kBM = 16 * RW, NT_ = RW * KW * 64;
  constexpr int CPR = D / 8;
  constexpr int KIT = (NKEY * CPR) / NT_;        // K chunks per thread per round
  constexpr int VITEMS = (NKEY / 4) * CPR;       // V items (4 keys x one chunk) per round
  constexpr int VIT = (VITEMS + NT_ - 1) / NT_;  // V items per thread per round
  constexpr int SW = D == 64 ? 2 : 1;            // swizzle step: a half-wave spans 128 / D key pairs
  static_assert((NKEY * CPR) % NT_ == 0, "K staging must divide evenly");
  static_assert(D == 64 || D == 128, "swizzle derived for head_dim 64 / 128");
  constexpr int kPartF = (DT * 4 + 8) * 64;      // floats per partial: O[DT][4][64 lanes], m[4][64], l[4][64]
  constexpr int kSmemEl = NKEY * LDK + D * LDV + RW * KW * 16 * LDP;
  static_assert(RW * (KW - 1) * kPartF * 4 + RW * 16 * LDO * 2 <= kSmemEl * 2, "partials + output staging alias the tile buffers");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Ks = reinterpret_cast<S*>(smem);   // [NKEY][LDK]
  S* Vt = Ks + NKEY * LDK;              // [D][LDV], key pairs swizzled
  S* Ps = Vt + D * LDV;                 // [RW*KW][16][LDP]
  float* part = reinterpret_cast<float*>(smem);                               // after the last round: [RW][KW-1][kPartF]
  S* Os = reinterpret_cast<S*>(smem + RW * (KW - 1) * kPartF * 4);            // and [RW][16][LDO]

  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  const int q0 = qt * kBM;
  if (q0 >= L) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rw = w % RW, kw = w / RW;
  const int lr = lane & 15, lg = lane >> 4;
  const int kvh = h / n_rep;
  const S* qb = reinterpret_cast<const S*>(q_) + (int64_t)tok0 * q_rs + (int64_t)h * D;
  const S* kb = reinterpret_cast<const S*>(k_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const S* vb = reinterpret_cast<const S*>(v_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const int n_keys = CAUSAL ? min(L, q0 + kBM) : L;  // keys any row of this workgroup can see
  const int n_rounds = (n_keys + NKEY - 1) / NKEY;
  DL_PSTAMP(0);

  uint4 kreg[KIT], vreg[VIT][4];
  auto fetch = [&](int base) {  // round -> registers (zeros beyond n_keys)
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = it * NT_ + tid;
      const int key = base + idx / CPR, ch = idx % CPR;
      kreg[it] = make_uint4(0, 0, 0, 0);
      if (key < n_keys) kreg[it] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * kv_rs + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int item = it * NT_ + tid;  // chunk fastest: 128 / D ... 16 lanes read one whole row (coalesced like K)
      const int kg = item / CPR, ch = item % CPR;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = base + kg * 4 + j;
        vreg[it][j] = make_uint4(0, 0, 0, 0);
        if (item < VITEMS && key < n_keys) vreg[it][j] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * kv_rs + ch * 8);
      }
    }
  };
  auto stash = [&]() {  // registers -> K (row-major) | V^T (swizzled)
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = it * NT_ + tid;
      *reinterpret_cast<uint4*>(Ks + (idx / CPR) * LDK + (idx % CPR) * 8) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int item = it * NT_ + tid;
      if (item < VITEMS) {
        const int kg = item / CPR, ch = item % CPR;
        // V^T[dim][key]: the 16-byte key pairs of a row are XOR-swizzled by (dim / 16) within blocks of 8 pairs, so that the lanes
        // of a half-wave (rows 8 * LDV apart: only two bank offsets) spread over all banks
        const int pair = kg >> 1;
        const int col = (((pair & ~7) | ((pair ^ (SW * (ch >> 1))) & 7)) << 3) + ((kg & 1) << 2);
        const uint32_t w0[4] = {vreg[it][0].x, vreg[it][0].y, vreg[it][0].z, vreg[it][0].w};
        const uint32_t w1[4] = {vreg[it][1].x, vreg[it][1].y, vreg[it][1].z, vreg[it][1].w};
        const uint32_t w2[4] = {vreg[it][2].x, vreg[it][2].y, vreg[it][2].z, vreg[it][2].w};
        const uint32_t w3[4] = {vreg[it][3].x, vreg[it][3].y, vreg[it][3].z, vreg[it][3].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // dims 2e, 2e+1 of the chunk: 4 keys each -> one 8-byte write per dim
          uint2 lo, hi;
          lo.x = (w0[e] & 0xffffu) | (w1[e] << 16);
          lo.y = (w2[e] & 0xffffu) | (w3[e] << 16);
          hi.x = (w0[e] >> 16) | (w1[e] & 0xffff0000u);
          hi.y = (w2[e] >> 16) | (w3[e] & 0xffff0000u);
          *reinterpret_cast<uint2*>(Vt + (ch * 8 + 2 * e) * LDV + col) = lo;
          *reinterpret_cast<uint2*>(Vt + (ch * 8 + 2 * e + 1) * LDV + col) = hi;
        }
      }
    }
  };

  fetch(0);
  uint4 qf[KS];
  {
    const int qrow = q0 + rw * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = make_uint4(0, 0, 0, 0);
      if (qrow < L) qf[ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * q_rs + ks * 32 + lg * 8);
    }
  }
  f32x4_t acc_o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc_o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
  }
  S* Pw = Ps + w * 16 * LDP;
  const int row_keys = CAUSAL ? min(L, q0 + rw * 16 + 16) : L;  // keys this wave's rows can see
  stash();
  __syncthreads();
  DL_PSTAMP(1);  // first round staged

  for (int rd = 0; rd < n_rounds; ++rd) {
    const bool more = rd + 1 < n_rounds;
    if (more) fetch((rd + 1) * NKEY);
    if (rd == 1) DL_PSTAMP(8);  // (timing builds) round 1: next round's loads issued AND landed (the stamp drains vmcnt)
    const int key0 = rd * NKEY + kw * kBN;  // this wave's 64 keys of the round
    if (key0 < row_keys) {                  // wave-uniform
      f32x4_t acc_s[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc_s[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (kw * kBN + nt * 16 + lr) * LDK + ks * 32 + lg * 8);
          acc_s[nt] = mfma16<T>(qf[ks], kf, acc_s[nt]);
        }
      }
      float alpha[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + rw * 16 + lg * 4 + r;
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int ki = key0 + nt * 16 + lr;
          float sv = acc_s[nt][r] * scale;
          if (ki >= L || (CAUSAL && ki > qi)) sv = -INFINITY;
          acc_s[nt][r] = sv;
          mx = fmaxf(mx, sv);
        }
        mx = row16_max(mx);
        const float mn = fmaxf(m[r], mx);
        const float ms = mn == -INFINITY ? 0.f : mn;
        alpha[r] = __expf(m[r] - ms);
        float rs = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float p = __expf(acc_s[nt][r] - ms);
          acc_s[nt][r] = p;
          rs += p;
        }
        rs = row16_sum(rs);
        l[r] = l[r] * alpha[r] + rs;
        m[r] = mn;
      }
      if (rd > 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha[r];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[(lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(acc_s[nt][r]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ks = 0; ks < kBN / 32; ++ks) {
        const uint4 pf = *reinterpret_cast<const uint4*>(Pw + lr * LDP + ks * 32 + lg * 8);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int pair = kw * 8 + ks * 4 + lg;  // keys kw*64 + ks*32 + lg*8 .. +8 of the round, un-swizzled with dim / 16 = dt
          const uint4 vf = *reinterpret_cast<const uint4*>(Vt + (dt * 16 + lr) * LDV + (((pair & ~7) | ((pair ^ (SW * dt)) & 7)) << 3));
          acc_o[dt] = mfma16<T>(pf, vf, acc_o[dt]);
        }
      }
    }
    if (rd == 1) DL_PSTAMP(9);   // round 1: S, softmax, P, PV done
    __syncthreads();  // the round's K / V^T are dead
    if (rd == 1) DL_PSTAMP(10);  // barrier
    if (more) {
      stash();
      if (rd == 1) DL_PSTAMP(11);  // stash (registers -> K | V^T)
      __syncthreads();
      if (rd == 1) DL_PSTAMP(12);
    }
  }
  DL_PSTAMP(2);  // all rounds
  if (kw > 0) {
    float* pp = part + ((rw * (KW - 1)) + (kw - 1)) * kPartF;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[(dt * 4 + r) * 64 + lane] = acc_o[dt][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pp[(DT * 4 + r) * 64 + lane] = m[r];
      pp[(DT * 4 + 4 + r) * 64 + lane] = l[r];
    }
  }
  __syncthreads();
  DL_PSTAMP(3);  // partials exchanged
  if (kw > 0) return;
  // ---- merge in key order ----
#pragma unroll
  for (int j = 1; j < KW; ++j) {
    const float* pp = part + ((rw * (KW - 1)) + (j - 1)) * kPartF;
    float a_own[4], a_oth[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mo = pp[(DT * 4 + r) * 64 + lane], lo = pp[(DT * 4 + 4 + r) * 64 + lane];
      const float mn = fmaxf(m[r], mo);
      const float ms = mn == -INFINITY ? 0.f : mn;
      a_own[r] = __expf(m[r] - ms);
      a_oth[r] = __expf(mo - ms);
      l[r] = l[r] * a_own[r] + lo * a_oth[r];
      m[r] = mn;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] = acc_o[dt][r] * a_own[r] + pp[(dt * 4 + r) * 64 + lane] * a_oth[r];
  }
  // ---- epilogue: C layout -> LDS -> 16-byte row stores ----
  S* Ow = Os + rw * 16 * LDO;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) Ow[(lg * 4 + r) * LDO + dt * 16 + lr] = Elem<T>::from_f(acc_o[dt][r] * inv);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  S* ob = reinterpret_cast<S*>(out_) + (int64_t)tok0 * out_rs + (int64_t)h * D;
#pragma unroll
  for (int i = 0; i < (16 * CPR) / 64; ++i) {
    const int c = i * 64 + lane;
    const int row = c / CPR, ch = c % CPR;
    const int qi = q0 + rw * 16 + row;
    if (qi < L) *reinterpret_cast<uint4*>(ob + (int64_t)qi * out_rs + ch * 8) = *reinterpret_cast<const uint4*>(Ow + row * LDO + ch * 8);
  }
  DL_PSTAMP(7);
}

// ---- whole-row variant for head_dim 64, non-causal, 257..608 keys (round 6): the CLIP ViT-L/14-336 tower (577 tokens, CTL / clip_encoder.py:53-71) and the vision
// predictor (576, DML:1348-1359).  The key-split kernel above walks such a row in three rounds of 256 keys, and a round costs its 16 waves ~3.3 us of latency
// (tools/pf_timing_clip.hip: fetch -> S -> softmax -> P through LDS -> PV -> barrier -> stash -> barrier), 17 us per launch for 1.4 GFLOP.  Here ALL keys of the head are
// on chip at once -- 76 KiB of K and 76 KiB of V^T in matrix-core FRAGMENT order (a 16 x 32 fragment = one lane-linear KiB: conflict-free ds_read_b128, nothing to swizzle) --
// staged in one round trip: K by LDS-DMA (per-lane global addresses, no register pass), V through registers (4 keys x 8 dims per thread -> eight 8-byte writes: the
// transpose).  Then no barrier until the end.  Both products keep the per-QUERY operand in registers and take the matrix operand from LDS:
//   S^T = K Q^T  (A = K fragment, B = the wave's 16 query rows)      -> lane (lr, lg) holds S^T[key 16 kt + 4 lg + r][query lr]: all of a lane's values belong to ONE query
//   O^T = V^T P^T (A = V^T fragment, B = P^T straight from the S^T accumulators: no trip through LDS) -> lane holds O^T[dim 16 dt + 4 lg + r][query lr]
// so the online softmax is in-register: the row maximum crosses lanes (xor 16, xor 32) once per 32 keys, the row sum only at the end.  The k order inside a 32-key chunk is
// the accumulators' own (keys 4 lg + r of the chunk's two tiles); the V^T image is written in that order, so nothing is permuted at run time.
// 8 waves = 4 row tiles (64 query rows per workgroup) x 2 key halves; the halves' (m, l, O) meet through LDS once, in half order (deterministic).
template <typename T, int KW>
__global__ __launch_bounds__(256 * KW) void attn_prefill_whole_d64_kernel(const void* __restrict__ q_, const void* __restrict__ k_, const void* __restrict__ v_, int64_t q_rs,
                                                                     int64_t kv_rs, void* __restrict__ out_, int64_t out_rs, const int32_t* __restrict__ cu, int n_rep,
                                                                     float scale) {
  using S = uint16_t;
  constexpr int D = 64, KT = 38, KC = 19;  // up to 608 keys
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((address_space(1))) void glob_v;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Kf = reinterpret_cast<S*>(smem);  // [key tile][dims half][64 lanes][8]
  S* Vf = Kf + KT * 2 * 512;           // [dim tile][32-key chunk][64 lanes][8]
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  const int q0 = qt * 64;
  if (q0 >= L) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NWV = 4 * KW, NT_ = 64 * NWV;
  const int rw = w & 3, kw = w >> 2;
  const int lr = lane & 15, lg = lane >> 4;
  const int kvh = h / n_rep;
  const S* qb = reinterpret_cast<const S*>(q_) + (int64_t)tok0 * q_rs + (int64_t)h * D;
  const S* kb = reinterpret_cast<const S*>(k_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const S* vb = reinterpret_cast<const S*>(v_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const int n_kc = (L + 31) >> 5;  // 32-key chunks (= pairs of key tiles); keys past L are staged as copies of key L - 1 and masked
  DL_PSTAMP(0);
  // ---- K: one DMA piece per (key tile, dims half) ----
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8*)smem;
    for (int kt = w; kt < n_kc * 2; kt += NWV) {  // both dims halves of a key tile by the same wave, back to back: the second piece's 128-byte lines are the first one's
      int key = kt * 16 + lr;
      key = key < L ? key : L - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint32_t voff = (uint32_t)key * (uint32_t)kv_rs * 2u + (uint32_t)(ks * 32 + lg * 8) * 2u;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(kt * 2 + ks) * 1024u);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"((const glob_v*)kb), "s"(dst)
                     : "memory");
      }
    }
  }
  DL_PSTAMP(8);  // (timing builds) K pieces issued and landed
  // ---- V: 4 keys x 8 dims per task -> V^T fragments (the 8-byte slot of key quad (h2, lg') in lane 16 lg' + dim) ----
  {
    // every load of the thread's (up to two per pass) tasks is requested before the first transpose: the staging is one memory round trip deep, not one per task
    const int n_tasks = n_kc * 64;
    for (int t0 = tid; t0 < n_tasks; t0 += 2 * NT_) {
      uint4 vr[2][4];
      int tt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = t0 + u * NT_;
        tt[u] = t < n_tasks ? t : t0;  // (a thread without a second task re-loads its first: unconditional loads, nothing is written twice)
        const int qd = tt[u] >> 3, ch = tt[u] & 7;
        const int kc = qd >> 3, h2 = (qd >> 2) & 1, lgf = qd & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int key = kc * 32 + h2 * 16 + lgf * 4 + j;
          key = key < L ? key : L - 1;
          vr[u][j] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * kv_rs + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && t0 + NT_ >= n_tasks) break;
        const int qd = tt[u] >> 3, ch = tt[u] & 7;
        const int kc = qd >> 3, h2 = (qd >> 2) & 1, lgf = qd & 3;
        const uint32_t w0[4] = {vr[u][0].x, vr[u][0].y, vr[u][0].z, vr[u][0].w}, w1[4] = {vr[u][1].x, vr[u][1].y, vr[u][1].z, vr[u][1].w};
        const uint32_t w2[4] = {vr[u][2].x, vr[u][2].y, vr[u][2].z, vr[u][2].w}, w3[4] = {vr[u][3].x, vr[u][3].y, vr[u][3].z, vr[u][3].w};
        // Slot of dim `lrf` inside its 16-dim x 4-group fragment: 16 B per (group, dim), XOR-swizzled in the low three dim bits by (dim tile, dim bit 3).  Unswizzled, the
        // 64 lanes of one of these 8-byte writes (8 dim chunks x 8 key quads) all land on the same two bank pairs -- 32-way, 3 us of staging (first build); the swizzle is a
        // permutation inside each aligned 16-lane block, which keeps the consumers' ds_read_b128 conflict-free.
        S* frag = Vf + ((ch >> 1) * KC + kc) * 512 + lgf * 16 * 8 + h2 * 4;
        const int swz = ((ch >> 1) << 1 | (ch & 1)) & 7, hi8 = (ch & 1) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // dims 2e, 2e + 1 of the chunk: 4 keys each -> one 8-byte write per dim
          uint2 lo, hi;
          lo.x = (w0[e] & 0xffffu) | (w1[e] << 16);
          lo.y = (w2[e] & 0xffffu) | (w3[e] << 16);
          hi.x = (w0[e] >> 16) | (w1[e] & 0xffff0000u);
          hi.y = (w2[e] >> 16) | (w3[e] & 0xffff0000u);
          *reinterpret_cast<uint2*>(frag + (hi8 + ((2 * e) ^ swz)) * 8) = lo;
          *reinterpret_cast<uint2*>(frag + (hi8 + ((2 * e + 1) ^ swz)) * 8) = hi;
        }
      }
    }
  }
  DL_PSTAMP(9);  // V staged
  // ---- Q: the wave's 16 rows as the B operand (lane: query lr, dims 32 ks + 8 lg ..) ----
  uint4 qf[2];
  {
    int qrow = q0 + rw * 16 + lr;
    qrow = qrow < L ? qrow : L - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * q_rs + ks * 32 + lg * 8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
  __syncthreads();
  DL_PSTAMP(1);  // everything staged

  f32x4_t acc_o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc_o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float scale2 = scale * 1.44269504088896340736f;
  float m = -INFINITY, l = 0.f;  // (m in units of log2 e) l: this lane's share of the row sum (its 8 keys per chunk); the four lane groups meet at the end
  const int c0 = (n_kc * kw) / KW, c1 = (n_kc * (kw + 1)) / KW;  // this wave's share of the 32-key chunks
  const S* vrd[4];  // this lane's 16 bytes of a V^T fragment, per dim tile (the staging's swizzle)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vrd[dt] = Vf + dt * KC * 512 + (lg * 16 + (lr ^ (((dt << 1) | (lr >> 3)) & 7))) * 8;
  // one softmax update per NCH chunks (64 keys where the wave's share allows: half the cross-lane maxima and rescale decisions per key; a last single chunk otherwise)
  auto step = [&](int kc, auto nch_) {
    constexpr int NCH = decltype(nch_)::value;
    float sv[NCH][8];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      f32x4_t s0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint4 a0 = *reinterpret_cast<const uint4*>(Kf + ((2 * (kc + c)) * 2 + ks) * 512 + lane * 8);
        const uint4 a1 = *reinterpret_cast<const uint4*>(Kf + ((2 * (kc + c) + 1) * 2 + ks) * 512 + lane * 8);
        s0 = mfma16<T>(a0, qf[ks], s0);
        s1 = mfma16<T>(a1, qf[ks], s1);
      }
      // scores in units of log2(e): p = exp2(s' - m') (one v_exp_f32 each, no multiply), the roundings of P are the hardware's RNE for bf16
#pragma unroll
      for (int j = 0; j < 8; ++j) sv[c][j] = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale2;
      if ((kc + c) * 32 + 32 > L) {  // (wave-uniform) only the row's last chunk holds keys past L (staged as copies of key L - 1): masked
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if ((kc + c) * 32 + (j >> 2) * 16 + lg * 4 + (j & 3) >= L) sv[c][j] = -INFINITY;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(mx, sv[c][j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float ms = mn == -INFINITY ? 0.f : mn;
    const float alpha = __builtin_amdgcn_exp2f(m - ms);
    float rs = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sv[c][j] = __builtin_amdgcn_exp2f(sv[c][j] - ms);
        rs += sv[c][j];
      }
    l = l * alpha + rs;
    if (__any(mn != m)) {  // (wave-uniform branch) the running maximum moved for some query of the wave: rescale; after the first chunks it rarely does
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha;
    }
    m = mn;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      uint4 pf;
      pf.x = pf_pack2<T>(sv[c][0], sv[c][1]);
      pf.y = pf_pack2<T>(sv[c][2], sv[c][3]);
      pf.z = pf_pack2<T>(sv[c][4], sv[c][5]);
      pf.w = pf_pack2<T>(sv[c][6], sv[c][7]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint4 vf = *reinterpret_cast<const uint4*>(vrd[dt] + (kc + c) * 512);
        acc_o[dt] = mfma16<T>(vf, pf, acc_o[dt]);
      }
    }
  };
  {
    int kc = c0;
    for (; kc + 2 <= c1; kc += 2) step(kc, std::integral_constant<int, 2>{});
    if (kc < c1) step(kc, std::integral_constant<int, 1>{});
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  DL_PSTAMP(2);  // all keys
  // ---- the key ranges meet: ranges 1.. publish (m, l, O^T) in the K region (dead after the barrier), range 0 merges them in range order ----
  __syncthreads();
  float* part = reinterpret_cast<float*>(smem) + (rw * (KW - 1)) * 18 * 64;
  if (kw > 0) {
    float* pp = part + (kw - 1) * 18 * 64;
    pp[lane] = m;
    pp[64 + lane] = l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[(2 + dt * 4 + r) * 64 + lane] = acc_o[dt][r];
  }
  __syncthreads();
  if (kw > 0) return;
#pragma unroll
  for (int o = 0; o < KW - 1; ++o) {
    const float* pp = part + o * 18 * 64;
    const float m1 = pp[lane], l1 = pp[64 + lane];
    const float mn = fmaxf(m, m1);
    const float ms = mn == -INFINITY ? 0.f : mn;
    const float a0 = __builtin_amdgcn_exp2f(m - ms), a1 = __builtin_amdgcn_exp2f(m1 - ms);
    l = l * a0 + l1 * a1;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] = acc_o[dt][r] * a0 + pp[(2 + dt * 4 + r) * 64 + lane] * a1;
  }
  DL_PSTAMP(3);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const int qi = q0 + rw * 16 + lr;
  if (qi < L) {
    S* ob = reinterpret_cast<S*>(out_) + (int64_t)(tok0 + qi) * out_rs + (int64_t)h * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 o;
      o.x = (uint32_t)Elem<T>::from_f(acc_o[dt][0] * inv) | ((uint32_t)Elem<T>::from_f(acc_o[dt][1] * inv) << 16);
      o.y = (uint32_t)Elem<T>::from_f(acc_o[dt][2] * inv) | ((uint32_t)Elem<T>::from_f(acc_o[dt][3] * inv) << 16);
      *reinterpret_cast<uint2*>(ob + dt * 16 + lg * 4) = o;
    }
  }
  DL_PSTAMP(7);
}

// ---- whole-head variant of the kernel above for MANY (image, head) pairs (late round 6): a batched prefill's CLIP tower (configs[2] / [3]: 32 images x 16 heads).  The
// kernel above gives every 64-row block of a head its own workgroup, and each of them stages all of the head's keys again (ten times per head, 5120 workgroups at 32
// images: 195 us per layer, 9.5 % MFMA-busy).  Here ONE 16-wave workgroup owns an (image, head): K / V^T staged once (the same fragment images), wave w takes the query
// tiles w, w + 16, w + 32 TOGETHER -- every K and V^T fragment it reads from LDS feeds two or three MFMAs instead of one (the loop is LDS-read-bound otherwise) --, each
// wave sees every key of its tiles: no ranges to merge, one barrier.
template <typename T>
__global__ __launch_bounds__(1024) void attn_prefill_head_d64_kernel(const void* __restrict__ q_, const void* __restrict__ k_, const void* __restrict__ v_, int64_t q_rs,
                                                                    int64_t kv_rs, void* __restrict__ out_, int64_t out_rs, const int32_t* __restrict__ cu, int n_rep,
                                                                    float scale) {
  using S = uint16_t;
  constexpr int D = 64, KT = 38, KC = 19, NWV = 16, NT_ = 64 * NWV;  // up to 608 keys
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((address_space(1))) void glob_v;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Kf = reinterpret_cast<S*>(smem);  // [key tile][dims half][64 lanes][8]
  S* Vf = Kf + KT * 2 * 512;           // [dim tile][32-key chunk][64 lanes][8]
  const int h = blockIdx.x, b = blockIdx.y;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  if (L <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int kvh = h / n_rep;
  const S* qb = reinterpret_cast<const S*>(q_) + (int64_t)tok0 * q_rs + (int64_t)h * D;
  const S* kb = reinterpret_cast<const S*>(k_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const S* vb = reinterpret_cast<const S*>(v_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const int n_kc = (L + 31) >> 5;
  const int n_qt = (L + 15) >> 4;
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8*)smem;
    for (int kt = w; kt < n_kc * 2; kt += NWV) {
      int key = kt * 16 + lr;
      key = key < L ? key : L - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint32_t voff = (uint32_t)key * (uint32_t)kv_rs * 2u + (uint32_t)(ks * 32 + lg * 8) * 2u;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(kt * 2 + ks) * 1024u);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"((const glob_v*)kb), "s"(dst)
                     : "memory");
      }
    }
  }
  {
    const int n_tasks = n_kc * 64;
    for (int t0 = tid; t0 < n_tasks; t0 += 2 * NT_) {
      uint4 vr[2][4];
      int tt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = t0 + u * NT_;
        tt[u] = t < n_tasks ? t : t0;
        const int qd = tt[u] >> 3, ch = tt[u] & 7;
        const int kc = qd >> 3, h2 = (qd >> 2) & 1, lgf = qd & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int key = kc * 32 + h2 * 16 + lgf * 4 + j;
          key = key < L ? key : L - 1;
          vr[u][j] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * kv_rs + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && t0 + NT_ >= n_tasks) break;
        const int qd = tt[u] >> 3, ch = tt[u] & 7;
        const int kc = qd >> 3, h2 = (qd >> 2) & 1, lgf = qd & 3;
        const uint32_t w0[4] = {vr[u][0].x, vr[u][0].y, vr[u][0].z, vr[u][0].w}, w1[4] = {vr[u][1].x, vr[u][1].y, vr[u][1].z, vr[u][1].w};
        const uint32_t w2[4] = {vr[u][2].x, vr[u][2].y, vr[u][2].z, vr[u][2].w}, w3[4] = {vr[u][3].x, vr[u][3].y, vr[u][3].z, vr[u][3].w};
        S* frag = Vf + ((ch >> 1) * KC + kc) * 512 + lgf * 16 * 8 + h2 * 4;
        const int swz = ((ch >> 1) << 1 | (ch & 1)) & 7, hi8 = (ch & 1) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint2 lo, hi;
          lo.x = (w0[e] & 0xffffu) | (w1[e] << 16);
          lo.y = (w2[e] & 0xffffu) | (w3[e] << 16);
          hi.x = (w0[e] >> 16) | (w1[e] & 0xffff0000u);
          hi.y = (w2[e] >> 16) | (w3[e] & 0xffff0000u);
          *reinterpret_cast<uint2*>(frag + (hi8 + ((2 * e) ^ swz)) * 8) = lo;
          *reinterpret_cast<uint2*>(frag + (hi8 + ((2 * e + 1) ^ swz)) * 8) = hi;
        }
      }
    }
  }
  // ---- Q of the wave's (up to three) tiles, requested before the staging is waited for ----
  uint4 qf[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int qt = w + j * NWV;
    int qrow = (qt < n_qt ? qt : w) * 16 + lr;
    qrow = qrow < L ? qrow : L - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[j][ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * q_rs + ks * 32 + lg * 8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (w >= n_qt) return;
  const int ntl = (w + 2 * NWV < n_qt) ? 3 : (w + NWV < n_qt) ? 2 : 1;  // (wave-uniform) tiles this wave carries through the keys together

  const float scale2 = scale * 1.44269504088896340736f;
  typedef __attribute__((address_space(3))) const S lds_s;  // 32-bit LDS addresses: the loop is register-tight (three tiles' accumulators, scores and query rows)
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  auto lds_ld = [](lds_s* p_) -> uint4 {
    const u32x4_t v_ = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(p_);
    return make_uint4(v_.x, v_.y, v_.z, v_.w);
  };
  lds_s* const Kl = (lds_s*)Kf + lane * 8;
  lds_s* vrd[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vrd[dt] = (lds_s*)Vf + dt * KC * 512 + (lg * 16 + (lr ^ (((dt << 1) | (lr >> 3)) & 7))) * 8;
  auto run = [&](auto ntl_) {
    constexpr int NTL = decltype(ntl_)::value;
    f32x4_t acc_o[NTL][4];
    float m[NTL], l[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      m[j] = -INFINITY, l[j] = 0.f;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc_o[j][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    for (int kc = 0; kc < n_kc; ++kc) {
      // phase 1: the chunk's four K fragments, read once, against every tile's query rows (raw scores: the scale, in units of log2 e, rides in the exponent's fused
      // multiply-add below).  All tiles' scores first, then their softmax: interleaving them tile by tile costs registers the compiler then spills (110 vs 102 us)
      float sv[NTL][8];
      {
        uint4 a[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          a[0][ks] = lds_ld(Kl + ((2 * kc) * 2 + ks) * 512);
          a[1][ks] = lds_ld(Kl + ((2 * kc + 1) * 2 + ks) * 512);
        }
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          f32x4_t s0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s0 = mfma16<T>(a[0][ks], qf[j][ks], s0);
            s1 = mfma16<T>(a[1][ks], qf[j][ks], s1);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) sv[j][e] = e < 4 ? s0[e & 3] : s1[e & 3];
        }
      }
      const bool tail = kc * 32 + 32 > L;  // (wave-uniform) the row's last chunk holds keys past L (staged as copies of key L - 1): masked
      // phase 2: online softmax per tile, P^T packed for the second product
      uint4 pf[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        if (tail) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (kc * 32 + (e >> 2) * 16 + lg * 4 + (e & 3) >= L) sv[j][e] = -INFINITY;
        }
        float mx = sv[j][0];
#pragma unroll
        for (int e = 1; e < 8; ++e) mx = fmaxf(mx, sv[j][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        mx *= scale2;  // (scale2 > 0: the maximum of the scaled scores)
        const float mn = fmaxf(m[j], mx);
        const float ms = mn == -INFINITY ? 0.f : mn;
        const float alpha = __builtin_amdgcn_exp2f(m[j] - ms);
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sv[j][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[j][e], scale2, -ms));  // one VALU op for scale and shift
          rs += sv[j][e];
        }
        l[j] = l[j] * alpha + rs;
        if (__any(mn != m[j])) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[j][dt][r] *= alpha;
        }
        m[j] = mn;
        pf[j].x = pf_pack2<T>(sv[j][0], sv[j][1]);
        pf[j].y = pf_pack2<T>(sv[j][2], sv[j][3]);
        pf[j].z = pf_pack2<T>(sv[j][4], sv[j][5]);
        pf[j].w = pf_pack2<T>(sv[j][6], sv[j][7]);
      }
      // phase 3: each V^T fragment, read once, against every tile's P^T
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint4 vf = lds_ld(vrd[dt] + kc * 512);
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc_o[j][dt] = mfma16<T>(vf, pf[j], acc_o[j][dt]);
      }
    }
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      float lj = l[j];
      lj += __shfl_xor(lj, 16, 64);
      lj += __shfl_xor(lj, 32, 64);
      const float inv = lj > 0.f ? 1.0f / lj : 0.f;
      const int qi = (w + j * NWV) * 16 + lr;
      if (qi < L) {
        S* ob = reinterpret_cast<S*>(out_) + (int64_t)(tok0 + qi) * out_rs + (int64_t)h * D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 o;
          o.x = pf_pack2<T>(acc_o[j][dt][0] * inv, acc_o[j][dt][1] * inv);
          o.y = pf_pack2<T>(acc_o[j][dt][2] * inv, acc_o[j][dt][3] * inv);
          *reinterpret_cast<uint2*>(ob + dt * 16 + lg * 4) = o;
        }
      }
    }
  };
  if (ntl == 3) run(std::integral_constant<int, 3>{});
  else if (ntl == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 1>{});
}

// ---- whole-head variant for head_dim 128, causal, 65..256 rows (late round 6): the decoder's post-compaction layers of a BATCHED prefill (configs[2] / [3]: 32 requests
// x 32 heads x 158..214 rows, DML:1061-1122).  The plain kernel walks such a launch as 6144 small workgroups at 3.6 % MFMA-busy (143 us per layer,
// profiles/r06_configs2_prefill_mfma_util.txt).  Here ONE workgroup owns a (request, head): all of its keys are staged once -- K by LDS-DMA, V transposed through
// registers, both in matrix-core fragment order as in the head_dim-64 kernel above (64 + 64 KiB) -- and its eight waves take the 16-row query tiles, heaviest first
// (tile n - 1 - w, then tile w - the causal triangle pairs a long tile with a short one), each wave running the in-register online softmax over the chunks at or below its
// tile's diagonal.  No key ranges to merge, one barrier (after staging).  Same products and roundings as the head_dim-64 kernel: S^T = K Q^T, O^T = V^T P^T with P^T
// taken straight from the S^T accumulators, scores in units of log2 e, P rounded by the hardware's RNE.
template <typename T>
__global__ __launch_bounds__(512) void attn_prefill_whole_d128_causal_kernel(const void* __restrict__ q_, const void* __restrict__ k_, const void* __restrict__ v_, int64_t q_rs,
                                                                             int64_t kv_rs, void* __restrict__ out_, int64_t out_rs, const int32_t* __restrict__ cu, int n_rep,
                                                                             float scale) {
  using S = uint16_t;
  constexpr int D = 128, KT = 16, KC = 8, NWV = 8, NT_ = 64 * NWV;  // up to 256 keys
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((address_space(1))) void glob_v;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Kf = reinterpret_cast<S*>(smem);  // [key tile][dims quarter][64 lanes][8]
  S* Vf = Kf + KT * 4 * 512;           // [dim tile][32-key chunk][64 lanes][8]
  const int h = blockIdx.x, b = blockIdx.y;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  if (L <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int kvh = h / n_rep;
  const S* qb = reinterpret_cast<const S*>(q_) + (int64_t)tok0 * q_rs + (int64_t)h * D;
  const S* kb = reinterpret_cast<const S*>(k_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const S* vb = reinterpret_cast<const S*>(v_) + (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  const int n_qt = (L + 15) >> 4;
  // Few (request, head) pairs (one or two requests): gridDim.z = 2 workgroups share a head's query tiles, interleaved (tile n - 1 - slot, slot = 2 w + z): one tile per wave,
  // half the LDS reads per workgroup; each stages the keys up to ITS last tile only.
  const int Z = (int)gridDim.z, z = (int)blockIdx.z;
  if (z >= n_qt) return;
  const int n_kc = (min(L, (n_qt - z) * 16) + 31) >> 5;  // 32-key chunks this workgroup needs; keys past L are staged as copies of key L - 1 (they lie above every diagonal)
  // ---- K: one DMA piece per (key tile, dims quarter) ----
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8*)smem;
    for (int kt = w; kt < n_kc * 2; kt += NWV) {
      int key = kt * 16 + lr;
      key = key < L ? key : L - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t voff = (uint32_t)key * (uint32_t)kv_rs * 2u + (uint32_t)(ks * 32 + lg * 8) * 2u;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(kt * 4 + ks) * 1024u);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"((const glob_v*)kb), "s"(dst)
                     : "memory");
      }
    }
  }
  // ---- V: 4 keys x 8 dims per task -> V^T fragments (same slot arithmetic and swizzle as the head_dim-64 kernel, 16 dim chunks per key quad) ----
  {
    const int n_tasks = n_kc * 128;
    for (int t0 = tid; t0 < n_tasks; t0 += 2 * NT_) {
      uint4 vr[2][4];
      int tt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = t0 + u * NT_;
        tt[u] = t < n_tasks ? t : t0;
        const int qd = tt[u] >> 4, ch = tt[u] & 15;
        const int kc = qd >> 3, h2 = (qd >> 2) & 1, lgf = qd & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int key = kc * 32 + h2 * 16 + lgf * 4 + j;
          key = key < L ? key : L - 1;
          vr[u][j] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * kv_rs + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && t0 + NT_ >= n_tasks) break;
        const int qd = tt[u] >> 4, ch = tt[u] & 15;
        const int kc = qd >> 3, h2 = (qd >> 2) & 1, lgf = qd & 3;
        const uint32_t w0[4] = {vr[u][0].x, vr[u][0].y, vr[u][0].z, vr[u][0].w}, w1[4] = {vr[u][1].x, vr[u][1].y, vr[u][1].z, vr[u][1].w};
        const uint32_t w2[4] = {vr[u][2].x, vr[u][2].y, vr[u][2].z, vr[u][2].w}, w3[4] = {vr[u][3].x, vr[u][3].y, vr[u][3].z, vr[u][3].w};
        S* frag = Vf + ((ch >> 1) * KC + kc) * 512 + lgf * 16 * 8 + h2 * 4;
        const int swz = ((ch >> 1) << 1 | (ch & 1)) & 7, hi8 = (ch & 1) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint2 lo, hi;
          lo.x = (w0[e] & 0xffffu) | (w1[e] << 16);
          lo.y = (w2[e] & 0xffffu) | (w3[e] << 16);
          hi.x = (w0[e] >> 16) | (w1[e] & 0xffff0000u);
          hi.y = (w2[e] >> 16) | (w3[e] & 0xffff0000u);
          *reinterpret_cast<uint2*>(frag + (hi8 + ((2 * e) ^ swz)) * 8) = lo;
          *reinterpret_cast<uint2*>(frag + (hi8 + ((2 * e + 1) ^ swz)) * 8) = hi;
        }
      }
    }
  }
  // ---- Q of the wave's (up to two) tiles, requested before the staging is waited for: the tile's 16 rows as the B operand (lane: query lr, dims 32 ks + 8 lg ..) ----
  uint4 qpre[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = (w + u * NWV) * Z + z;  // slot: the first 8 Z slots take the tiles from the last (longest) one down, the rest (Z = 1, more than 8 tiles) from tile 0 up
    int qt = i < NWV * Z ? n_qt - 1 - i : i - NWV * Z;
    qt = (i < n_qt && qt >= 0) ? qt : 0;
    const int qi = qt * 16 + lr;
    const int qrow = qi < L ? qi : L - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qpre[u][ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * q_rs + ks * 32 + lg * 8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces (and its query rows) have landed
  __syncthreads();

  const float scale2 = scale * 1.44269504088896340736f;
  const S* vrd[8];  // this lane's 16 bytes of a V^T fragment, per dim tile (the staging's swizzle)
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) vrd[dt] = Vf + dt * KC * 512 + (lg * 16 + (lr ^ (((dt << 1) | (lr >> 3)) & 7))) * 8;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = (w + u * NWV) * Z + z;
    if (i >= n_qt) break;
    const int qt = i < NWV * Z ? n_qt - 1 - i : i - NWV * Z;  // heaviest tiles first; a wave's second tile is a short one
    uint4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qpre[u][ks];
    const int qi = qt * 16 + lr;
    f32x4_t acc_o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc_o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const int c1 = (qt >> 1) + 1;  // chunks that hold keys <= the tile's last row
    auto step = [&](int kc, auto nch_) {
      constexpr int NCH = decltype(nch_)::value;
      float sv[NCH][8];
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f32x4_t s0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 a0 = *reinterpret_cast<const uint4*>(Kf + ((2 * (kc + c)) * 4 + ks) * 512 + lane * 8);
          const uint4 a1 = *reinterpret_cast<const uint4*>(Kf + ((2 * (kc + c) + 1) * 4 + ks) * 512 + lane * 8);
          s0 = mfma16<T>(a0, qf[ks], s0);
          s1 = mfma16<T>(a1, qf[ks], s1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[c][j] = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale2;
        if ((kc + c) * 32 + 31 > qt * 16) {  // (wave-uniform) the chunk reaches past the tile's first row: causal mask, and the keys past L with it (key <= query < L)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if ((kc + c) * 32 + (j >> 2) * 16 + lg * 4 + (j & 3) > qi) sv[c][j] = -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, sv[c][j]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float ms = mn == -INFINITY ? 0.f : mn;
      const float alpha = __builtin_amdgcn_exp2f(m - ms);
      float rs = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sv[c][j] = __builtin_amdgcn_exp2f(sv[c][j] - ms);
          rs += sv[c][j];
        }
      l = l * alpha + rs;
      if (__any(mn != m)) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha;
      }
      m = mn;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        uint4 pf;
        pf.x = pf_pack2<T>(sv[c][0], sv[c][1]);
        pf.y = pf_pack2<T>(sv[c][2], sv[c][3]);
        pf.z = pf_pack2<T>(sv[c][4], sv[c][5]);
        pf.w = pf_pack2<T>(sv[c][6], sv[c][7]);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const uint4 vf = *reinterpret_cast<const uint4*>(vrd[dt] + (kc + c) * 512);
          acc_o[dt] = mfma16<T>(vf, pf, acc_o[dt]);
        }
      }
    };
    {
      int kc = 0;
      for (; kc + 2 <= c1; kc += 2) step(kc, std::integral_constant<int, 2>{});
      if (kc < c1) step(kc, std::integral_constant<int, 1>{});
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (qi < L) {
      S* ob = reinterpret_cast<S*>(out_) + (int64_t)(tok0 + qi) * out_rs + (int64_t)h * D;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        uint2 o;
        o.x = pf_pack2<T>(acc_o[dt][0] * inv, acc_o[dt][1] * inv);
        o.y = pf_pack2<T>(acc_o[dt][2] * inv, acc_o[dt][3] * inv);
        *reinterpret_cast<uint2*>(ob + dt * 16 + lg * 4) = o;
      }
    }
  }
}

// ---- generic path: one wave per query row, lanes over keys (scores) then over dims (output) ----
template <typename T>
__global__ __launch_bounds__(256) void attn_prefill_simple_kernel(const void* __restrict__ q_, const void* __restrict__ k_,
                                                                   const void* __restrict__ v_, int64_t q_rs, int64_t kv_rs,
                                                                   void* __restrict__ out_, int64_t out_rs,
                                                                   const int32_t* __restrict__ cu, int n_rep, float scale, int D,
                                                                   int causal, int max_len, const int32_t* __restrict__ kv_len, int64_t kv_sb,
                                                                   int64_t kv_sh) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);  // [4][D]
  float* sc = qs + 4 * D;                      // [4][max_len]
  const int h = blockIdx.y, b = blockIdx.z;
  const int tok0 = cu[b];
  const int L = cu[b + 1] - tok0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + w;
  const bool valid = qi < L;  // invalid waves still take part in the barriers
  const int kvh = h / n_rep;
  const int off = kv_len ? kv_len[b] : 0;
  const int Lk = off + L;
  if (kv_len) kv_rs = D;
  const int64_t kbase = kv_len ? (int64_t)b * kv_sb + (int64_t)kvh * kv_sh : (int64_t)tok0 * kv_rs + (int64_t)kvh * D;
  float* myq = qs + w * D;
  float* mys = sc + (int64_t)w * max_len;
  if (valid)
    for (int e = lane; e < D; e += 64) myq[e] = load1<T>(q_, ((int64_t)tok0 + qi) * q_rs + (int64_t)h * D + e);
  __syncthreads();
  const int nk = valid ? (causal ? min(qi + off + 1, Lk) : Lk) : 0;
  float mx = -INFINITY;
  for (int key = lane; key < nk; key += 64) {
    float a = 0.f;
    const int64_t ko = kbase + (int64_t)key * kv_rs;
    for (int e = 0; e < D; ++e) a += myq[e] * load1<T>(k_, ko + e);
    a *= scale;
    mys[key] = a;
    mx = fmaxf(mx, a);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int key = lane; key < nk; key += 64) {
    const float p = expf(mys[key] - mx);
    mys[key] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  if (!valid) return;
  const float inv = 1.0f / sum;
  for (int e = lane; e < D; e += 64) {
    float o = 0.f;
    for (int key = 0; key < nk; ++key) o += mys[key] * load1<T>(v_, kbase + (int64_t)key * kv_rs + e);
    store1<T>(out_, ((int64_t)tok0 + qi) * out_rs + (int64_t)h * D + e, o * inv);
  }
}

template <typename T, int D>
static void launch_mfma(const void* q, const void* k, const void* v, int64_t q_rs, int64_t kv_rs, void* out, int64_t out_rs,
                        const int32_t* cu, int B, int max_seqlen, int n_heads, int n_rep, int causal, hipStream_t st,
                        const int32_t* kv_len = nullptr, int64_t kv_sb = 0, int64_t kv_sh = 0) {
  const float scale = 1.0f / sqrtf((float)D);
  // tools/bench_attn_prefill.py: 2 waves (32 query rows) per workgroup win for short rows (more workgroups, 16-180 us range),
  // 4 waves for long ones; 1 wave never wins (each workgroup then stages whole K/V tiles alone)
  int nw = max_seqlen <= 256 ? 2 : 4;
  if (const char* e = getenv("DL_PF_NW")) nw = atoi(e) == 1 ? 1 : (atoi(e) == 2 ? 2 : 4);  // tuning experiments only
  // tools/bench_attn_prefill.py: the software-pipelined kernel wins once rows span several K/V tiles (T=631: 61 -> 46 us;
  // B=32 T=700: 974 -> 708 us), is a wash at T=170 (3 tiles, latency of the dependent S -> softmax -> PV chain dominates) and loses
  // at B=32 T=215, where 3 resident workgroups per CU hide the loads better than one double-buffered one
  // (head_dim 64 -- CLIP, the vision predictor -- stays on the plain kernel: 31.6 vs 34.6 us at T=577, its tiles are too small to pay
  // for the second buffer)
  bool pipe = max_seqlen > 256 && D == 128;
  if (const char* e = getenv("DL_PF_PIPE")) pipe = atoi(e) != 0;  // tuning experiments only
  // K/V tile size of the plain kernel: 128 keys for the head_dim-64 towers (CLIP T=577: 27.3 -> 22 us per layer), else 64.  (One 192-key
  // tile for the T=170 decoder rows was no faster at B=1 -- 15.4 vs 15.1 us -- and 2x slower at B=8: 116 KB of LDS, one workgroup per CU.)
  const int bn = (D == 64 && max_seqlen > 128) ? 128 : 64;
#define DL_LAUNCH_PLAIN_BN(NWV, CAUS, BNV)                                                                                               \
  {                                                                                                                                      \
    const size_t smem = (size_t)(BNV * (D + kPad) + D * (BNV + kPad) + NWV * 16 * (BNV + kPad)) * 2;                                     \
    auto kfn = attn_prefill_mfma_kernel<T, D, CAUS, NWV, BNV>;                                                                           \
    static bool attr_set = false;                                                                                                        \
    if (!attr_set && smem > 64 * 1024) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);             \
      attr_set = true;                                                                                                                   \
    }                                                                                                                                    \
    hipLaunchKernelGGL(kfn, grid, dim3(NWV * 64), smem, st, q, k, v, q_rs, kv_rs, out, out_rs, cu, n_rep, scale, kv_len, kv_sb, kv_sh);   \
  }
#define DL_LAUNCH_PLAIN(NWV, CAUS)                                                                                                       \
  {                                                                                                                                      \
    bool done = false;                                                                                                                   \
    if constexpr (D == 64 && NWV == 4) {                                                                                                 \
      if (bn == 128) {                                                                                                                   \
        DL_LAUNCH_PLAIN_BN(4, CAUS, 128) done = true;                                                                                    \
      }                                                                                                                                  \
    }                                                                                                                                    \
    if (!done) DL_LAUNCH_PLAIN_BN(NWV, CAUS, 64)                                                                                         \
  }
#define DL_LAUNCH_PF(NWV, CAUS)                                                                                                          \
  {                                                                                                                                      \
    const dim3 grid((unsigned)((max_seqlen + 16 * NWV - 1) / (16 * NWV)), (unsigned)n_heads, (unsigned)B);                              \
    if (pipe && NWV > 1) {                                                                                                               \
      const size_t smem = (size_t)(2 * (kBN * (D + kPad) + D * (kBN + kPad)) + NWV * 16 * (kBN + kPad)) * 2;                             \
      auto kfn = attn_prefill_mfma_pipe_kernel<T, D, CAUS, (NWV > 1 ? NWV : 2)>;                                                         \
      static bool attr_set = false;                                                                                                      \
      if (!attr_set && smem > 64 * 1024) {                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);           \
        attr_set = true;                                                                                                                 \
      }                                                                                                                                  \
      hipLaunchKernelGGL(kfn, grid, dim3(NWV * 64), smem, st, q, k, v, q_rs, kv_rs, out, out_rs, cu, n_rep, scale, kv_len, kv_sb, kv_sh); \
    } else {                                                                                                                             \
      DL_LAUNCH_PLAIN(NWV, CAUS)                                                                                                         \
    }                                                                                                                                    \
  }
  // key-split kernel (one workgroup per CU: 90-120 KB of LDS): fresh prefill while the launch fits the chip in a single round of
  // workgroups -- at larger batches the plain kernel's three resident workgroups per CU win.
  //   head_dim 128, causal, 64 < rows <= 192 (decoder layers >= 2 at B=1): 32 rows x (2 | 3) key tiles, one round  (T=170: 15.2 -> 8.9 us)
  //   head_dim 64, full, rows > 128 (CLIP tower, vision predictor): 64 rows x 4 key tiles, rounds of 256 keys
  if constexpr (D == 64) {
    // whole-row kernel (round 6): all keys of a head on chip in fragment order, no rounds (CLIP tower: 17.4 -> see profiles/r06_attn_whole_row.txt)
    bool whole = !causal && !kv_len && max_seqlen > 256 && max_seqlen <= 608 && q_rs % 8 == 0 && kv_rs % 8 == 0 && out_rs % 4 == 0 &&
                 (int64_t)max_seqlen * kv_rs * 2 < ((int64_t)1 << 31);
    if (const char* e = getenv("DL_PF_WHOLE")) whole = whole && atoi(e) != 0;  // A/B against the key-split kernel
    if (whole) {
      const size_t smem = (size_t)(38 * 2 + 4 * 19) * 1024;
      // many (image, head) pairs: one 16-wave workgroup per pair, every fragment read feeding two or three query tiles (tools/bench_attn_prefill_batched.py --clip)
      static const int head_pairs = getenv("DL_PF_HEAD64_MIN") ? atoi(getenv("DL_PF_HEAD64_MIN")) : 256;
      if ((int64_t)B * n_heads >= head_pairs) {
        auto kfn = attn_prefill_head_d64_kernel<T>;
        static bool attr_set = false;
        if (!attr_set) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
          attr_set = true;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)n_heads, (unsigned)B), dim3(1024), smem, st, q, k, v, q_rs, kv_rs, out, out_rs, cu, n_rep, scale);
        return;
      }
      int kwv = 4;
      if (const char* e = getenv("DL_PF_WHOLE_KW")) kwv = atoi(e) == 2 ? 2 : 4;  // tuning experiments only
#define DL_LAUNCH_WHOLE(KWV)                                                                                                             \
  {                                                                                                                                      \
    auto kfn = attn_prefill_whole_d64_kernel<T, KWV>;                                                                                    \
    static bool attr_set = false;                                                                                                        \
    if (!attr_set) {                                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);             \
      attr_set = true;                                                                                                                   \
    }                                                                                                                                    \
    hipLaunchKernelGGL(kfn, dim3((unsigned)((max_seqlen + 63) / 64), (unsigned)n_heads, (unsigned)B), dim3(256 * KWV), smem, st, q, k, v, q_rs, kv_rs, out, out_rs, cu, \
                       n_rep, scale);                                                                                                    \
  }
      if (kwv == 2) DL_LAUNCH_WHOLE(2) else DL_LAUNCH_WHOLE(4)
#undef DL_LAUNCH_WHOLE
      return;
    }
  }
  if constexpr (D == 128) {
    // whole-head kernel (late round 6): the compacted layers of a prefill -- one workgroup per (request, head).  tools/bench_attn_prefill_batched.py, 158..214 rows, us:
    // 32 requests 47.8 (plain kernel 140.0), 8: 16.7 (54.1), 4: 13.3 (33.5), 2: 11.4 (24.0), and even ONE request's 32 workgroups on 32 CUs 8.6 against the key-split
    // kernel's 192 workgroups 9.7 (T = 170); with two workgroups per head at <= 128 pairs (below): 7.2 / 8.5 / 9.7 at one / two / four requests
    static const int min_pairs = getenv("DL_PF_WHOLE128_MIN") ? atoi(getenv("DL_PF_WHOLE128_MIN")) : 1;  // (tuning: tools/bench_attn_prefill_batched.py)
    bool whole = causal && !kv_len && max_seqlen > 64 && max_seqlen <= 256 && (int64_t)B * n_heads >= min_pairs && q_rs % 8 == 0 && kv_rs % 8 == 0 && out_rs % 4 == 0 &&
                 (int64_t)max_seqlen * kv_rs * 2 < ((int64_t)1 << 31);
    if (const char* e = getenv("DL_PF_WHOLE128")) whole = whole && atoi(e) != 0;  // A/B against the plain kernel
    if (whole) {
      const size_t smem = (size_t)(16 * 4 + 8 * 8) * 1024;
      auto kfn = attn_prefill_whole_d128_causal_kernel<T>;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
      }
      static const int z_pairs = getenv("DL_PF_WHOLE128_ZPAIRS") ? atoi(getenv("DL_PF_WHOLE128_ZPAIRS")) : 128;  // two workgroups per head up to this many (request, head) pairs: 1 request 8.6 -> 7.2 us, 2: 11.5 -> 8.5, 4: 13.2 -> 9.7; 8: 16.7 -> 19.2 (one per head stays)
      const unsigned zs = ((int64_t)B * n_heads <= z_pairs) ? 2u : 1u;
      hipLaunchKernelGGL(kfn, dim3((unsigned)n_heads, (unsigned)B, zs), dim3(512), smem, st, q, k, v, q_rs, kv_rs, out, out_rs, cu, n_rep, scale);
      return;
    }
  }
  if constexpr (D == 128 || D == 64) {
    bool ksplit = false;
    if (D == 128) ksplit = causal && !kv_len && max_seqlen > 64 && max_seqlen <= 192 && (int64_t)B * n_heads * ((max_seqlen + 31) / 32) <= 256;
    if (D == 64) ksplit = !causal && !kv_len && max_seqlen > 128 && (int64_t)B * n_heads * ((max_seqlen + 63) / 64) <= 256;
    if (const char* e = getenv("DL_PF_KSPLIT")) ksplit = ksplit && atoi(e) != 0;  // tuning experiments only
    if (ksplit) {
#define DL_LAUNCH_KS(CAUS, RWV, KWV)                                                                                                     \
  {                                                                                                                                      \
    const dim3 grid((unsigned)((max_seqlen + 16 * RWV - 1) / (16 * RWV)), (unsigned)n_heads, (unsigned)B);                              \
    const size_t smem = (size_t)(KWV * 64 * (D + kPad) + D * (KWV * 64 + kPad) + RWV * KWV * 16 * (kBN + kPad)) * 2;                     \
    auto kfn = attn_prefill_keysplit_kernel<T, D, CAUS, RWV, KWV>;                                                                       \
    static bool attr_set = false;                                                                                                        \
    if (!attr_set) {                                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);             \
      attr_set = true;                                                                                                                   \
    }                                                                                                                                    \
    hipLaunchKernelGGL(kfn, grid, dim3(RWV * KWV * 64), smem, st, q, k, v, q_rs, kv_rs, out, out_rs, cu, n_rep, scale);                  \
  }
      if constexpr (D == 128) {
        if (max_seqlen <= 128) DL_LAUNCH_KS(true, 2, 2) else DL_LAUNCH_KS(true, 2, 3)
      } else {
        DL_LAUNCH_KS(false, 4, 4)
      }
#undef DL_LAUNCH_KS
      return;
    }
  }
  if (causal) {
    if (nw == 4) DL_LAUNCH_PF(4, true) else if (nw == 2) DL_LAUNCH_PF(2, true) else DL_LAUNCH_PF(1, true)
  } else {
    if (nw == 4) DL_LAUNCH_PF(4, false) else if (nw == 2) DL_LAUNCH_PF(2, false) else DL_LAUNCH_PF(1, false)
  }
#undef DL_LAUNCH_PF
#undef DL_LAUNCH_PLAIN
#undef DL_LAUNCH_PLAIN_BN
}

}  // namespace dl

using namespace dl;

static int attn_prefill_impl(const char* who, const void* q, const void* k, const void* v, int64_t q_row_stride, int64_t kv_row_stride,
                             void* out, int64_t out_row_stride, const int32_t* cu_seqlens, int B, int max_seqlen, int max_kv_len, int n_heads,
                             int n_kv_heads, int head_dim, int causal, int dtype, void* stream, const int32_t* kv_len, int64_t kv_sb,
                             int64_t kv_sh) {
  DL_REQUIRE(B > 0 && max_seqlen >= 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0, "%s: bad shape", who);
  if (max_seqlen == 0) return DL_OK;  // an empty input is a no-op, whatever its (possibly NULL) pointers
  DL_REQUIRE(q && k && v && out && cu_seqlens, "%s: NULL pointer", who);
  hipStream_t st = as_stream(stream);
  const int n_rep = n_heads / n_kv_heads;
  if (dtype == DL_F32) {
    DL_REQUIRE(head_dim > 0 && head_dim <= 256, "%s: head_dim=%d unsupported for f32", who, head_dim);
    DL_REQUIRE(max_kv_len <= 8192, "%s: f32 path supports up to 8192 keys", who);
    const size_t smem = (size_t)(4 * head_dim + 4 * (size_t)max_kv_len) * sizeof(float);
    hipLaunchKernelGGL((attn_prefill_simple_kernel<f32_t>), dim3((unsigned)((max_seqlen + 3) / 4), (unsigned)n_heads, (unsigned)B),
                       dim3(256), smem, st, q, k, v, q_row_stride, kv_row_stride, out, out_row_stride, cu_seqlens, n_rep,
                       1.0f / sqrtf((float)head_dim), head_dim, causal, max_kv_len, kv_len, kv_sb, kv_sh);
  } else if (dtype == DL_F16 || dtype == DL_BF16) {
    DL_REQUIRE(head_dim == 32 || head_dim == 64 || head_dim == 128, "%s: head_dim=%d unsupported (32, 64 or 128)", who, head_dim);
    DL_REQUIRE(q_row_stride % 8 == 0 && kv_row_stride % 8 == 0, "%s: row strides must be multiples of 8 elements", who);
#define DL_PF_ARGS q, k, v, q_row_stride, kv_row_stride, out, out_row_stride, cu_seqlens, B, max_seqlen, n_heads, n_rep, causal, st, kv_len, kv_sb, kv_sh
    if (dtype == DL_BF16) {
      if (head_dim == 128) launch_mfma<bf16_t, 128>(DL_PF_ARGS); else if (head_dim == 64) launch_mfma<bf16_t, 64>(DL_PF_ARGS); else launch_mfma<bf16_t, 32>(DL_PF_ARGS);
    } else {
      if (head_dim == 128) launch_mfma<f16_t, 128>(DL_PF_ARGS); else if (head_dim == 64) launch_mfma<f16_t, 64>(DL_PF_ARGS); else launch_mfma<f16_t, 32>(DL_PF_ARGS);
    }
#undef DL_PF_ARGS
  } else {
    set_error("%s: unsupported dtype %d", who, dtype);
    return DL_ERR_ARG;
  }
  DL_CHECK_LAUNCH(who);
  return DL_OK;
}

extern "C" int dl_attn_prefill(const void* q, const void* k, const void* v, int64_t q_row_stride, int64_t kv_row_stride, void* out,
                               int64_t out_row_stride, const int32_t* cu_seqlens, int B, int max_seqlen, int n_heads, int n_kv_heads,
                               int head_dim, int causal, int dtype, void* stream) {
  return attn_prefill_impl("dl_attn_prefill", q, k, v, q_row_stride, kv_row_stride, out, out_row_stride, cu_seqlens, B, max_seqlen, max_seqlen,
                           n_heads, n_kv_heads, head_dim, causal, dtype, stream, nullptr, 0, 0);
}

extern "C" int dl_attn_prefill_cached(const void* q, int64_t q_row_stride, const void* k_slab, const void* v_slab, int64_t slab_stride_b,
                                      int64_t slab_stride_h, const int32_t* kv_len, void* out, int64_t out_row_stride,
                                      const int32_t* cu_seqlens, int B, int max_seqlen, int max_kv_len, int n_heads, int n_kv_heads,
                                      int head_dim, int dtype, void* stream) {
  DL_REQUIRE(kv_len != nullptr, "dl_attn_prefill_cached: kv_len is NULL");
  DL_REQUIRE(max_kv_len >= max_seqlen, "dl_attn_prefill_cached: max_kv_len must bound kv_len[b] + row length");
  return attn_prefill_impl("dl_attn_prefill_cached", q, k_slab, v_slab, q_row_stride, head_dim, out, out_row_stride, cu_seqlens, B, max_seqlen,
                           max_kv_len, n_heads, n_kv_heads, head_dim, 1, dtype, stream, kv_len, slab_stride_b, slab_stride_h);
}
