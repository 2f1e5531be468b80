// N5 -- training-time ops of Dynamic-LLaVA, fused for gfx950:
//   * attention with a differentiable keep POLICY over the keys (`scaled_dot_product_attention_with_policy` +
//     `softmax_with_policy`, DML:913-970): forward and backward without ever materialising the [B,H,N,N] fp32 score / probability
//     tensors the reference builds (5 of them per layer in the forward alone);
//   * the Gumbel hard keep mask (`F.gumbel_softmax(log_probs, tau, hard=True)[:, :, 0:1] * prev_decision`, DML:1868-1876), forward
//     and backward, with the noise drawn by the caller (torch's generator, so the random stream is the reference's).
//
// Math (per batch row b, head h; s_ij = scale * q_i.k_j + bias_ij, -inf where masked):
//   m_i   = max_j s_ij                         (over every visible key, whatever its policy -- DML:921)
//   pe_ij = exp(s_ij - m_i) * p'_ij,           p'_ij = policy_j, except p'_ii = 1 (a dropped token still sees itself, DML:916-920)
//   Dn_i  = sum_j pe_ij + eps,                 A_ij = (pe_ij + eps/N) / Dn_i  for EVERY j < N, masked ones included (DML:928)
//   o_i   = sum_j A_ij v_j = (sum_j pe_ij v_j + (eps/N) * sum_j v_j) / Dn_i
// Backward (delta_i = dO_i . o_i, dA_ij = dO_i . v_j, dE_ij = (dA_ij - delta_i) / Dn_i):
//   dV_j = sum_i A_ij dO_i = sum_i (pe_ij / Dn_i) dO_i + (eps/N) * sum_i dO_i / Dn_i
//   dS_ij = dE_ij * pe_ij;  dQ_i = scale * sum_j dS_ij k_j;  dK_j = scale * sum_i dS_ij q_i
//   dpolicy_j = sum_h sum_{i != j} dE_ij * exp(s_ij - m_i)
// (the gradient through max_j -- O(eps) because softmax is shift invariant up to the eps terms -- is not propagated.)
//
// Kernels: MFMA 16x16x32 flash-style tiles, 64 query rows x 64 keys, 4 waves.  The backward is two deterministic passes (no atomics):
// one workgroup per KEY tile accumulates dK / dV / dpolicy over the query tiles, one per QUERY tile accumulates dQ over the key tiles.
// Transposed operands (V^T, K^T, Q^T, dO^T) are built in LDS from coalesced row loads with 8-byte writes into an XOR-swizzled image.
#include <type_traits>

#include "dl_common.h"

namespace dl {

typedef __bf16 tp_bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 tp_f16x8_t __attribute__((ext_vector_type(8)));
typedef float tp_f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ tp_f32x4_t tp_mfma(const uint4& a, const uint4& b, tp_f32x4_t c);
template <>
__device__ __forceinline__ tp_f32x4_t tp_mfma<bf16_t>(const uint4& a, const uint4& b, tp_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tp_bf16x8_t, a), __builtin_bit_cast(tp_bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ tp_f32x4_t tp_mfma<f16_t>(const uint4& a, const uint4& b, tp_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tp_f16x8_t, a), __builtin_bit_cast(tp_f16x8_t, b), c, 0, 0, 0);
}

constexpr int kTpTile = 64, kTpPad = 8, kTpThreads = 256;

struct TpStrides {
  int64_t b, h, l;  // element strides of a [B,H,L,d] view (d contiguous)
};

// ---- tile staging: 64 rows x D of a [.., L, D] operand -> LDS, row-major and / or transposed ----
template <int D, int NTHR = kTpThreads>
struct TpStage {
  static constexpr int CPR = D / 8;                                   // 16-byte chunks per row
  static constexpr int RIT = (kTpTile * CPR) / NTHR;                  // row-major chunks per thread
  static constexpr int TITEMS = (kTpTile / 4) * CPR;                  // transposed items (4 rows x one chunk)
  static constexpr int TIT = (TITEMS + NTHR - 1) / NTHR;              // per thread
  static_assert((kTpTile * CPR) % NTHR == 0, "row staging must divide evenly");
  static constexpr int SW = D == 64 ? 2 : 1;                          // swizzle step (a half-wave spans 128 / D row pairs)
  static constexpr int LDR = D + kTpPad, LDT = kTpTile + kTpPad;
};

// rows [row0, row0+64) of x (row stride sl), zeros beyond n_rows
template <int D, int NTHR = kTpThreads>
__device__ __forceinline__ void tp_fetch_rows(const uint16_t* __restrict__ x, int64_t sl, int row0, int n_rows, int tid, uint4 (&r)[TpStage<D, NTHR>::RIT]) {
  using St = TpStage<D, NTHR>;
#pragma unroll
  for (int it = 0; it < St::RIT; ++it) {
    const int idx = it * NTHR + tid;
    const int row = row0 + idx / St::CPR, ch = idx % St::CPR;
    r[it] = make_uint4(0, 0, 0, 0);
    if (row < n_rows) r[it] = *reinterpret_cast<const uint4*>(x + (int64_t)row * sl + ch * 8);
  }
}
template <int D, int NTHR = kTpThreads>
__device__ __forceinline__ void tp_stash_rows(uint16_t* __restrict__ Xs, int tid, const uint4 (&r)[TpStage<D, NTHR>::RIT]) {
  using St = TpStage<D, NTHR>;
#pragma unroll
  for (int it = 0; it < St::RIT; ++it) {
    const int idx = it * NTHR + tid;
    *reinterpret_cast<uint4*>(Xs + (idx / St::CPR) * St::LDR + (idx % St::CPR) * 8) = r[it];
  }
}
template <int D, int NTHR = kTpThreads>
__device__ __forceinline__ void tp_fetch_t(const uint16_t* __restrict__ x, int64_t sl, int row0, int n_rows, int tid, uint4 (&r)[TpStage<D, NTHR>::TIT][4]) {
  using St = TpStage<D, NTHR>;
#pragma unroll
  for (int it = 0; it < St::TIT; ++it) {
    const int item = it * NTHR + tid;  // chunk fastest: the lanes of a row group read one whole row (coalesced)
    const int rg = item / St::CPR, ch = item % St::CPR;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row0 + rg * 4 + j;
      r[it][j] = make_uint4(0, 0, 0, 0);
      if (item < St::TITEMS && row < n_rows) r[it][j] = *reinterpret_cast<const uint4*>(x + (int64_t)row * sl + ch * 8);
    }
  }
}
// X^T[dim][row]: the 16-byte row pairs (8 rows) of a dim are XOR-swizzled by dim / 16 so that the 8-byte transposing writes of a
// half-wave (dims 8 apart: only two bank offsets) spread over all banks; tp_tfrag() un-swizzles on the read side.
template <int D, int NTHR = kTpThreads>
__device__ __forceinline__ void tp_stash_t(uint16_t* __restrict__ Xt, int tid, const uint4 (&r)[TpStage<D, NTHR>::TIT][4]) {
  using St = TpStage<D, NTHR>;
#pragma unroll
  for (int it = 0; it < St::TIT; ++it) {
    const int item = it * NTHR + tid;
    if (item < St::TITEMS) {
      const int rg = item / St::CPR, ch = item % St::CPR;
      const int pair = rg >> 1;
      const int col = (((pair ^ (St::SW * (ch >> 1))) & 7) << 3) + ((rg & 1) << 2);
      const uint32_t w0[4] = {r[it][0].x, r[it][0].y, r[it][0].z, r[it][0].w};
      const uint32_t w1[4] = {r[it][1].x, r[it][1].y, r[it][1].z, r[it][1].w};
      const uint32_t w2[4] = {r[it][2].x, r[it][2].y, r[it][2].z, r[it][2].w};
      const uint32_t w3[4] = {r[it][3].x, r[it][3].y, r[it][3].z, r[it][3].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint2 lo, hi;
        lo.x = (w0[e] & 0xffffu) | (w1[e] << 16);
        lo.y = (w2[e] & 0xffffu) | (w3[e] << 16);
        hi.x = (w0[e] >> 16) | (w1[e] & 0xffff0000u);
        hi.y = (w2[e] >> 16) | (w3[e] & 0xffff0000u);
        *reinterpret_cast<uint2*>(Xt + (ch * 8 + 2 * e) * St::LDT + col) = lo;
        *reinterpret_cast<uint2*>(Xt + (ch * 8 + 2 * e + 1) * St::LDT + col) = hi;
      }
    }
  }
}
// MFMA B fragment of X^T: column dt*16 + lr of the output, rows ks*32 + lg*8 .. +8 of the tile as the contraction index
template <int D>
__device__ __forceinline__ uint4 tp_tfrag(const uint16_t* __restrict__ Xt, int dt, int ks, int lr, int lg) {
  using St = TpStage<D>;
  const int pair = ks * 4 + lg;
  return *reinterpret_cast<const uint4*>(Xt + (dt * 16 + lr) * St::LDT + (((pair ^ (St::SW * dt)) & 7) << 3));
}

// ---- out[b,h,:] = sum_i x[b,h,i,:] * (w ? 1 / w[b,h,i] : 1)   (fp32; the eps/N terms of the forward and of dV) ----
template <typename T>
__global__ __launch_bounds__(256) void tp_colsum_kernel(const void* __restrict__ x_, TpStrides xs, const float* __restrict__ w, float* __restrict__ out,
                                                        int H, int L, int D) {
  const int h = blockIdx.x, b = blockIdx.y;
  const int cpr = D / 8;
  const int tid = threadIdx.x, ch = tid % cpr, r0 = tid / cpr, rstep = 256 / cpr;
  const uint16_t* x = reinterpret_cast<const uint16_t*>(x_) + b * xs.b + h * xs.h;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = r0; i < L; i += rstep) {
    float f[8];
    load16<T>(x + (int64_t)i * xs.l + ch * 8, f);
    const float wi = w ? 1.0f / w[((int64_t)b * H + h) * L + i] : 1.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e] * wi;
  }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; ++e) red[(r0 * cpr + ch) * 8 + e] = acc[e];
  __syncthreads();
  if (tid < D) {
    const int c = tid / 8, e = tid % 8;
    float s = 0.f;
    for (int r = 0; r < rstep; ++r) s += red[(r * cpr + c) * 8 + e];  // fixed order
    out[((int64_t)b * H + h) * D + tid] = s;
  }
}

// ---- delta[b,h,i] = dO_i . o_i ----
template <typename T>
__global__ __launch_bounds__(256) void tp_delta_kernel(const void* __restrict__ o_, const void* __restrict__ do_, TpStrides os, float* __restrict__ delta,
                                                       int H, int L, int D) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x & 63;
  if (i >= L) return;
  const int64_t base = b * os.b + h * os.h + (int64_t)i * os.l;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += load1<T>(o_, base + d) * load1<T>(do_, base + d);
  s = wave_sum(s);
  if (lane == 0) delta[((int64_t)b * H + h) * L + i] = s;
}

// ---- forward ----
// RT row tiles of 16 per wave (a workgroup = 64 * RT query rows): every K / V^T fragment read from LDS feeds RT MFMAs.  With one row
// tile per wave the kernel is LDS-bound (34 KB of fragment reads per 32 MFMAs and wave; 8 waves per CU share 128 B/clk).
// NWF waves per workgroup share one staged K / V^T tile.
template <typename T, int D, bool CAUSAL, int RT, int NWF>
__global__ __launch_bounds__(NWF * 64) void tp_fwd_kernel(const void* __restrict__ q_, const void* __restrict__ k_, const void* __restrict__ v_,
                                                            TpStrides qs, void* __restrict__ o_, TpStrides os, const float* __restrict__ policy,
                                                            const void* __restrict__ bias_, int64_t bias_sb, int64_t bias_sl,
                                                            const float* __restrict__ sumv, float* __restrict__ Mout, float* __restrict__ Dnout, int H, int L,
                                                            float scale, float eps, float c_leak) {
  using S = uint16_t;
  constexpr int NTHR = NWF * 64;
  using St = TpStage<D, NTHR>;
  constexpr int KS = D / 32, DT = D / 16, NT = kTpTile / 16, LDP = kTpTile + kTpPad;
  constexpr int kRows = NWF * 16 * RT;  // query rows per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Ks = reinterpret_cast<S*>(smem);        // [64][LDR]
  S* Vt = Ks + kTpTile * St::LDR;            // [D][LDT] swizzled
  S* Ps = Vt + D * St::LDT;                  // [NWF][RT][16][LDP]
  const int qt = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x, h = blockIdx.y, b = blockIdx.z;  // causal: longest rows first
  const int q0 = qt * kRows;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const S* qb = reinterpret_cast<const S*>(q_) + b * qs.b + h * qs.h;
  const S* kb = reinterpret_cast<const S*>(k_) + b * qs.b + h * qs.h;
  const S* vb = reinterpret_cast<const S*>(v_) + b * qs.b + h * qs.h;
  const float* pol = policy + (int64_t)b * L;
  const S* bias = bias_ ? reinterpret_cast<const S*>(bias_) + b * bias_sb : nullptr;
  const int wrow0 = q0 + w * 16 * RT;  // this wave's first row; row tile rt covers wrow0 + rt*16 .. +16

  uint4 qf[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int qrow = wrow0 + rt * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[rt][ks] = make_uint4(0, 0, 0, 0);
      if (qrow < L) qf[rt][ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * qs.l + ks * 32 + lg * 8);
    }
  }
  tp_f32x4_t acc_o[RT][DT];
  float m[RT][4], l[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int i = 0; i < DT; ++i) acc_o[rt][i] = tp_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m[rt][r] = -INFINITY;
      l[rt][r] = 0.f;
    }
  }
  S* Pw = Ps + w * RT * 16 * LDP;
  const int n_tiles = CAUSAL ? min((L + kTpTile - 1) / kTpTile, (q0 + kRows - 1) / kTpTile + 1) : (L + kTpTile - 1) / kTpTile;
  uint4 kreg[St::RIT], vreg[St::TIT][4];
  tp_fetch_rows<D, NTHR>(kb, qs.l, 0, L, tid, kreg);
  tp_fetch_t<D, NTHR>(vb, qs.l, 0, L, tid, vreg);
  for (int jt = 0; jt < n_tiles; ++jt) {
    const int key0 = jt * kTpTile;
    tp_stash_rows<D, NTHR>(Ks, tid, kreg);
    tp_stash_t<D, NTHR>(Vt, tid, vreg);
    __syncthreads();
    if (jt + 1 < n_tiles) {  // next tile in flight during the MFMAs
      tp_fetch_rows<D, NTHR>(kb, qs.l, key0 + kTpTile, L, tid, kreg);
      tp_fetch_t<D, NTHR>(vb, qs.l, key0 + kTpTile, L, tid, vreg);
    }
    if (!CAUSAL || key0 <= wrow0 + 16 * RT - 1) {  // wave-uniform: tiles wholly in this wave's future are skipped
      float pk[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ki = key0 + nt * 16 + lr;
        pk[nt] = ki < L ? pol[ki] : 0.f;
      }
      tp_f32x4_t acc_s[RT][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc_s[rt][nt] = tp_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (nt * 16 + lr) * St::LDR + ks * 32 + lg * 8);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc_s[rt][nt] = tp_mfma<T>(qf[rt][ks], kf, acc_s[rt][nt]);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = wrow0 + rt * 16 + lg * 4 + r;
          float mx = -INFINITY;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int ki = key0 + nt * 16 + lr;
            float sv = acc_s[rt][nt][r] * scale;
            if (bias && qi < L && ki < L) sv += Elem<T>::to_f(bias[(int64_t)qi * bias_sl + ki]);
            if (ki >= L || (CAUSAL && ki > qi)) sv = -INFINITY;
            acc_s[rt][nt][r] = sv;
            mx = fmaxf(mx, sv);
          }
          mx = row16_max(mx);
          const float mn = fmaxf(m[rt][r], mx);
          const float ms = mn == -INFINITY ? 0.f : mn;
          alpha[r] = __expf(m[rt][r] - ms);
          float rs = 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int ki = key0 + nt * 16 + lr;
            const float p = __expf(acc_s[rt][nt][r] - ms) * (ki == qi ? 1.0f : pk[nt]);
            acc_s[rt][nt][r] = p;
            rs += p;
          }
          rs = row16_sum(rs);
          l[rt][r] = l[rt][r] * alpha[r] + rs;
          m[rt][r] = mn;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_o[rt][dt][r] *= alpha[r];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Pw[(rt * 16 + lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(acc_s[rt][nt][r]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ks = 0; ks < kTpTile / 32; ++ks) {
        uint4 pf[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) pf[rt] = *reinterpret_cast<const uint4*>(Pw + (rt * 16 + lr) * LDP + ks * 32 + lg * 8);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint4 vf = tp_tfrag<D>(Vt, dt, ks, lr, lg);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc_o[rt][dt] = tp_mfma<T>(pf[rt], vf, acc_o[rt][dt]);
        }
      }
    }
    __syncthreads();  // Ks / Vt are rewritten at the top of the next iteration
  }
  S* ob = reinterpret_cast<S*>(o_) + b * os.b + h * os.h;
  const float* sv = sumv + ((int64_t)b * H + h) * D;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = wrow0 + rt * 16 + lg * 4 + r;
      if (qi < L) {
        const float dn = l[rt][r] + eps;
        const float inv = 1.0f / dn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ob[(int64_t)qi * os.l + dt * 16 + lr] = Elem<T>::from_f((acc_o[rt][dt][r] + c_leak * sv[dt * 16 + lr]) * inv);
        if (lr == 0) {
          Mout[((int64_t)b * H + h) * L + qi] = m[rt][r];
          Dnout[((int64_t)b * H + h) * L + qi] = dn;
        }
      }
    }
}

// ---- backward, query side: dQ_i = scale * sum_j dS_ij k_j ----
// NWQ waves of 16 query rows each share the staged K / V / K^T tile (8 waves: two per SIMD, as in the key-side kernel).
template <typename T, int D, bool CAUSAL, int NWQ>
__global__ __launch_bounds__(NWQ * 64) void tp_bwd_dq_kernel(const void* __restrict__ q_, const void* __restrict__ k_, const void* __restrict__ v_,
                                                               TpStrides qs, const void* __restrict__ do_, void* __restrict__ dq_, TpStrides os,
                                                               const float* __restrict__ policy, const void* __restrict__ bias_, int64_t bias_sb,
                                                               int64_t bias_sl, const float* __restrict__ Mx, const float* __restrict__ Dn,
                                                               const float* __restrict__ delta, int H, int L, float scale) {
  using S = uint16_t;
  constexpr int NTHR = NWQ * 64, kRows = NWQ * 16;
  using St = TpStage<D, NTHR>;
  constexpr int KS = D / 32, DT = D / 16, NT = kTpTile / 16, LDP = kTpTile + kTpPad;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Ks = reinterpret_cast<S*>(smem);        // [64][LDR]
  S* Vs = Ks + kTpTile * St::LDR;            // [64][LDR]
  S* Kt = Vs + kTpTile * St::LDR;            // [D][LDT] swizzled
  S* Ps = Kt + D * St::LDT;                  // [NWQ][16][LDP]
  const int qt = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x, h = blockIdx.y, b = blockIdx.z;  // causal: longest rows first
  const int q0 = qt * kRows;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const S* qb = reinterpret_cast<const S*>(q_) + b * qs.b + h * qs.h;
  const S* kb = reinterpret_cast<const S*>(k_) + b * qs.b + h * qs.h;
  const S* vb = reinterpret_cast<const S*>(v_) + b * qs.b + h * qs.h;
  const S* dob = reinterpret_cast<const S*>(do_) + b * os.b + h * os.h;
  const float* pol = policy + (int64_t)b * L;
  const S* bias = bias_ ? reinterpret_cast<const S*>(bias_) + b * bias_sb : nullptr;
  const int64_t st0 = ((int64_t)b * H + h) * L;

  uint4 qf[KS], dof[KS];
  {
    const int qrow = q0 + w * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = dof[ks] = make_uint4(0, 0, 0, 0);
      if (qrow < L) {
        qf[ks] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * qs.l + ks * 32 + lg * 8);
        dof[ks] = *reinterpret_cast<const uint4*>(dob + (int64_t)qrow * os.l + ks * 32 + lg * 8);
      }
    }
  }
  float mrow[4], dinv[4], dlt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + w * 16 + lg * 4 + r;
    mrow[r] = qi < L ? Mx[st0 + qi] : 0.f;
    dinv[r] = qi < L ? 1.0f / Dn[st0 + qi] : 0.f;
    dlt[r] = qi < L ? delta[st0 + qi] : 0.f;
  }
  tp_f32x4_t acc_q[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc_q[i] = tp_f32x4_t{0.f, 0.f, 0.f, 0.f};
  S* Pw = Ps + w * 16 * LDP;
  const int n_tiles = CAUSAL ? min((L + kTpTile - 1) / kTpTile, (q0 + kRows - 1) / kTpTile + 1) : (L + kTpTile - 1) / kTpTile;
  uint4 kreg[St::RIT], vreg[St::RIT], ktreg[St::TIT][4];
  tp_fetch_rows<D, NTHR>(kb, qs.l, 0, L, tid, kreg);
  tp_fetch_rows<D, NTHR>(vb, qs.l, 0, L, tid, vreg);
  tp_fetch_t<D, NTHR>(kb, qs.l, 0, L, tid, ktreg);
  for (int jt = 0; jt < n_tiles; ++jt) {
    const int key0 = jt * kTpTile;
    tp_stash_rows<D, NTHR>(Ks, tid, kreg);
    tp_stash_rows<D, NTHR>(Vs, tid, vreg);
    tp_stash_t<D, NTHR>(Kt, tid, ktreg);
    __syncthreads();
    if (jt + 1 < n_tiles) {
      tp_fetch_rows<D, NTHR>(kb, qs.l, key0 + kTpTile, L, tid, kreg);
      tp_fetch_rows<D, NTHR>(vb, qs.l, key0 + kTpTile, L, tid, vreg);
      tp_fetch_t<D, NTHR>(kb, qs.l, key0 + kTpTile, L, tid, ktreg);
    }
    if (!CAUSAL || key0 <= q0 + w * 16 + 15) {  // wave-uniform: tiles wholly in this wave's future are skipped
    float pk[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ki = key0 + nt * 16 + lr;
      pk[nt] = ki < L ? pol[ki] : 0.f;
    }
    tp_f32x4_t acc_s[NT], acc_a[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc_s[nt] = acc_a[nt] = tp_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (nt * 16 + lr) * St::LDR + ks * 32 + lg * 8);
        const uint4 vf = *reinterpret_cast<const uint4*>(Vs + (nt * 16 + lr) * St::LDR + ks * 32 + lg * 8);
        acc_s[nt] = tp_mfma<T>(qf[ks], kf, acc_s[nt]);
        acc_a[nt] = tp_mfma<T>(dof[ks], vf, acc_a[nt]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + w * 16 + lg * 4 + r;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ki = key0 + nt * 16 + lr;
        float sv = acc_s[nt][r] * scale;
        if (bias && qi < L && ki < L) sv += Elem<T>::to_f(bias[(int64_t)qi * bias_sl + ki]);
        const bool valid = ki < L && qi < L && !(CAUSAL && ki > qi);
        const float e = valid ? __expf(sv - mrow[r]) : 0.f;
        const float pe = e * (ki == qi ? 1.0f : pk[nt]);
        const float dE = (acc_a[nt][r] - dlt[r]) * dinv[r];
        Pw[(lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(dE * pe);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < kTpTile / 32; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(Pw + lr * LDP + ks * 32 + lg * 8);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc_q[dt] = tp_mfma<T>(pf, tp_tfrag<D>(Kt, dt, ks, lr, lg), acc_q[dt]);
    }
    }
    __syncthreads();
  }
  S* dqb = reinterpret_cast<S*>(dq_) + b * os.b + h * os.h;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + w * 16 + lg * 4 + r;
    if (qi < L) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dqb[(int64_t)qi * os.l + dt * 16 + lr] = Elem<T>::from_f(acc_q[dt][r] * scale);
    }
  }
}

// ---- backward, key side: dK_j, dV_j, dpolicy_j (per head) ----
// NWK waves of 16 keys each per workgroup share one staged query tile (Q, dO, Q^T, dO^T: ~73 KB of LDS whatever NWK is): 8 waves put two
// waves on every SIMD where 4 left one (LDS allows a single workgroup per CU).
template <typename T, int D, bool CAUSAL, int NWK>
__global__ __launch_bounds__(NWK * 64) void tp_bwd_dkv_kernel(const void* __restrict__ q_, const void* __restrict__ k_, const void* __restrict__ v_,
                                                                TpStrides qs, const void* __restrict__ do_, void* __restrict__ dk_, void* __restrict__ dv_,
                                                                TpStrides os, const float* __restrict__ policy, const void* __restrict__ bias_,
                                                                int64_t bias_sb, int64_t bias_sl, const float* __restrict__ Mx, const float* __restrict__ Dn,
                                                                const float* __restrict__ delta, const float* __restrict__ gsum,
                                                                float* __restrict__ dpol_heads, int H, int L, float scale, float c_leak) {
  using S = uint16_t;
  constexpr int NTHR = NWK * 64, kKeys = NWK * 16;
  using St = TpStage<D, NTHR>;
  constexpr int KS = D / 32, DT = D / 16, NT = kTpTile / 16, LDP = kTpTile + kTpPad;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S* Qs = reinterpret_cast<S*>(smem);        // [64][LDR]
  S* Os = Qs + kTpTile * St::LDR;            // dO [64][LDR]
  S* Qt = Os + kTpTile * St::LDR;            // [D][LDT] swizzled
  S* Ot = Qt + D * St::LDT;                  // dO^T
  S* Ps = Ot + D * St::LDT;                  // [NWK][2][16][LDP]: A^T and dS^T of each wave
  float* stat = reinterpret_cast<float*>(Ps + NWK * 2 * 16 * LDP);  // [3][64]: m, 1/Dn, delta of the query tile
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int key0 = kt * kKeys;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const S* qb = reinterpret_cast<const S*>(q_) + b * qs.b + h * qs.h;
  const S* kb = reinterpret_cast<const S*>(k_) + b * qs.b + h * qs.h;
  const S* vb = reinterpret_cast<const S*>(v_) + b * qs.b + h * qs.h;
  const S* dob = reinterpret_cast<const S*>(do_) + b * os.b + h * os.h;
  const float* pol = policy + (int64_t)b * L;
  const S* bias = bias_ ? reinterpret_cast<const S*>(bias_) + b * bias_sb : nullptr;
  const int64_t st0 = ((int64_t)b * H + h) * L;

  uint4 kf[KS], vf[KS];  // A operands: this wave's 16 keys
  {
    const int krow = key0 + w * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = vf[ks] = make_uint4(0, 0, 0, 0);
      if (krow < L) {
        kf[ks] = *reinterpret_cast<const uint4*>(kb + (int64_t)krow * qs.l + ks * 32 + lg * 8);
        vf[ks] = *reinterpret_cast<const uint4*>(vb + (int64_t)krow * qs.l + ks * 32 + lg * 8);
      }
    }
  }
  float pkey[4], dpol[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kj = key0 + w * 16 + lg * 4 + r;
    pkey[r] = kj < L ? pol[kj] : 0.f;
    dpol[r] = 0.f;
  }
  tp_f32x4_t acc_k[DT], acc_v[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc_k[i] = acc_v[i] = tp_f32x4_t{0.f, 0.f, 0.f, 0.f};
  S* Aw = Ps + w * 2 * 16 * LDP;
  S* Dw = Aw + 16 * LDP;
  const int n_qt = (L + kTpTile - 1) / kTpTile;
  const int qt0 = CAUSAL ? key0 / kTpTile : 0;  // first query tile that can see this workgroup's first key
  uint4 qreg[St::RIT], oreg[St::RIT], qtreg[St::TIT][4], otreg[St::TIT][4];
  float sreg[3] = {0.f, 0.f, 0.f};
  auto fetch = [&](int q0) {
    tp_fetch_rows<D, NTHR>(qb, qs.l, q0, L, tid, qreg);
    tp_fetch_rows<D, NTHR>(dob, os.l, q0, L, tid, oreg);
    tp_fetch_t<D, NTHR>(qb, qs.l, q0, L, tid, qtreg);
    tp_fetch_t<D, NTHR>(dob, os.l, q0, L, tid, otreg);
    if (tid < 192) {
      const int which = tid >> 6, qi = q0 + (tid & 63);
      float x = 0.f;
      if (qi < L) x = which == 0 ? Mx[st0 + qi] : (which == 1 ? 1.0f / Dn[st0 + qi] : delta[st0 + qi]);
      sreg[0] = x;
    }
  };
  fetch(qt0 * kTpTile);
  for (int qt = qt0; qt < n_qt; ++qt) {
    const int q0 = qt * kTpTile;
    tp_stash_rows<D, NTHR>(Qs, tid, qreg);
    tp_stash_rows<D, NTHR>(Os, tid, oreg);
    tp_stash_t<D, NTHR>(Qt, tid, qtreg);
    tp_stash_t<D, NTHR>(Ot, tid, otreg);
    if (tid < 192) stat[tid] = sreg[0];
    __syncthreads();
    if (qt + 1 < n_qt) fetch(q0 + kTpTile);
    if (!CAUSAL || key0 + w * 16 <= q0 + kTpTile - 1) {  // wave-uniform: a tile wholly before this wave's keys contributes nothing
    // S^T = K Q^T and dA^T = V dO^T: rows = this wave's keys, columns = the tile's query rows
    tp_f32x4_t acc_s[NT], acc_a[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc_s[nt] = acc_a[nt] = tp_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 qfr = *reinterpret_cast<const uint4*>(Qs + (nt * 16 + lr) * St::LDR + ks * 32 + lg * 8);
        const uint4 ofr = *reinterpret_cast<const uint4*>(Os + (nt * 16 + lr) * St::LDR + ks * 32 + lg * 8);
        acc_s[nt] = tp_mfma<T>(kf[ks], qfr, acc_s[nt]);
        acc_a[nt] = tp_mfma<T>(vf[ks], ofr, acc_a[nt]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int qi = q0 + nt * 16 + lr;
      const float mq = stat[nt * 16 + lr], di = stat[64 + nt * 16 + lr], dl = stat[128 + nt * 16 + lr];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = key0 + w * 16 + lg * 4 + r;
        float sv = acc_s[nt][r] * scale;
        if (bias && qi < L && kj < L) sv += Elem<T>::to_f(bias[(int64_t)qi * bias_sl + kj]);
        const bool valid = kj < L && qi < L && !(CAUSAL && kj > qi);
        const float e = valid ? __expf(sv - mq) : 0.f;
        const float pe = e * (kj == qi ? 1.0f : pkey[r]);
        const float dE = (acc_a[nt][r] - dl) * di;
        if (kj != qi) dpol[r] += dE * e;
        Aw[(lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(pe * di);
        Dw[(lg * 4 + r) * LDP + nt * 16 + lr] = Elem<T>::from_f(dE * pe);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < kTpTile / 32; ++ks) {
      const uint4 af = *reinterpret_cast<const uint4*>(Aw + lr * LDP + ks * 32 + lg * 8);
      const uint4 df = *reinterpret_cast<const uint4*>(Dw + lr * LDP + ks * 32 + lg * 8);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        acc_v[dt] = tp_mfma<T>(af, tp_tfrag<D>(Ot, dt, ks, lr, lg), acc_v[dt]);
        acc_k[dt] = tp_mfma<T>(df, tp_tfrag<D>(Qt, dt, ks, lr, lg), acc_k[dt]);
      }
    }
    }
    __syncthreads();
  }
  S* dkb = reinterpret_cast<S*>(dk_) + b * os.b + h * os.h;
  S* dvb = reinterpret_cast<S*>(dv_) + b * os.b + h * os.h;
  const float* gs = gsum + ((int64_t)b * H + h) * D;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kj = key0 + w * 16 + lg * 4 + r;
    const float dp = row16_sum(dpol[r]);
    if (kj < L) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dkb[(int64_t)kj * os.l + dt * 16 + lr] = Elem<T>::from_f(acc_k[dt][r] * scale);
        dvb[(int64_t)kj * os.l + dt * 16 + lr] = Elem<T>::from_f(acc_v[dt][r] + c_leak * gs[dt * 16 + lr]);
      }
      if (lr == 0) dpol_heads[st0 + kj] = dp;
    }
  }
}

// ---- Gumbel hard keep (DML:1868-1876; torch.nn.functional.gumbel_softmax, hard=True), 2 classes ----
// y = (logp + g) / tau; y_soft = softmax(y); ret = onehot(argmax) - y_soft + y_soft (evaluated in the model dtype, as autograd's
// forward does); keep = ret[0] * prev.  Saves y_soft[0] for the backward.
template <typename T>
__global__ __launch_bounds__(256) void tp_gumbel_fwd_kernel(const void* __restrict__ logp_, const void* __restrict__ g_, const void* __restrict__ prev_,
                                                            void* __restrict__ keep_, void* __restrict__ ysoft_, int64_t n, float tau) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float y[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float s = Elem<T>::round(load1<T>(logp_, 2 * i + c) + load1<T>(g_, 2 * i + c));  // tensor + tensor: rounded to the dtype
    y[c] = Elem<T>::round(s / tau);                                                         // tensor / python float: fp32 divide, one rounding
  }
  // softmax over the 2 classes: fp32 internally, one rounding at the end (at::softmax on low-precision tensors)
  const float mx = fmaxf(y[0], y[1]);
  const float e0 = expf(y[0] - mx), e1 = expf(y[1] - mx);
  const float s0 = Elem<T>::round(e0 / (e0 + e1)), s1 = Elem<T>::round(e1 / (e0 + e1));
  const float hard0 = s0 >= s1 ? 1.f : 0.f;  // max() returns the first index on ties
  const float ret0 = Elem<T>::round(Elem<T>::round(hard0 - s0) + s0);
  store1<T>(keep_, i, ret0 * load1<T>(prev_, i));
  store1<T>(ysoft_, 2 * i, s0);
  store1<T>(ysoft_, 2 * i + 1, s1);
}
// dlogp[c] = (y_soft[c] * (dy[c] - sum_c' y_soft[c'] dy[c'])) / tau with dy = (dkeep * prev, 0)   (at::_softmax_backward_data)
template <typename T>
__global__ __launch_bounds__(256) void tp_gumbel_bwd_kernel(const void* __restrict__ dkeep_, const void* __restrict__ prev_, const void* __restrict__ ysoft_,
                                                            void* __restrict__ dlogp_, void* __restrict__ dprev_, int64_t n, float tau) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float dy0 = Elem<T>::round(load1<T>(dkeep_, i) * load1<T>(prev_, i));
  const float s0 = load1<T>(ysoft_, 2 * i), s1 = load1<T>(ysoft_, 2 * i + 1);
  if (dprev_) {  // d(ret0 * prev) / d prev = ret0
    const float ret0 = Elem<T>::round(Elem<T>::round((s0 >= s1 ? 1.f : 0.f) - s0) + s0);
    store1<T>(dprev_, i, load1<T>(dkeep_, i) * ret0);
  }
  const float dot = dy0 * s0;
  const float d0 = Elem<T>::round((dy0 - dot) * s0);
  const float d1 = Elem<T>::round((0.f - dot) * s1);
  store1<T>(dlogp_, 2 * i, Elem<T>::round(d0 / tau));
  store1<T>(dlogp_, 2 * i + 1, Elem<T>::round(d1 / tau));
}

constexpr int kTpFwdRT = 1;  // row tiles per wave in the forward.  2 (128 query rows per workgroup, every K / V fragment feeding two MFMAs) was
                             // measured SLOWER: 473 vs 347 us at L=2048 -- 349 registers leave one wave per SIMD, and latency hiding
                             // matters more here than the LDS fragment traffic
template <typename T, int D, int NW>
static size_t tp_smem_fwd() {
  using St = TpStage<D>;
  return (size_t)(kTpTile * St::LDR + D * St::LDT + NW * kTpFwdRT * 16 * (kTpTile + kTpPad)) * 2;
}
template <typename T, int D, int NW>
static size_t tp_smem_dq() {
  using St = TpStage<D>;
  return (size_t)(2 * kTpTile * St::LDR + D * St::LDT + NW * 16 * (kTpTile + kTpPad)) * 2;
}
template <typename T, int D, int NW>
static size_t tp_smem_dkv() {
  using St = TpStage<D>;
  return (size_t)(2 * kTpTile * St::LDR + 2 * D * St::LDT + NW * 2 * 16 * (kTpTile + kTpPad)) * 2 + 3 * 64 * sizeof(float);
}

template <typename K>
static bool tp_raise_lds(K kfn, size_t smem) {
  if (smem <= 64 * 1024) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
    (void)hipGetLastError();
    set_error("dl_attn_policy: cannot raise the dynamic LDS limit to %zu bytes", smem);
    return false;
  }
  return true;
}

template <typename T, int D, bool CAUSAL>
static int tp_fwd_go(const void* q, const void* k, const void* v, TpStrides qs, void* o, TpStrides os, const float* policy, const void* bias,
                     int64_t bias_sb, int64_t bias_sl, float* sumv, float* M, float* Dn, int B, int H, int L, float scale, float eps, float c_leak,
                     hipStream_t st) {
  hipLaunchKernelGGL((tp_colsum_kernel<T>), dim3((unsigned)H, (unsigned)B), dim3(256), 0, st, v, qs, (const float*)nullptr, sumv, H, L, D);
  auto go = [&](auto nw_tag) -> int {
    constexpr int NW = decltype(nw_tag)::value;
    auto kfn = tp_fwd_kernel<T, D, CAUSAL, kTpFwdRT, NW>;
    const size_t smem = tp_smem_fwd<T, D, NW>();
    if (!tp_raise_lds(kfn, smem)) return DL_ERR_LAUNCH;
    constexpr int kRows = NW * 16 * kTpFwdRT;
    hipLaunchKernelGGL(kfn, dim3((unsigned)((L + kRows - 1) / kRows), (unsigned)H, (unsigned)B), dim3(NW * 64), smem, st, q, k, v, qs, o, os, policy,
                       bias, bias_sb, bias_sl, (const float*)sumv, M, Dn, H, L, scale, eps, c_leak);
    return DL_OK;
  };
  // 8 waves sharing a K / V^T tile measured the same as 4 (358 vs 361 us at L=2048): the forward already runs two workgroups per CU
  return go(std::integral_constant<int, 4>{});
}

template <typename T, int D, bool CAUSAL>
static int tp_bwd_go(const void* q, const void* k, const void* v, TpStrides qs, const void* o, const void* d_o, void* dq, void* dk, void* dv, TpStrides os,
                     const float* policy, const void* bias, int64_t bias_sb, int64_t bias_sl, const float* M, const float* Dn, float* delta, float* gsum,
                     float* dpol_heads, int B, int H, int L, float scale, float c_leak, hipStream_t st) {
  hipLaunchKernelGGL((tp_delta_kernel<T>), dim3((unsigned)((L + 3) / 4), (unsigned)H, (unsigned)B), dim3(256), 0, st, o, d_o, os, delta, H, L, D);
  hipLaunchKernelGGL((tp_colsum_kernel<T>), dim3((unsigned)H, (unsigned)B), dim3(256), 0, st, d_o, os, Dn, gsum, H, L, D);
  // waves per workgroup of the two backward kernels (16 keys / 16 query rows each, sharing one staged tile of the other side): 8 at
  // head_dim 128 once the launch still has >= 256 workgroups (two waves per SIMD where four leave one: L=2048 1160 -> 685 us), else 4
  // (head_dim 64 already fits two 4-wave workgroups per CU; short rows need the workgroup count: L=631 187 vs 229 us)
  const bool wide = D == 128 && (int64_t)B * H * ((L + 127) / 128) >= 256;
  auto go = [&](auto nw_tag) -> int {
    constexpr int NW = decltype(nw_tag)::value;
    const dim3 grid((unsigned)((L + NW * 16 - 1) / (NW * 16)), (unsigned)H, (unsigned)B);
    {
      auto kfn = tp_bwd_dkv_kernel<T, D, CAUSAL, NW>;
      const size_t smem = tp_smem_dkv<T, D, NW>();
      if (!tp_raise_lds(kfn, smem)) return DL_ERR_LAUNCH;
      hipLaunchKernelGGL(kfn, grid, dim3(NW * 64), smem, st, q, k, v, qs, d_o, dk, dv, os, policy, bias, bias_sb, bias_sl, M, Dn, (const float*)delta,
                         (const float*)gsum, dpol_heads, H, L, scale, c_leak);
    }
    {
      auto kfn = tp_bwd_dq_kernel<T, D, CAUSAL, NW>;
      const size_t smem = tp_smem_dq<T, D, NW>();
      if (!tp_raise_lds(kfn, smem)) return DL_ERR_LAUNCH;
      hipLaunchKernelGGL(kfn, grid, dim3(NW * 64), smem, st, q, k, v, qs, d_o, dq, os, policy, bias, bias_sb, bias_sl, M, Dn, (const float*)delta, H, L, scale);
    }
    return DL_OK;
  };
  if (wide) return go(std::integral_constant<int, 8>{});
  return go(std::integral_constant<int, 4>{});
}

}  // namespace dl

using namespace dl;

static bool tp_common_ok(const char* who, int B, int H, int L, int head_dim, int dtype, const int64_t* qkv_strides, const int64_t* o_strides) {
  if (!(B > 0 && H > 0 && L > 0)) {
    set_error("%s: bad shape B=%d H=%d L=%d", who, B, H, L);
    return false;
  }
  if (!(head_dim == 64 || head_dim == 128)) {
    set_error("%s: head_dim=%d unsupported (64 or 128)", who, head_dim);
    return false;
  }
  if (!(dtype == DL_BF16 || dtype == DL_F16)) {
    set_error("%s: bf16 / f16 only (MFMA path)", who);
    return false;
  }
  for (int i = 0; i < 3; ++i)
    if (qkv_strides[i] % 8 || o_strides[i] % 8) {
      set_error("%s: strides must be multiples of 8 elements (16-byte rows)", who);
      return false;
    }
  return true;
}

extern "C" int64_t dl_attn_policy_workspace_floats(int B, int H, int L, int head_dim) {
  // forward: sumv [B,H,d]; backward: delta [B,H,L] + gsum [B,H,d]
  return (int64_t)B * H * (L + 2 * (int64_t)head_dim);
}

extern "C" int dl_attn_policy_fwd(const void* q, const void* k, const void* v, const int64_t* qkv_strides, void* out, const int64_t* o_strides,
                                  const float* policy, const void* bias, int64_t bias_stride_b, int64_t bias_stride_row, float* row_max,
                                  float* row_denom, float* workspace, int B, int H, int L, int head_dim, int causal, float scale, float eps, int n_for_eps,
                                  int dtype, void* stream) {
  DL_REQUIRE(q && k && v && out && policy && row_max && row_denom && workspace && qkv_strides && o_strides, "dl_attn_policy_fwd: NULL pointer");
  if (!tp_common_ok("dl_attn_policy_fwd", B, H, L, head_dim, dtype, qkv_strides, o_strides)) return DL_ERR_ARG;
  DL_REQUIRE(!(causal && bias), "dl_attn_policy_fwd: is_causal and an explicit mask are exclusive (DML:944)");
  DL_REQUIRE(n_for_eps > 0, "dl_attn_policy_fwd: n_for_eps must be the padded key count N of eps / N");
  const TpStrides qs{qkv_strides[0], qkv_strides[1], qkv_strides[2]}, os{o_strides[0], o_strides[1], o_strides[2]};
  hipStream_t st = as_stream(stream);
  const float c_leak = eps / (float)n_for_eps;
  int rc;
#define DL_TP_FWD(TT, DD, CC) rc = tp_fwd_go<TT, DD, CC>(q, k, v, qs, out, os, policy, bias, bias_stride_b, bias_stride_row, workspace, row_max, row_denom, B, H, L, scale, eps, c_leak, st)
#define DL_TP_DISPATCH(MACRO)                                                              \
  if (dtype == DL_BF16) {                                                                  \
    if (head_dim == 128) { if (causal) MACRO(bf16_t, 128, true); else MACRO(bf16_t, 128, false); } \
    else { if (causal) MACRO(bf16_t, 64, true); else MACRO(bf16_t, 64, false); }           \
  } else {                                                                                 \
    if (head_dim == 128) { if (causal) MACRO(f16_t, 128, true); else MACRO(f16_t, 128, false); }   \
    else { if (causal) MACRO(f16_t, 64, true); else MACRO(f16_t, 64, false); }             \
  }
  DL_TP_DISPATCH(DL_TP_FWD)
#undef DL_TP_FWD
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_attn_policy_fwd");
  return DL_OK;
}

extern "C" int dl_attn_policy_bwd(const void* q, const void* k, const void* v, const int64_t* qkv_strides, const void* out, const void* d_out, void* dq,
                                  void* dk, void* dv, const int64_t* o_strides, const float* policy, const void* bias, int64_t bias_stride_b,
                                  int64_t bias_stride_row, const float* row_max, const float* row_denom, float* dpolicy_heads, float* workspace, int B, int H,
                                  int L, int head_dim, int causal, float scale, float eps, int n_for_eps, int dtype, void* stream) {
  DL_REQUIRE(q && k && v && out && d_out && dq && dk && dv && policy && row_max && row_denom && dpolicy_heads && workspace && qkv_strides && o_strides,
             "dl_attn_policy_bwd: NULL pointer");
  if (!tp_common_ok("dl_attn_policy_bwd", B, H, L, head_dim, dtype, qkv_strides, o_strides)) return DL_ERR_ARG;
  DL_REQUIRE(!(causal && bias), "dl_attn_policy_bwd: is_causal and an explicit mask are exclusive (DML:944)");
  DL_REQUIRE(n_for_eps > 0, "dl_attn_policy_bwd: n_for_eps must be the padded key count N of eps / N");
  const TpStrides qs{qkv_strides[0], qkv_strides[1], qkv_strides[2]}, os{o_strides[0], o_strides[1], o_strides[2]};
  hipStream_t st = as_stream(stream);
  const float c_leak = eps / (float)n_for_eps;
  float* delta = workspace + (int64_t)B * H * head_dim;  // after the forward's sumv
  float* gsum = delta + (int64_t)B * H * L;
  int rc;
#define DL_TP_BWD(TT, DD, CC) rc = tp_bwd_go<TT, DD, CC>(q, k, v, qs, out, d_out, dq, dk, dv, os, policy, bias, bias_stride_b, bias_stride_row, row_max, row_denom, delta, gsum, dpolicy_heads, B, H, L, scale, c_leak, st)
  DL_TP_DISPATCH(DL_TP_BWD)
#undef DL_TP_BWD
#undef DL_TP_DISPATCH
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_attn_policy_bwd");
  return DL_OK;
}

extern "C" int dl_gumbel_hard_keep_fwd(const void* log_probs, const void* gumbels, const void* prev_decision, void* keep, void* y_soft, int64_t n, float tau,
                                       int dtype, void* stream) {
  DL_REQUIRE(log_probs && gumbels && prev_decision && keep && y_soft, "dl_gumbel_hard_keep_fwd: NULL pointer");
  DL_REQUIRE(n >= 0 && tau > 0.f, "dl_gumbel_hard_keep_fwd: bad n / tau");
  if (n == 0) return DL_OK;
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == DL_BF16) hipLaunchKernelGGL((tp_gumbel_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, log_probs, gumbels, prev_decision, keep, y_soft, n, tau);
  else if (dtype == DL_F16) hipLaunchKernelGGL((tp_gumbel_fwd_kernel<f16_t>), grid, dim3(256), 0, st, log_probs, gumbels, prev_decision, keep, y_soft, n, tau);
  else if (dtype == DL_F32) hipLaunchKernelGGL((tp_gumbel_fwd_kernel<f32_t>), grid, dim3(256), 0, st, log_probs, gumbels, prev_decision, keep, y_soft, n, tau);
  else {
    set_error("dl_gumbel_hard_keep_fwd: unsupported dtype %d", dtype);
    return DL_ERR_ARG;
  }
  DL_CHECK_LAUNCH("dl_gumbel_hard_keep_fwd");
  return DL_OK;
}

extern "C" int dl_gumbel_hard_keep_bwd(const void* d_keep, const void* prev_decision, const void* y_soft, void* d_log_probs, void* d_prev, int64_t n, float tau,
                                       int dtype, void* stream) {
  DL_REQUIRE(d_keep && prev_decision && y_soft && d_log_probs, "dl_gumbel_hard_keep_bwd: NULL pointer");
  DL_REQUIRE(n >= 0 && tau > 0.f, "dl_gumbel_hard_keep_bwd: bad n / tau");
  if (n == 0) return DL_OK;
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == DL_BF16) hipLaunchKernelGGL((tp_gumbel_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, d_keep, prev_decision, y_soft, d_log_probs, d_prev, n, tau);
  else if (dtype == DL_F16) hipLaunchKernelGGL((tp_gumbel_bwd_kernel<f16_t>), grid, dim3(256), 0, st, d_keep, prev_decision, y_soft, d_log_probs, d_prev, n, tau);
  else if (dtype == DL_F32) hipLaunchKernelGGL((tp_gumbel_bwd_kernel<f32_t>), grid, dim3(256), 0, st, d_keep, prev_decision, y_soft, d_log_probs, d_prev, n, tau);
  else {
    set_error("dl_gumbel_hard_keep_bwd: unsupported dtype %d", dtype);
    return DL_ERR_ARG;
  }
  DL_CHECK_LAUNCH("dl_gumbel_hard_keep_bwd");
  return DL_OK;
}
