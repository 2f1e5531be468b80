// One decode step (batch 1, 16-bit dtypes) as ONE persistent launch: replaces the ~6 launches per layer of the launch path
// (dl_gemv x4 + dl_attn_decode_rope [+ combine]) -- DML:1011-1013, 1127, 328, 2709 (GEMMs), DML:134-139 / 1289 / 1295 (norm +
// residual), DML:260-285 + CU:109-268 + DML:1114-1122 (RoPE, KV append, attention) -- with the SAME arithmetic in the same order
// (gemv_dot.h / attn_decode_body.h are shared), so its logits are bit-identical to the launch path's.
//
// Why: at batch 1 every launch streams 33-180 MB and pays ~1.5 us of boundary + 3-5 us until its first cold bytes arrive; 197
// launches cost ~0.55 ms of a 2.65 ms step.  Here the weight stream never stops: a workgroup per CU walks a phase table
// (q|k|v GEMV -> attention -> o GEMV -> gate|up GEMV -> down GEMV, per layer, then lm_head); each streaming wave keeps two 16 KiB
// batches of weight rows in flight ACROSS phase boundaries (the next phase's first rows are requested before the current phase's
// results have been exchanged), and phase outputs travel between workgroups as 8-byte {tag, value} granules written with one
// agent-scope store and swept by dedicated poller waves (no fences: a granule is its own flag; MI355X_MICROARCH.md price list,
// rows handoff-1to1 / allgather).  Per phase an XCD-sharded arrival counter is only a HINT that tells the pollers when a sweep is
// worth issuing (granules validate themselves by tag), so polling traffic stays off the fabric while producers stream.
//
// Roles inside a 512-thread workgroup: waves 0..5 stream weight rows (row pairs dealt round-robin over all streaming waves of the
// grid), waves 6..7 gather the next phase's input vector into LDS; all 8 waves run the attention body (two 4-wave splits side by
// side).  The residual stream h lives in LDS in every workgroup (each applies the same updates), as does x = rmsnorm(h) * w.
//
// Safety: every spin is bounded (abort word + give-up count), the kernel never blocks on a workgroup that is not resident as long
// as grid <= resident capacity (1 workgroup per CU requested; the host passes the CU count).
#include <mutex>

#include "attn_decode_body.h"
#include "gemv_dot.h"
#include "granule.h"

namespace dl {

constexpr int kPW = 8;           // waves per workgroup
constexpr int kPT = kPW * 64;    // threads
constexpr int kPS = 6;           // streaming waves (0..5)
constexpr int kPP = kPW - kPS;   // poller waves (6..7)
constexpr int kPU = 8;           // 16-byte chunks per row per batch
constexpr int kPD = 128;         // head_dim
constexpr int kPartGr = kPD + 2; // granules of one split partial: M, L, O[D]
constexpr int kCtrlGr = 16;      // control granules at the start of the sync buffer (word 0: abort)
constexpr int kGU = 44;          // granules per lane per sweep group (pollers: <= 6144 granules in one round trip)

// Device-side view of DlDecodePhase (same layout): the pointer fields are typed as GLOBAL pointers.  Pointers loaded from memory
// are generic to the compiler (flat_load: 64-bit VGPR addresses, both wait counters); typed this way it emits global_load with
// scalar bases, also inside the inlined attention body.
#define DL_GLOBAL __attribute__((address_space(1)))
typedef DL_GLOBAL const uint16_t* gc16_t;
typedef DL_GLOBAL uint16_t* g16_t;
struct PhaseDev {
  int32_t kind, flags;
  int32_t N, K;
  int64_t in_region, out_region;
  int32_t in_expect, n_splits, len_group, reserved;
  gc16_t W;
  gc16_t norm_w;
  g16_t out;
  g16_t dump;
  g16_t k_slab;
  g16_t v_slab;
};
static_assert(sizeof(PhaseDev) == sizeof(DlDecodePhase), "PhaseDev must mirror DlDecodePhase");

// plain 16-byte global load / store through builtin vector types (HIP's uint4 struct cannot live behind an address-space pointer)
typedef uint32_t pu32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_g16(gc16_t p) {
  const pu32x4_t r = *(DL_GLOBAL const pu32x4_t*)p;
  return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void st_g16(g16_t p, const uint4& v) {
  pu32x4_t r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  *(DL_GLOBAL pu32x4_t*)p = r;
}

// 16-byte non-temporal load at (uniform row pointer + 32-bit per-lane byte offset): global_load_dwordx4 v, v_off, s[base] offset:imm --
// one offset VGPR per row instead of a 64-bit address pair per load
__device__ __forceinline__ uint4 ldg_nt_off(gc16_t row, uint32_t byte_off) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = __builtin_nontemporal_load((DL_GLOBAL const u32x4_t*)((DL_GLOBAL const char*)row + byte_off));
  return make_uint4(r.x, r.y, r.z, r.w);
}

struct PParams {
  const PhaseDev* phases;
  int n_phases;
  u64_t* sync;            // [kCtrlGr control][n_phases * 8 arrival counters as u32 pairs][granule regions]
  int64_t r_cnt, r_qkv, r_part, r_attn, r_o, r_act, r_dn;  // region offsets (granules)
  int H, Hpad, Kpad, n_heads, n_kv_heads, max_splits;
  float eps, scale;
  const void* cos_tab;
  const void* sin_tab;
  int n_pos;
  const int32_t* pos_base;   // [1] position of the new token (= un-evicted length)
  const int32_t* kv_len0;    // [1] layers < sparse_layer
  const int32_t* kv_len1;    // [1] layers >= sparse_layer
  const int64_t* cur_ids;    // [1]
  int64_t slab_stride_h;
  int T_cap;
  int spin_limit;
  long long* stamps;         // debug: [n_phases][8] wall-clock stamps of workgroup `stamp_wg` (NULL in production)
  int stamp_wg;
};

// ---------------------------------------------------------------------------------------------------------------------------
// arrival hint + bounded polling
// ---------------------------------------------------------------------------------------------------------------------------
struct Poll {
  u64_t* sync;
  int spin_limit;
  bool dead;
};

__device__ __forceinline__ bool poll_aborted(Poll& pl) {
  if (!pl.dead && ctr_load(reinterpret_cast<const uint32_t*>(pl.sync)) != 0u) pl.dead = true;
  return pl.dead;
}
__device__ __forceinline__ void poll_give_up(Poll& pl, uint32_t code) {
  pl.dead = true;
  __hip_atomic_store((gu32_t*)(pl.sync), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sweep `n_gr` granules of `region` (tag `tag`) into LDS words dst[i] = payload(granule i); the `n_thr` calling threads (wave
// multiples; `t` = index of this thread among them) share the range.  Re-polls only what is still missing.
// There is no separate "ready" flag (a counter polled by 512 waves delayed its own increments by ~20 us): readiness is judged on a
// sample of the granules themselves (wait_sample), then this sweep runs; `gap` selects the s_sleep between failed passes.
template <int GU>
__device__ __forceinline__ void sweep(Poll& pl, const u64_t* region, int n_gr, uint32_t tag, uint32_t* dst, int t, int n_thr, uint32_t code,
                                      int gap) {
  static_assert(GU <= 64, "one bit per granule slot");
  for (int base = 0; base < n_gr; base += n_thr * GU) {
    u64_t v[GU];
    u64_t have = 0;  // bit k: slot k holds a granule with the right tag (per lane)
    for (int spins = 0;; ++spins) {
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int idx = base + k * n_thr + t;
        if (idx < n_gr && !((have >> k) & 1)) v[k] = gr_load(region + idx);
      }
      bool ok = true;
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int idx = base + k * n_thr + t;
        if (idx < n_gr && !((have >> k) & 1)) {
          const bool h = (uint32_t)(v[k] >> 32) == tag;
          have |= (u64_t)h << k;
          ok &= h;
        }
      }
      if (__all(ok) || pl.dead) break;
      if ((spins & 63) == 63 && poll_aborted(pl)) break;
      if (spins > pl.spin_limit) {
        poll_give_up(pl, code);
        break;
      }
      if (gap > 32) __builtin_amdgcn_s_sleep(64);
      else if (gap > 8) __builtin_amdgcn_s_sleep(24);
      else __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int k = 0; k < GU; ++k) {
      const int idx = base + k * n_thr + t;
      if (idx < n_gr) dst[idx] = (uint32_t)v[k];
    }
  }
}

// Cheap readiness poll: ONE 8-byte load per lane on a SAMPLE of the region -- `count` granules, `stride` apart, ending at the last
// granule (for a GEMV phase: spread over the units of the last round, which finish last; for attention: one per head).  512 bytes
// per pass instead of the 16-44 KB of a full sweep, so it can run for the whole producer phase without loading the fabric; the full
// sweep that follows still validates every granule.
__device__ __forceinline__ void wait_sample(Poll& pl, const u64_t* region, int n_gr, int count, int stride, uint32_t tag, int lane, uint32_t code) {
  if (pl.dead || n_gr <= 0) return;
  int idx = n_gr - 1 - (lane % count) * stride;
  idx = idx < 0 ? 0 : idx;
  for (int spins = 0;; ++spins) {
    const u64_t v = gr_load(region + idx);
    if (__all((uint32_t)(v >> 32) == tag)) return;
    if ((spins & 63) == 63 && poll_aborted(pl)) return;
    if (spins > pl.spin_limit) {
      poll_give_up(pl, code);
      return;
    }
    __builtin_amdgcn_s_sleep(6);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight-row batches: (row pair, 8 chunk slots).  Per lane the chunks of a row are visited in ascending order (lane + 64 c), exactly
// as dl_gemv does, so the fp32 partial sums are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------------
struct Batch {
  gc16_t w0;            // row pointers of the pair
  gc16_t w1;
  int ph;               // phase this batch belongs to (-1: none)
  int u, sp, bi;        // unit, sub-pair inside the unit, batch inside the sub-pair
};

struct GemvGeom {
  gc16_t W;
  int N, K, nvec, n_out, n_units, nsp, nb;
  bool pair;
};

__device__ __forceinline__ GemvGeom gemv_geom(const PhaseDev& d) {
  GemvGeom g;
  g.W = d.W;
  g.N = d.N;
  g.K = d.K;
  g.nvec = d.K / 8;
  g.pair = (d.flags & DL_PHASE_OUT_SILU_PAIR) != 0;
  g.n_out = g.pair ? d.N / 2 : d.N;
  g.n_units = (g.n_out + 1) / 2;
  g.nsp = g.pair ? 2 : 1;
  g.nb = ((g.nvec + 63) / 64 + kPU - 1) / kPU;
  return g;
}

__device__ __forceinline__ void batch_rows(const GemvGeom& g, Batch& b) {
  int r0, r1;
  if (g.pair) {
    int o = 2 * b.u + b.sp;
    o = o < g.n_out ? o : g.n_out - 1;
    r0 = o;
    r1 = g.n_out + o;
  } else {
    r0 = 2 * b.u;
    r1 = 2 * b.u + 1 < g.N ? 2 * b.u + 1 : g.N - 1;
  }
  b.w0 = g.W + (int64_t)r0 * g.K;
  b.w1 = g.W + (int64_t)r1 * g.K;
}

// Always issues 2 x kPU loads (so that the compiler can count them: a conditional issue makes it drain vmcnt to 0 before the
// previous batch is consumed).  An invalid batch reads one 16-byte line of the weight matrix over and over (L2 hit, no traffic).
__device__ __forceinline__ void issue(const Batch& b, bool valid, gc16_t dummy, int nvec, int lane, uint4 (&buf)[2][kPU]) {
  gc16_t w0 = valid ? b.w0 : dummy;
  gc16_t w1 = valid ? b.w1 : dummy;
  const bool full = 64 * (b.bi + 1) * kPU <= nvec;  // wave-uniform
  if (valid && full) {
    const uint32_t off = (uint32_t)(lane + 64 * b.bi * kPU) * 16u;
#pragma unroll
    for (int j = 0; j < kPU; ++j) {
      buf[0][j] = ldg_nt_off(w0, off + (uint32_t)j * 1024u);
      buf[1][j] = ldg_nt_off(w1, off + (uint32_t)j * 1024u);
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPU; ++j) {
      int v = lane + 64 * (b.bi * kPU + j);
      v = v < nvec ? v : nvec - 1;  // clamped: a lane past the row end re-reads a valid chunk and ignores it
      v = valid ? v : 0;
      buf[0][j] = ldg_nt_off(w0, (uint32_t)v * 16u);
      buf[1][j] = ldg_nt_off(w1, (uint32_t)v * 16u);
    }
  }
}

template <typename T>
__device__ __forceinline__ void consume(const Batch& b, int nvec, int lane, const uint16_t* xs, const uint4 (&buf)[2][kPU], float& acc0, float& acc1) {
  const int v0 = lane + 64 * b.bi * kPU;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(xs) + (uint32_t)v0 * 16u;
  if (64 * (b.bi + 1) * kPU <= nvec) {  // full batch (wave-uniform): no per-chunk masking
#pragma unroll
    for (int j = 0; j < kPU; ++j) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xb + j * 1024);
      acc0 = dot16<T>(buf[0][j], xv, acc0);
      acc1 = dot16<T>(buf[1][j], xv, acc1);
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPU; ++j) {
      if (v0 + 64 * j < nvec) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xb + j * 1024);
        acc0 = dot16<T>(buf[0][j], xv, acc0);
        acc1 = dot16<T>(buf[1][j], xv, acc1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the kernel.  Roles are split at the TOP level (two phase loops, same number of workgroup barriers per phase) so that each role gets
// its own register allocation: the streamers' 128 VGPRs of in-flight weight batches never coexist with the pollers' sweep / norm state.
// ---------------------------------------------------------------------------------------------------------------------------
// one descriptor from the LDS copy of the table, every word made provably wave-uniform (SGPRs: scalar branches, scalar row pointers)
__device__ __forceinline__ PhaseDev load_phase(const PhaseDev* tab, int ph) {
  constexpr int NW = (int)(sizeof(PhaseDev) / 4);
  const uint32_t* w = reinterpret_cast<const uint32_t*>(tab + ph);
  uint32_t r[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) r[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);
  PhaseDev d;
  __builtin_memcpy(&d, r, sizeof(PhaseDev));
  return d;
}

#define DL_PSTAMP(ph, slot)                                                                                  \
  do {                                                                                                     \
    if (p.stamps && lane == 0) p.stamps[((int64_t)blockIdx.x * p.n_phases + (ph)) * 8 + (slot)] = wall_clock64();       \
  } while (0)

struct Lds {
  uint16_t* h;        // residual stream [Hpad]   (owned by poller wave 0)
  uint16_t* dbuf;     // gathered delta [Hpad]
  uint16_t* x[2];     // x of the current / next GEMV phase [Kpad]
  float* sm_att;      // 2 x (m[16], l[16], o[16][128])
  uint32_t* qkvraw;   // q | k | v of this head, 3 x 64 words
  float* comb;        // [max_splits][D + kAttnPartPad]
  int* cnt;           // spare words
  float* red;         // [4] partial sums of squares of the norm
  const PhaseDev* tab; // the phase table, copied to LDS at kernel start (a descriptor read from global memory is a dependent ~2 us round trip)
};

// RoPE + KV append + split-KV attention of one layer; items (head, split pair) -> workgroups [0, n_heads * nv).  Run by ALL 8 waves of
// the workgroups involved (three __syncthreads() for attention workgroups, two more for combiners: uniform per workgroup).
template <typename T>
__device__ __forceinline__ void attn_phase(const PParams& p, const PhaseDev& d, int ph, Poll& pl, const Lds& L, int tid, int lane, int wid,
                                           int T_old, int pos) {
  constexpr int D = kPD;
  constexpr int ANG = 16;  // lane groups of one 4-wave attention split
  const uint32_t tag = (uint32_t)ph + 1u;
  const int ns = d.n_splits;
  const int nv = (ns + 1) / 2;
  const int item = blockIdx.x;
  const bool is_attn = item < p.n_heads * nv;
  const int n_rep = p.n_heads / p.n_kv_heads;
  if (is_attn) {
    const int hh = item / nv;
    const int kvh = hh / n_rep;
    const int vw = wid >> 2;                    // virtual 4-wave workgroup
    const int vtid = tid & 255;
    const int split = (item % nv) * 2 + vw;     // may be == ns for the odd tail: an empty key range
    AttnSplitState<T, D, 4, 4> st;
    attn_split_issue<T, D, 4, true, 4>(st, vtid, (const void*)d.k_slab, (const void*)d.v_slab, 0, p.slab_stride_h, T_old, 1, 0, kvh, split, ns,
                                       p.T_cap, 0);
    // q | k | v of this head from the q|k|v phase's granules (pair u = elements 2u, 2u+1)
    if (wid == 6) DL_PSTAMP(ph, 0);
    if (wid == 6) {  // small (192 granules per workgroup): the sweep itself is the poll
      sweep<4>(pl, p.sync + p.r_qkv + (int64_t)hh * (D / 2), D / 2, tag - 1, L.qkvraw, lane, 64, 0x20000u | (uint32_t)ph, 16);
    } else if (wid == 7) {
      sweep<4>(pl, p.sync + p.r_qkv + (int64_t)(p.n_heads + kvh) * (D / 2), D / 2, tag - 1, L.qkvraw + D / 2, lane, 64, 0x20001u | (uint32_t)ph, 16);
      sweep<4>(pl, p.sync + p.r_qkv + (int64_t)(p.n_heads + p.n_kv_heads + kvh) * (D / 2), D / 2, tag - 1, L.qkvraw + D, lane, 64,
               0x20002u | (uint32_t)ph, 16);
    }
    if (wid == 6) DL_PSTAMP(ph, 1);
    __syncthreads();
    float M, Lsum, O;
    float* sa = L.sm_att + vw * (2 * ANG + ANG * D);
    attn_split_finish<T, D, 4, true, 4>(st, vtid, L.qkvraw, L.qkvraw + D / 2, L.qkvraw + D, p.cos_tab, p.sin_tab, p.n_pos, pos, p.scale,
                                        hh % n_rep == 0, p.T_cap, sa, sa + ANG, sa + 2 * ANG, M, Lsum, O);
    if (wid == 6) DL_PSTAMP(ph, 2);
    if (ns == 1) {
      // no merge needed: normalise, round, publish element pairs straight into the attention-output region
      if (vw == 0 && vtid < D) {
        const uint32_t mine = Elem<T>::from_f(Lsum > 0.f ? O / Lsum : 0.f);
        const uint32_t up = __shfl_down(mine, 1, 64);
        if ((vtid & 1) == 0) gr_store(p.sync + p.r_attn + (int64_t)hh * (D / 2) + vtid / 2, tag, mine | (up << 16));
      }
    } else if (split < ns && vtid < D) {
      u64_t* pr = p.sync + p.r_part + ((int64_t)hh * ns + split) * kPartGr;
      gr_store(pr + 2 + vtid, tag, __float_as_uint(O));
      if (vtid == 0) {
        gr_store(pr, tag, __float_as_uint(M));
        gr_store(pr + 1, tag, __float_as_uint(Lsum));
      }
    }
    if (wid == 6) DL_PSTAMP(ph, 3);
    __syncthreads();  // qkvraw / sm_att are free again (a combiner role of this workgroup, or the next layer, reuses LDS)
  }
  if (ns > 1 && item < p.n_heads) {
    // ---- combiner of head `item`: the ns partials, merged in split order exactly as attn_decode_combine_kernel does ----
    const u64_t* pr = p.sync + p.r_part + (int64_t)item * ns * kPartGr;
    for (int i = tid; i < ns * kPartGr; i += kPT) {  // every thread polls its own granules (small: <= 32 x 130)
      u64_t v = 0;
      for (int spins = 0;; ++spins) {
        v = gr_load(pr + i);
        if ((uint32_t)(v >> 32) == tag || pl.dead) break;
        if ((spins & 63) == 63 && poll_aborted(pl)) break;
        if (spins > pl.spin_limit) {
          poll_give_up(pl, 0x30000u | (uint32_t)ph);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const int s_ = i / kPartGr, e = i % kPartGr;
      L.comb[s_ * (D + kAttnPartPad) + (e < 2 ? e : e + 2)] = __uint_as_float((uint32_t)v);
    }
    if (wid == 6) DL_PSTAMP(ph, 4);
    __syncthreads();
    if (tid < D) {
      float o1[1];
      attn_split_merge<1>(L.comb, ns, D, tid, o1);
      const uint32_t mine = Elem<T>::from_f(o1[0]);
      const uint32_t up = __shfl_down(mine, 1, 64);
      if ((tid & 1) == 0) gr_store(p.sync + p.r_attn + (int64_t)item * (D / 2) + tid / 2, tag, mine | (up << 16));
    }
    if (wid == 6) DL_PSTAMP(ph, 5);
    __syncthreads();
  }
}

// ---- streaming waves: weights only ----
template <typename T>
__device__ __forceinline__ void streamer_loop(const PParams& p, const Lds& L, Poll& pl, int T0, int T1, int pos) {
  using S = uint16_t;
  const int G = gridDim.x;
  int xsel = 0;
  for (int ph = 0; ph < p.n_phases; ++ph) {
    // the thread index is re-materialised in every phase: otherwise the compiler hoists per-lane addresses / masks of every phase kind
    // out of this loop and keeps them alive across it
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: batch descriptors / row pointers stay in SGPRs
    const PhaseDev d = load_phase(L.tab, ph);
    const uint32_t tag = (uint32_t)ph + 1u;
    constexpr int V = 8;
    constexpr int MAXC = 4;  // residual-stream chunks per thread of waves 0..3 (thread t owns chunks t + 256 c: dl_gemv's ADDNORM map)
    const int nvh = p.H / V;
    if (d.kind == DL_PHASE_EMBED) {  // h = embed[cur_id]: every thread of waves 0..3 loads the chunks it owns
      if (wid < 4) {
        const int64_t id = p.cur_ids[0];
        gc16_t row = d.W + id * (int64_t)p.H;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * 256;
          if (v < nvh) *reinterpret_cast<uint4*>(L.h + v * V) = ld_g16(row + (int64_t)v * V);
        }
      }
      continue;
    }
    if (d.kind == DL_PHASE_ATTN) {
      attn_phase<T>(p, d, ph, pl, L, tid, lane, wid, d.len_group == 0 ? T0 : T1, pos);
      continue;
    }
    const GemvGeom g = gemv_geom(d);
    S* xs = L.x[xsel];
    xsel ^= 1;
    const bool addnorm = (d.flags & DL_PHASE_ADDNORM) != 0;
    const bool has_delta = (d.flags & DL_PHASE_HAS_DELTA) != 0;
    if (wid == 0) DL_PSTAMP(ph, 0);
    // ---- the first two batches of this phase are requested BEFORE the input vector is waited for: the weight stream keeps running
    // through the exchange of the previous phase's results ----
    uint4 A[2][kPU], B[2][kPU];
    Batch b0, b1;
    b0.w0 = b0.w1 = b1.w0 = b1.w1 = g.W;
    b0.ph = b1.ph = ph;
    b0.u = b1.u = b0.sp = b1.sp = b0.bi = b1.bi = 0;
    // what follows the batch `b` inside this phase (false: the wave's work in this phase ends with `b`)
    auto next_batch = [&](const Batch& b, Batch& n) -> bool {
      n = b;
      if (b.bi + 1 < g.nb) {
        n.bi = b.bi + 1;
        return true;
      }
      n.bi = 0;
      if (b.sp + 1 < g.nsp) {
        n.sp = b.sp + 1;
        batch_rows(g, n);
        return true;
      }
      n.sp = 0;
      if (b.u + kPS * G < g.n_units) {  // units are dealt to WORKGROUPS round-robin (u = k G + workgroup), wave = k mod kPS: the bytes
        n.u = b.u + kPS * G;            // per CU are what bounds a phase (a CU streams ~25 GB/s), so CUs must get equal shares
        batch_rows(g, n);
        return true;
      }
      return false;
    };
    b0.u = wid * G + (int)blockIdx.x;
    bool v0 = b0.u < g.n_units;
    if (v0) batch_rows(g, b0);
    bool v1 = v0 && next_batch(b0, b1);
    issue(b0, v0, g.W, g.nvec, lane, A);
    issue(b1, v1, g.W, g.nvec, lane, B);
    if (wid == 0) DL_PSTAMP(ph, 1);
    if (addnorm) {
      // h += delta (rounded), x = w * round(h * rstd): waves 0..3 replay dl_gemv's ADDNORM prologue (same chunk -> thread map, same
      // summation order: bit-identical), while their weight batches are in flight.  The pollers gathered delta into dbuf.
      uint4 wr[MAXC];
      float a[MAXC][V];
      if (wid < 4) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * 256;
          if (v < nvh) wr[c] = ld_g16(d.norm_w + v * V);
        }
      }
      __syncthreads();  // delta complete
      if (wid == 0) DL_PSTAMP(ph, 5);
      float ss = 0.f;
      if (wid < 4) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * 256;
          if (v < nvh) {
            unpack16<T>(*reinterpret_cast<const uint4*>(L.h + v * V), a[c]);
            if (has_delta) {
              float dd[V];
              unpack16<T>(*reinterpret_cast<const uint4*>(L.dbuf + v * V), dd);
#pragma unroll
              for (int e = 0; e < V; ++e) a[c][e] = Elem<T>::round(a[c][e] + dd[e]);
              store16<T>(L.h + v * V, a[c]);
            }
            if (d.dump && blockIdx.x == 0) st_g16(d.dump + v * V, pack16<T>(a[c]));
#pragma unroll
            for (int e = 0; e < V; ++e) ss += a[c][e] * a[c][e];
          }
        }
        ss = wave_sum(ss);
        if (lane == 0) L.red[wid] = ss;
      }
      __syncthreads();
      if (wid < 4) {
        float tsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) tsum += L.red[i];
        const float rstd = rsqrtf(tsum / (float)p.H + p.eps);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int v = tid + c * 256;
          if (v < nvh) {
            float w[V];
            unpack16<T>(wr[c], w);
#pragma unroll
            for (int e = 0; e < V; ++e) a[c][e] = w[e] * Elem<T>::round(a[c][e] * rstd);
            store16<T>(xs + v * V, a[c]);
          }
        }
      }
    }
    __syncthreads();  // x of this phase is in LDS
    if (wid == 0) DL_PSTAMP(ph, 2);

    float acc0 = 0.f, acc1 = 0.f;
    uint32_t act_lo = 0;
    const bool out_global = (d.flags & DL_PHASE_OUT_GLOBAL) != 0;
    // finish one batch: on the last batch of a sub-pair reduce + epilogue + publish
    auto finish = [&](const Batch& b) {
      if (b.bi + 1 < g.nb) return;
      const float s0 = wave_sum(acc0), s1 = wave_sum(acc1);
      acc0 = acc1 = 0.f;
      if (g.pair) {
        const float gg = Elem<T>::round(s0), uu = Elem<T>::round(s1);
        const uint32_t act = Elem<T>::from_f(Elem<T>::round(gg / (1.0f + expf(-gg))) * uu);
        if (b.sp == 0) {
          act_lo = act;
        } else if (lane == 0) {
          gr_store(p.sync + d.out_region + b.u, tag, act_lo | (act << 16));
        }
      } else {
        const uint32_t lo = Elem<T>::from_f(s0), hi = Elem<T>::from_f(s1);
        if (lane == 0) {
          if (out_global) {
            g16_t y = d.out;
            y[2 * b.u] = (S)lo;
            if (2 * b.u + 1 < g.N) y[2 * b.u + 1] = (S)hi;
          } else {
            gr_store(p.sync + d.out_region + b.u, tag, lo | (hi << 16));
          }
        }
      }
    };
    while (v0) {
      consume<T>(b0, g.nvec, lane, xs, A, acc0, acc1);
      finish(b0);
      Batch n;
      const bool vn = v1 && next_batch(b1, n);
      issue(n, vn, g.W, g.nvec, lane, A);
      if (!v1) break;
      consume<T>(b1, g.nvec, lane, xs, B, acc0, acc1);
      finish(b1);
      b0 = n;
      v0 = vn;
      v1 = v0 && next_batch(b0, b1);
      issue(b1, v1, g.W, g.nvec, lane, B);
    }
    if (wid == 0) DL_PSTAMP(ph, 3);
  }
}

// ---- poller waves: gather the next phase's input vector into LDS ----
template <typename T>
__device__ __forceinline__ void poller_loop(const PParams& p, const Lds& L, Poll& pl, int T0, int T1, int pos) {
  using S = uint16_t;
  int xsel = 0;
  for (int ph = 0; ph < p.n_phases; ++ph) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wid - kPS;  // poller index 0..kPP-1
    const PhaseDev d = load_phase(L.tab, ph);
    if (d.kind == DL_PHASE_EMBED) continue;
    if (d.kind == DL_PHASE_ATTN) {
      attn_phase<T>(p, d, ph, pl, L, tid, lane, wid, d.len_group == 0 ? T0 : T1, pos);
      continue;
    }
    const bool addnorm = (d.flags & DL_PHASE_ADDNORM) != 0;
    const bool has_delta = (d.flags & DL_PHASE_HAS_DELTA) != 0;
    S* xs = L.x[xsel];
    xsel ^= 1;
    if (pw == 0) DL_PSTAMP(ph, 4);
    if (has_delta || !addnorm) {
      const int in_gr = addnorm ? p.H / 2 : d.K / 2;
      const int prod = ph - 1;  // producer phase of the input region
      const uint32_t ptag = (uint32_t)prod + 1u;
      // readiness sample: after an attention phase one granule per head (in_expect = n_heads publishers of 64 granules each), after
      // a GEMV phase 64 granules spread over the units of its last round
      int count, stride;
      if (d.in_expect > 0 && d.in_expect < 64) {
        count = d.in_expect;
        stride = in_gr / d.in_expect;
      } else {
        const int G = gridDim.x;
        int last_round = in_gr - (in_gr - 1) / G * G;  // units of the last dealing round (k = kmax) ...
        if (in_gr > G) last_round += G;                // ... and the one before it (other waves of the same workgroups)
        count = last_round < 64 ? last_round : 64;
        stride = last_round / count;
      }
      wait_sample(pl, p.sync + d.in_region, in_gr, count, stride, ptag, lane, 0x50000u | (uint32_t)ph);
      if (pw == 0) DL_PSTAMP(ph, 6);
      sweep<kGU>(pl, p.sync + d.in_region, in_gr, ptag, reinterpret_cast<uint32_t*>(addnorm ? L.dbuf : xs), pw * 64 + lane, kPP * 64,
                 0x40000u | (uint32_t)ph, 4);
    }
    if (pw == 0) DL_PSTAMP(ph, 7);
    if (addnorm) {
      __syncthreads();  // delta complete (waves 0..3 do the residual add + norm)
      __syncthreads();
    }
    __syncthreads();  // x of this phase is ready
  }
}

// tags / arrival counters / abort word start from zero on every call.  A kernel, not hipMemsetAsync: under stream capture the memset
// node replayed a garbage fill pattern on ROCm 7.2 (observed: the buffer came back filled with a repeating 16-byte pointer pair).
__global__ __launch_bounds__(256) void zero_sync_kernel(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

template <typename T>
__global__ __launch_bounds__(kPT, 2) void decode_persistent_kernel(const PParams p) {
  constexpr int D = kPD;
  constexpr int ANG = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Lds L;
  L.h = reinterpret_cast<uint16_t*>(smem);
  L.dbuf = L.h + p.Hpad;
  L.x[0] = L.dbuf + p.Hpad;
  L.x[1] = L.x[0] + p.Kpad;
  L.sm_att = reinterpret_cast<float*>(L.x[1] + p.Kpad);
  L.qkvraw = reinterpret_cast<uint32_t*>(L.sm_att + 2 * (2 * ANG + ANG * D));
  L.comb = reinterpret_cast<float*>(L.qkvraw + 3 * (D / 2));
  L.cnt = reinterpret_cast<int*>(L.comb + (int64_t)p.max_splits * (D + kAttnPartPad));
  L.red = reinterpret_cast<float*>(L.cnt + 24);
  uint4* tab = reinterpret_cast<uint4*>(L.cnt + 32);
  L.tab = reinterpret_cast<const PhaseDev*>(tab);
  if (threadIdx.x < 32) L.cnt[threadIdx.x] = 0;
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.phases);
    const int n16 = p.n_phases * (int)(sizeof(PhaseDev) / 16);
    for (int i = threadIdx.x; i < n16; i += kPT) tab[i] = src[i];
  }
  __syncthreads();
  Poll pl{p.sync, p.spin_limit, false};
  const int T0 = p.kv_len0[0], T1 = p.kv_len1[0], pos = p.pos_base[0];
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid < kPS) streamer_loop<T>(p, L, pl, T0, T1, pos);
  else poller_loop<T>(p, L, pl, T0, T1, pos);
}

}  // namespace dl

using namespace dl;

namespace {
struct PLayout {
  int64_t r_cnt, r_qkv, r_part, r_attn, r_o, r_act, r_dn, total;  // granules
};
PLayout p_layout(int n_phases, int H, int I, int n_heads, int n_kv_heads, int max_splits) {
  PLayout L;
  int64_t o = kCtrlGr;
  auto take = [&](int64_t n) {
    const int64_t at = o;
    o += (n + 15) / 16 * 16;  // 128-byte aligned regions
    return at;
  };
  L.r_cnt = take((int64_t)n_phases * 4);  // 8 x u32 per phase
  L.r_qkv = take((int64_t)(n_heads + 2 * n_kv_heads) * kPD / 2);
  L.r_part = take((int64_t)n_heads * max_splits * kPartGr);
  L.r_attn = take((int64_t)n_heads * kPD / 2);
  L.r_o = take(H / 2);
  L.r_act = take((I + 1) / 2);
  L.r_dn = take(H / 2);
  L.total = o;
  return L;
}
size_t p_lds_bytes(int H, int Kmax, int max_splits, int n_phases) {
  const int Hpad = (H + 7) / 8 * 8, Kpad = (Kmax + 7) / 8 * 8;
  size_t b = (size_t)(2 * Hpad + 2 * Kpad) * 2;
  b += (size_t)2 * (2 * 16 + 16 * kPD) * 4;
  b += (size_t)3 * (kPD / 2) * 4;
  b += (size_t)max_splits * (kPD + kAttnPartPad) * 4;
  b += 32 * 4;
  b += (size_t)n_phases * sizeof(DlDecodePhase);
  return (b + 15) / 16 * 16;
}
}  // namespace

extern "C" int64_t dl_decode_persistent_sync_bytes(int n_phases, int H, int I, int n_heads, int n_kv_heads, int head_dim, int max_splits) {
  if (head_dim != kPD || n_phases <= 0 || max_splits < 1) return 0;
  return p_layout(n_phases, H, I, n_heads, n_kv_heads, max_splits).total * 8;
}

extern "C" int dl_decode_persistent_region(int which, int n_phases, int H, int I, int n_heads, int n_kv_heads, int max_splits, int64_t* offset_granules) {
  DL_REQUIRE(offset_granules, "dl_decode_persistent_region: NULL pointer");
  const PLayout L = p_layout(n_phases, H, I, n_heads, n_kv_heads, max_splits);
  switch (which) {
    case DL_REGION_QKV: *offset_granules = L.r_qkv; break;
    case DL_REGION_ATTN: *offset_granules = L.r_attn; break;
    case DL_REGION_O: *offset_granules = L.r_o; break;
    case DL_REGION_ACT: *offset_granules = L.r_act; break;
    case DL_REGION_DN: *offset_granules = L.r_dn; break;
    default: dl::set_error("dl_decode_persistent_region: unknown region %d", which); return DL_ERR_ARG;
  }
  return DL_OK;
}

extern "C" int dl_decode_persistent(const DlDecodePhase* phases_dev, int n_phases, void* sync_buf, int64_t sync_bytes, int H, int I, int n_heads,
                                    int n_kv_heads, int head_dim, int max_splits, float eps, const void* cos_tab, const void* sin_tab, int n_pos,
                                    const int32_t* pos_base, const int32_t* kv_len0, const int32_t* kv_len1, const int64_t* cur_ids,
                                    int64_t slab_stride_h, int T_cap, int n_workgroups, int spin_limit, void* debug_stamps, int debug_wg, int dtype,
                                    void* stream) {
  DL_REQUIRE(phases_dev && sync_buf && cos_tab && sin_tab && pos_base && kv_len0 && kv_len1 && cur_ids, "dl_decode_persistent: NULL pointer");
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_decode_persistent: 16-bit dtypes only");
  DL_REQUIRE(head_dim == kPD, "dl_decode_persistent: head_dim must be %d", kPD);
  DL_REQUIRE(H > 0 && H % 8 == 0 && H <= 8192 && I > 0 && I % 8 == 0, "dl_decode_persistent: H / I must be multiples of 8, H <= 8192");
  DL_REQUIRE(n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0 && n_heads * head_dim == H, "dl_decode_persistent: bad head counts");
  DL_REQUIRE(n_phases > 0 && max_splits >= 1 && max_splits <= 32 && n_pos > 0 && T_cap > 0, "dl_decode_persistent: bad shape");
  static int n_cu = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    (void)hipGetLastError();
  });
  int G = n_workgroups > 0 ? n_workgroups : n_cu;
  DL_REQUIRE(G > 0 && (n_cu == 0 || G <= n_cu), "dl_decode_persistent: %d workgroups cannot all be resident on %d CUs", G, n_cu);
  DL_REQUIRE(n_heads * ((max_splits + 1) / 2) <= G, "dl_decode_persistent: n_heads x splits does not fit %d workgroups", G);
  const PLayout L = p_layout(n_phases, H, I, n_heads, n_kv_heads, max_splits);
  DL_REQUIRE(sync_bytes >= L.total * 8, "dl_decode_persistent: sync buffer too small (%lld < %lld)", (long long)sync_bytes, (long long)(L.total * 8));
  const int Kmax = H > I ? H : I;
  const size_t lds = p_lds_bytes(H, Kmax, max_splits, n_phases);
  DL_REQUIRE(lds <= 160 * 1024, "dl_decode_persistent: %zu bytes of LDS needed", lds);
  hipStream_t st = as_stream(stream);
  PParams pp;
  pp.phases = reinterpret_cast<const PhaseDev*>(phases_dev);
  pp.n_phases = n_phases;
  pp.sync = reinterpret_cast<u64_t*>(sync_buf);
  pp.r_cnt = L.r_cnt; pp.r_qkv = L.r_qkv; pp.r_part = L.r_part; pp.r_attn = L.r_attn; pp.r_o = L.r_o; pp.r_act = L.r_act; pp.r_dn = L.r_dn;
  pp.H = H; pp.Hpad = (H + 7) / 8 * 8; pp.Kpad = (Kmax + 7) / 8 * 8;
  pp.n_heads = n_heads; pp.n_kv_heads = n_kv_heads; pp.max_splits = max_splits;
  pp.eps = eps; pp.scale = 1.0f / sqrtf((float)head_dim);
  pp.cos_tab = cos_tab; pp.sin_tab = sin_tab; pp.n_pos = n_pos;
  pp.pos_base = pos_base; pp.kv_len0 = kv_len0; pp.kv_len1 = kv_len1; pp.cur_ids = cur_ids;
  pp.slab_stride_h = slab_stride_h; pp.T_cap = T_cap;
  pp.spin_limit = spin_limit > 0 ? spin_limit : (1 << 18);
  pp.stamps = reinterpret_cast<long long*>(debug_stamps);
  pp.stamp_wg = debug_wg;
  {
    const int64_t n16 = L.total / 2;  // regions are 128-byte aligned: the total is a multiple of 16 granules
    const int blocks = (int)((n16 + 255) / 256 < 512 ? (n16 + 255) / 256 : 512);
    hipLaunchKernelGGL(zero_sync_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(sync_buf), n16);
  }
  auto go = [&](auto kfn) -> int {
    static std::once_flag attr_once;
    static bool attr_ok = false;
    std::call_once(attr_once, [&] {
      attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      if (!attr_ok) (void)hipGetLastError();
    });
    if (!attr_ok) {
      dl::set_error("dl_decode_persistent: cannot raise the dynamic LDS limit");
      return DL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(kPT), lds, st, pp);
    return DL_OK;
  };
  int rc;
  if (dtype == DL_BF16) rc = go(decode_persistent_kernel<bf16_t>);
  else rc = go(decode_persistent_kernel<f16_t>);
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_decode_persistent");
  return DL_OK;
}
