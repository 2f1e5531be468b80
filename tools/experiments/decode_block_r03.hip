// The weight-streaming part of a batch-1 decode layer as ONE launch on the LDS-DMA engine: o_proj -> gate|up (+ residual add, RMSNorm,
// SiLU*up) -> down_proj -> the NEXT layer's q|k|v (+ residual add, RMSNorm; lm_head after the last layer) -- DML:1127, 328, 1289-1295,
// 134-139, 1011-1013, 2709 -- with the launch path's arithmetic in the launch path's order (gemv_dot.h: same per-lane chunk order, same
// wave reduction, same roundings), so the step stays bit-identical to dl_gemv's.  The attention stays its own launch (dl_attn_decode_rope):
// the launch is cut at the one edge whose consumer cannot use a prefetched weight stream (MI355X_MICROARCH.md price list, rows allgather /
// prefetch-credit / engine-vs-launches), which turns 6 dependent launches per layer into 2.
//
// Engine (one 256-thread workgroup per CU, every workgroup resident):
//   wave 0      LOADER: streams this CU's share of every phase's weight rows HBM -> LDS with global_load_lds_dwordx4 ... nt (one
//               wave-instruction = one 1 KiB "piece" of a row, lane l lands at byte 16 l: the lane-linear image dl_gemv's lanes read
//               from HBM) into a ring of 1 KiB pieces that fills most of the 160 KiB.  It never looks at data: it runs AHEAD of the
//               consumers across phase edges until the ring is full (~32 MB chip-wide = ~5 us of HBM stream banked per edge), publishes
//               "pieces landed" through an LDS word after counted s_waitcnt vmcnt, and is thinned to one outstanding fill while its
//               own CU gathers.
//   waves 1..3  CONSUMERS: claim "units" (the two rows of one output pair; four for gate|up) by an LDS ticket, read weight pieces and x
//               from LDS (ds_read_b128), v_dot2c, wave reduction, epilogue; phase outputs travel between CUs as 8-byte {tag, bf16 pair}
//               granules (granule.h: one sc1 store publishes data and "ready" together); after its last unit of a phase a CU's
//               consumers sweep the whole output vector of that phase into LDS (the next phase's x or residual delta).
// Dealing: workgroup w owns the contiguous units [n_units w / G, n_units (w + 1) / G) of a phase; its waves take them in ring order.  Every wait is bounded; a give-up raises *err_flag (checked by generate() at its final sync).
#include <mutex>

#include "gemv_dot.h"
#include "granule.h"
#include "lds_dma_r03.h"
#include "dynllava.h"
#include "experiments_abi.h"

namespace dl {

constexpr int kBT = 256;     // threads: wave 0 loads, waves 1..3 consume
constexpr int kBC = 3;       // consumer waves
constexpr int kBPiece = 1024;
constexpr int kBMaxPh = 4;
constexpr int kBCtl = 64;    // control words in LDS
constexpr int kBLag = 32;    // pieces the loader may have in flight before it waits for the oldest (two 16 KiB fills)

struct BPhase {  // device view of dl_block_phase (same layout)
  bgc16_t W;
  bgc16_t norm_w;
  bg16_t out;
  bgc16_t x_in;
  bgc16_t h_in;
  bg16_t h_out;
  int32_t N, K, flags, reserved;
};
static_assert(sizeof(BPhase) == sizeof(dl_block_phase), "BPhase must mirror dl_block_phase");

struct BParams {
  BPhase ph[kBMaxPh];
  int n_phases;
  u64_t* sync;          // [kBMaxPh][region_gr] granule regions (region i = output of phase i)
  int region_gr;
  const int32_t* pos_base;
  int call_tag;
  float eps;
  int ring;             // pieces
  int x_bytes, h_bytes;
  int spin_limit;
  int lag;              // pieces in flight before the loader waits for the oldest (16 / 32 / 48)
  int dbg;              // measurement modes (tools/bench_block.py): 1 = consumers skip the arithmetic, 2 = loader alone (no consumers)
  int32_t* err;
  long long* stamps;    // debug: [G][n_phases][8] wall-clock stamps (NULL in production)
};

// control words (LDS ints)
enum { C_LANDED = 0, C_GATHER = 1, C_BAR = 2, C_ABORT = 3, C_START = 4 /* 3 */, C_TICKET = 8 /* kBMaxPh */, C_RED = 16 /* 4 floats */ };

// A phase's weights reach a CU as ONE stream of 1 KiB pieces: the CU's rows in unit order.  For a plain projection the rows of a CU are
// contiguous in memory, so the stream is a single sequential run and pieces ignore row boundaries (the 7B down_proj row is 21.5 pieces);
// for gate|up it alternates between the gate run and the up run row by row (rows are whole pieces there).  Row r of the CU's t-th unit
// starts at stream byte (t * rows + r) * row_bytes.
struct BGeom {
  int row_bytes;
  int rows;        // rows per unit (2, or 4 for gate|up)
  int n_units;     // units of the whole phase (output pairs)
  int first;       // this workgroup's units: [first, first + mine) -- a CONTIGUOUS block of rows
  int mine;
  int pieces;      // pieces of this workgroup's stream of the phase
  int base;        // global piece index of the stream's first piece
};

__device__ __forceinline__ BGeom block_geom(const BPhase& d, int wg, int G, int base) {
  BGeom g;
  g.row_bytes = d.K * 2;
  const bool pair = (d.flags & DL_BLK_SILU_PAIR) != 0;
  g.rows = pair ? 4 : 2;
  const int n_out = pair ? d.N / 2 : d.N;
  g.n_units = (n_out + 1) / 2;
  // contiguous dealing: a CU streams one sequential run of rows per phase (two for gate|up): the single loader wave cannot overlap its
  // own address-translation misses the way dl_gemv's sixteen waves per CU do
  g.first = (int)((int64_t)g.n_units * wg / G);
  g.mine = (int)((int64_t)g.n_units * (wg + 1) / G) - g.first;
  g.pieces = (int)(((int64_t)g.mine * g.rows * g.row_bytes + kBPiece - 1) / kBPiece);
  g.base = base;
  return g;
}

// row r (0..rows-1) of unit u: plain = rows 2u, 2u+1; gate|up = gate_2u, up_2u, gate_2u+1, up_2u+1 (rows o and N/2 + o)
__device__ __forceinline__ int block_row(const BPhase& d, int u, int r) {
  if (d.flags & DL_BLK_SILU_PAIR) {
    const int n_out = d.N / 2;
    int o = 2 * u + (r >> 1);
    o = o < n_out ? o : n_out - 1;
    return (r & 1) ? n_out + o : o;
  }
  const int n = 2 * u + r;
  return n < d.N ? n : d.N - 1;
}

// one LDS-DMA piece: 64 lanes x 16 bytes, global (per-lane address) -> LDS (wave-uniform base in M0 + 16 * lane), non-temporal.
// The compiler neither sees the LDS write nor counts the load: completion is this wave's own s_waitcnt vmcnt (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma_piece(const DL_GLOBAL void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// eight consecutive pieces of one row into eight consecutive ring slots.  The instruction offset advances BOTH addresses (LDS address =
// m0 + offset + 16 * lane, global address = s_base + v_off + offset: tools/lds_dma_probe.hip), so four pieces share one m0 / v_off:
// 1.5 instructions per piece (the loader must issue a 16 KiB fill well inside its ~0.64 us landing cadence).
__device__ __forceinline__ void dma_8pieces(const DL_GLOBAL void* s_row, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep, off2;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "v_add_u32 %1, 0x1000, %2\n\t"
      "global_load_lds_dwordx4 %2, %3 nt\n\t"
      "global_load_lds_dwordx4 %2, %3 offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %2, %3 offset:2048 nt\n\t"
      "global_load_lds_dwordx4 %2, %3 offset:3072 nt\n\t"
      "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3 nt\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&v"(off2)
      : "v"(v_off), "s"(s_row), "s"(lds_dst)
      : "memory", "scc");
}

// n = 4 / 2 / 1 pieces, same addressing (SGPR row pointer + 16 * lane + offset)
template <int N>
__device__ __forceinline__ void dma_small(const DL_GLOBAL void* s_row, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep;
  if constexpr (N == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v_off), "s"(s_row), "s"(lds_dst) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v_off), "s"(s_row), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v_off), "s"(s_row), "s"(lds_dst) : "memory");
}


#define DL_BSTAMP(ph_, slot_)                                                                                         \
  do {                                                                                                                \
    if (p.stamps && lane == 0) p.stamps[((int64_t)blockIdx.x * p.n_phases + (ph_)) * 8 + (slot_)] = wall_clock64(); \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------------------
// loader wave
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void loader_loop(const BParams& p, lvi_t ctl, uint32_t ring_lds, int lane) {
  // The loader wave is ISSUE-bound: one LDS-DMA instruction costs it ~60-90 cycles, so eight of them plus anything else must fit the
  // ~690 cycles in which 8 KiB land at 28 GB/s (tools/lds_dma_bw.hip: 7.2 TB/s chip-wide from one such wave per CU).  The first version of
  // this loop (per-piece ring arithmetic, three LDS reads and VALU compares per group) ran at 18 GB/s.  Hot path now: SALU space check
  // against a cached tail, one asm group, one counted wait, one LDS store.
  const int G = gridDim.x, wg = blockIdx.x;
  const uint32_t voff = (uint32_t)lane * 16u;
  int gp = 0;          // pieces issued so far (global piece index of the next one)
  int tail = 0;        // cached: pieces below this index are free (min of the consumers' row starts)
  int slot = 0;        // gp mod ring
  int since = 0;
  int base = 0;
  for (int ph = 0; ph < p.n_phases; ++ph) {
    const BPhase& d = p.ph[ph];
    const BGeom g = block_geom(d, wg, G, base);
    base += g.pieces;
    const bool pair = (d.flags & DL_BLK_SILU_PAIR) != 0;
    // segments: plain = one run of g.pieces; gate|up = mine * 4 runs of one row each, alternating gate / up
    const int n_seg = pair ? g.mine * 4 : (g.mine > 0 ? 1 : 0);
    const int seg_pieces = pair ? g.row_bytes / kBPiece : g.pieces;
    const DL_GLOBAL char* gate = (const DL_GLOBAL char*)d.W + (int64_t)(2 * g.first) * g.row_bytes;
    const DL_GLOBAL char* up = gate + (int64_t)(d.N / 2) * g.row_bytes;
    const int64_t seg_bytes_total = (int64_t)g.mine * g.rows * g.row_bytes;  // plain: bytes of the run (its last piece may be partial)
    long long n_block = 0, t_block = 0;
    for (int sg = 0; sg < n_seg; ++sg) {
      const DL_GLOBAL char* ptr = pair ? ((sg & 1) ? up + (int64_t)(sg >> 1) * g.row_bytes : gate + (int64_t)(sg >> 1) * g.row_bytes) : gate;
      for (int pc = 0; pc < seg_pieces;) {
        int n = seg_pieces - pc;
        const int room = p.ring - slot;
        n = n < room ? n : room;
        n = n >= 8 ? 8 : (n >= 4 ? 4 : (n >= 2 ? 2 : 1));  // two groups per wait were measured slower (gate|up 35.5 vs 30.5 us loader-only)
        if (gp + n - p.ring > tail) {
          // out of KNOWN space: refresh the cached tail; if the consumers really are that far behind (an edge: they are gathering),
          // publish what is in flight step by step while waiting -- a consumer may be waiting for exactly the newest pieces
          int stage = 0;
          const long long tb0 = p.stamps ? wall_clock64() : 0;
          for (int spins = 0;; ++spins) {
            const int a = lds_ld(ctl + C_START), b = lds_ld(ctl + C_START + 1), c = lds_ld(ctl + C_START + 2);
            tail = min(a, min(b, c));
            if (gp + n - p.ring <= tail) {
              if (p.stamps && spins > 0) {
                ++n_block;
                t_block += wall_clock64() - tb0;
              }
              break;
            }
            if (stage == 0) {
              asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
              if (lane == 0) ctl[C_LANDED] = gp - 16;
            } else if (stage == 1) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              if (lane == 0) ctl[C_LANDED] = gp;
            }
            stage = stage < 2 ? stage + 1 : 2;
            since = 0;
            if (lds_ld(ctl + C_ABORT) || spins > p.spin_limit) {
              if (lane == 0) {
                ctl[C_ABORT] = 1;
                if (p.err) atomicOr(p.err, 0x100 | ph);
              }
              return;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        const uint32_t dst = ring_lds + (uint32_t)slot * kBPiece;
        const DL_GLOBAL char* src = ptr + (int64_t)pc * kBPiece;
        if (n == 8) {
          dma_8pieces(src, voff, dst);
        } else if (n == 4) {
          dma_small<4>(src, voff, dst);
        } else if (n == 2) {
          dma_small<2>(src, voff, dst);
        } else {
          // the run's last piece may be partial: the surplus lanes re-read its last chunk (never past the matrix)
          const int64_t left = seg_bytes_total - (int64_t)pc * kBPiece;
          uint32_t vo = voff;
          if (!pair && left < kBPiece) vo = voff + 16 <= (uint32_t)left ? voff : (uint32_t)left - 16u;
          dma_small<1>(src, vo, dst);
        }
        pc += n;
        gp += n;
        slot += n;
        slot = slot >= p.ring ? 0 : slot;
        since += n;
        if (since >= 8) {
          if (p.lag >= 48) {
            asm volatile("s_waitcnt vmcnt(48)" ::: "memory");  // everything but the newest 48 pieces has landed
            if (lane == 0) ctl[C_LANDED] = gp - 48;
          } else if (p.lag >= 32) {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            if (lane == 0) ctl[C_LANDED] = gp - 32;
          } else {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (lane == 0) ctl[C_LANDED] = gp - 16;
          }
          since = 0;
        }
      }
    }
    if (p.stamps && lane == 0) {
      long long* st = p.stamps + ((int64_t)blockIdx.x * p.n_phases + ph) * 8;
      st[5] = wall_clock64();
      st[6] = n_block;
      st[7] = t_block;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) ctl[C_LANDED] = gp;
}

// ---------------------------------------------------------------------------------------------------------------------------
// consumers
// ---------------------------------------------------------------------------------------------------------------------------
struct CSt {
  lvi_t ctl;
  int spin_limit;
  int32_t* err;
  int bar_epoch;
  bool dead;
};

__device__ __forceinline__ void c_fail(CSt& s, int code, int lane) {
  s.dead = true;
  if (lane == 0) {
    s.ctl[C_ABORT] = 1;
    if (s.err) atomicOr(s.err, code);
  }
}

// barrier of the three consumer waves (the loader never joins: s_barrier would stall its stream)
__device__ __forceinline__ void c_barrier(CSt& s, int lane) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  ++s.bar_epoch;
  if (lane == 0) lds_add(s.ctl + C_BAR, 1);
  for (int spins = 0; !s.dead; ++spins) {
    if (lds_ld(s.ctl + C_BAR) >= kBC * s.bar_epoch) break;
    if (lds_ld(s.ctl + C_ABORT) || spins > s.spin_limit) {
      c_fail(s, 0x200, lane);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// sweep `n_gr` granules of `region` (tag `tag`) into LDS words dst[i]; the 192 consumer threads share the range, a lane re-polls only
// what it is still missing (same scheme as the round-2 kernel: the granules validate themselves)
template <int GU>
__device__ __forceinline__ void c_sweep(CSt& s, const u64_t* region, int n_gr, uint32_t tag, DL_LDS uint32_t* dst, int t, int lane, int code) {
  constexpr int NT = kBC * 64;
  for (int base = 0; base < n_gr; base += NT * GU) {
    u64_t v[GU];
    uint32_t have = 0;
    for (int spins = 0;; ++spins) {
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int idx = base + k * NT + t;
        if (idx < n_gr && !((have >> k) & 1)) v[k] = gr_load(region + idx);
      }
      bool ok = true;
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int idx = base + k * NT + t;
        if (idx < n_gr && !((have >> k) & 1)) {
          const bool h = (uint32_t)(v[k] >> 32) == tag;
          have |= (uint32_t)h << k;
          ok &= h;
        }
      }
      if (__all(ok) || s.dead) break;
      if ((spins & 15) == 15 && lds_ld(s.ctl + C_ABORT)) {
        s.dead = true;
        break;
      }
      if (spins > s.spin_limit) {
        c_fail(s, code, lane);
        break;
      }
      __builtin_amdgcn_s_sleep(3);
    }
#pragma unroll
    for (int k = 0; k < GU; ++k) {
      const int idx = base + k * NT + t;
      if (idx < n_gr) dst[idx] = (uint32_t)v[k];
    }
  }
}

// cheap readiness poll on the granules of the LAST dealing rounds (they finish last): one 8-byte load per lane
__device__ __forceinline__ void c_wait_sample(CSt& s, const u64_t* region, int n_gr, uint32_t tag, int lane, int code) {
  if (s.dead || n_gr <= 0) return;
  int idx = n_gr - 1 - lane * 4;
  idx = idx < 0 ? 0 : idx;
  for (int spins = 0;; ++spins) {
    const u64_t v = gr_load(region + idx);
    if (__all((uint32_t)(v >> 32) == tag)) return;
    if ((spins & 15) == 15 && lds_ld(s.ctl + C_ABORT)) {
      s.dead = true;
      return;
    }
    if (spins > s.spin_limit) {
      c_fail(s, code, lane);
      return;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

template <typename T>
__device__ __forceinline__ void consumer_loop(const BParams& p, lvi_t ctl, l8_t ring, l8_t xs, l8_t hs, int tid) {
  constexpr int V = 8;
  const int lane = tid & 63;
  const int cw = __builtin_amdgcn_readfirstlane((tid >> 6) - 1);  // consumer wave 0..2
  const int ct = cw * 64 + lane;                                    // consumer thread 0..191
  const int G = gridDim.x, wg = blockIdx.x;
  CSt s{ctl, p.spin_limit, p.err, 0, false};
  const uint32_t tag0 = ((((uint32_t)p.pos_base[0] & 0x3fffffu) << 10) | (((uint32_t)p.call_tag & 0xffu) << 2)) + 1u;
  int base = 0;
  int landed = 0;
  for (int ph = 0; ph < p.n_phases; ++ph) {
    const BPhase& d = p.ph[ph];
    const BGeom g = block_geom(d, wg, G, base);
    base += g.pieces;
    const int nvec = d.K / V;
    const bool addnorm = (d.flags & DL_BLK_ADDNORM) != 0;
    const bool pair = (d.flags & DL_BLK_SILU_PAIR) != 0;
    if (cw == 0) DL_BSTAMP(ph, 0);
    // ---- prologue.  Every global load it needs (input vector, residual stream, norm weight) is requested FIRST, in one round trip: a
    // dependent load costs ~2 us here (the CU's own loader keeps its memory queue full), and the loader can only bank ~4.6 us of stream
    // in the ring while the consumers are not consuming.  add+norm replays dl_gemv's ADDNORM prologue (thread t of its 256 owns chunks
    // t + 256 c): three waves stand in for its four -- consumer wave w plays virtual wave w, wave 0 also plays virtual wave 3 -- with the
    // same chunk -> lane map and summation order: bit-identical.
    constexpr int MAXC = 4;
    const int n_role = addnorm ? (cw == 0 ? 2 : 1) : 0;
    bu32x4_t nwr[2][MAXC], hr[2][MAXC], dr[2][MAXC];
    if (addnorm) {
#pragma unroll
      for (int ro = 0; ro < 2; ++ro) {
        if (ro < n_role) {
          const int tt = (ro == 0 ? cw : 3) * 64 + lane;
#pragma unroll
          for (int c = 0; c < MAXC; ++c) {
            const int v = tt + c * 256;
            if (v < nvec) {
              nwr[ro][c] = *(const DL_GLOBAL bu32x4_t*)(d.norm_w + (int64_t)v * V);
              if (d.h_in) hr[ro][c] = *(const DL_GLOBAL bu32x4_t*)(d.h_in + (int64_t)v * V);
              if (d.x_in) dr[ro][c] = *(const DL_GLOBAL bu32x4_t*)(d.x_in + (int64_t)v * V);
            }
          }
        }
      }
    } else if (d.x_in) {
      for (int v = ct; v < nvec; v += kBC * 64) *(DL_LDS bu32x4_t*)(xs + v * 16) = *(const DL_GLOBAL bu32x4_t*)(d.x_in + (int64_t)v * V);
    }
    if (!d.x_in && ph > 0) {
      // the previous phase's output vector, from its granules (it lands in xs as bf16 pairs = the chunk layout)
      if (lane == 0 && cw == 0) ctl[C_GATHER] = 1;
      const u64_t* region = p.sync + (int64_t)(ph - 1) * p.region_gr;
      const uint32_t gtag = tag0 + (uint32_t)(ph - 1);
      c_wait_sample(s, region, d.K / 2, gtag, lane, 0x400 | ph);
      if (cw == 0) DL_BSTAMP(ph, 1);
      c_sweep<8>(s, region, d.K / 2, gtag, (DL_LDS uint32_t*)xs, ct, lane, 0x800 | ph);
      c_barrier(s, lane);
      if (lane == 0 && cw == 0) ctl[C_GATHER] = 0;
    }
    if (cw == 0) DL_BSTAMP(ph, 2);
    if (addnorm) {
      DL_LDS volatile float* red = (DL_LDS volatile float*)(ctl + C_RED);
      float a[2][MAXC][V];
#pragma unroll
      for (int ro = 0; ro < 2; ++ro) {
        if (ro < n_role) {
          const int vw = ro == 0 ? cw : 3;
          const int tt = vw * 64 + lane;
          float ss = 0.f;
#pragma unroll
          for (int c = 0; c < MAXC; ++c) {
            const int v = tt + c * 256;
            if (v < nvec) {
              if (d.h_in) unpack16<T>(make_uint4(hr[ro][c].x, hr[ro][c].y, hr[ro][c].z, hr[ro][c].w), a[ro][c]);
              else unpack16<T>(lds_ld16(hs + v * 16), a[ro][c]);
              float dd[V];
              if (d.x_in) unpack16<T>(make_uint4(dr[ro][c].x, dr[ro][c].y, dr[ro][c].z, dr[ro][c].w), dd);
              else unpack16<T>(lds_ld16(xs + v * 16), dd);
#pragma unroll
              for (int e = 0; e < V; ++e) a[ro][c][e] = Elem<T>::round(a[ro][c][e] + dd[e]);
              const uint4 packed = pack16<T>(a[ro][c]);
              lds_st16(hs + v * 16, packed);
              if (d.h_out && wg == 0) {
                bu32x4_t o;
                o.x = packed.x; o.y = packed.y; o.z = packed.z; o.w = packed.w;
                *(DL_GLOBAL bu32x4_t*)(d.h_out + (int64_t)v * V) = o;
              }
#pragma unroll
              for (int e = 0; e < V; ++e) ss += a[ro][c][e] * a[ro][c][e];
            }
          }
          ss = wave_sum(ss);
          if (lane == 0) red[vw] = ss;
        }
      }
      c_barrier(s, lane);
      float tsum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) tsum += red[i];
      const float rstd = rsqrtf(tsum / (float)d.K + p.eps);
#pragma unroll
      for (int ro = 0; ro < 2; ++ro) {
        if (ro < n_role) {
          const int tt = (ro == 0 ? cw : 3) * 64 + lane;
#pragma unroll
          for (int c = 0; c < MAXC; ++c) {
            const int v = tt + c * 256;
            if (v < nvec) {
              float w[V];
              unpack16<T>(make_uint4(nwr[ro][c].x, nwr[ro][c].y, nwr[ro][c].z, nwr[ro][c].w), w);
#pragma unroll
              for (int e = 0; e < V; ++e) a[ro][c][e] = w[e] * Elem<T>::round(a[ro][c][e] * rstd);
              lds_st16(xs + v * 16, pack16<T>(a[ro][c]));
            }
          }
        }
      }
    }
    c_barrier(s, lane);
    if (cw == 0) DL_BSTAMP(ph, 3);

    // ---- stream: claim units in ring order ----
    const uint32_t tag = tag0 + (uint32_t)ph;
    u64_t* out_region = p.sync + (int64_t)ph * p.region_gr;
    const uint32_t ring_bytes = (uint32_t)p.ring * kBPiece;
    const bool pow2 = (ring_bytes & (ring_bytes - 1)) == 0;
    const uint32_t phase_rb = ((uint32_t)g.base % (uint32_t)p.ring) * kBPiece;  // ring byte offset of the stream's first byte
    const uint32_t row_bytes = (uint32_t)g.row_bytes, unit_bytes = row_bytes * (uint32_t)g.rows;
    const int np = (g.row_bytes + kBPiece - 1) / kBPiece;
    for (;;) {
      int t = 0;
      if (lane == 0) t = lds_add(ctl + C_TICKET + ph, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= g.mine || s.dead) break;
      const int u = g.first + t;
      float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;
      // rows are taken two at a time (the two rows of an output pair; gate and up of one output): they share the x reads and give the
      // wave two independent chains.  Per lane the chunks of a row are still visited in ascending order: dl_gemv's sums, bit for bit.
#pragma unroll 1
      for (int r = 0; r < g.rows; r += 2) {
        const uint32_t off0 = (uint32_t)t * unit_bytes + (uint32_t)r * row_bytes;  // stream byte offset of the first row of the pair
        const int p0 = g.base + (int)(off0 >> 10);
        const int p1 = g.base + (int)((off0 + 2 * row_bytes + kBPiece - 1) >> 10);  // one past the pair's last piece
        if (lane == 0) ctl[C_START + cw] = p0;  // everything below is consumed: the ring may reuse it
        if (landed < p1) {
          for (int spins = 0;; ++spins) {
            landed = lds_ld(ctl + C_LANDED);
            if (landed >= p1) break;
            if (((spins & 31) == 31 && lds_ld(ctl + C_ABORT)) || spins > s.spin_limit) {
              c_fail(s, 0x1000 | ph, lane);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          if (s.dead) break;
        }
        asm volatile("" ::: "memory");
        uint32_t rb0 = phase_rb + off0;
        rb0 = pow2 ? (rb0 & (ring_bytes - 1)) : rb0 % ring_bytes;
        uint32_t rb1 = rb0 + row_bytes;
        rb1 = rb1 >= ring_bytes ? rb1 - ring_bytes : rb1;
        const uint32_t a0 = rb0 + lane * 16, a1 = rb1 + lane * 16;
        float acc0 = 0.f, acc1 = 0.f;
        int pc = (p.dbg & 1) ? np : 0;
        for (; pc + 8 <= np && (pc + 8) * 64 <= nvec; pc += 8) {  // eight full pieces of both rows: 24 LDS reads in flight
          uint4 w0[8], w1[8], xv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            uint32_t b0 = a0 + (uint32_t)(pc + j) * kBPiece, b1 = a1 + (uint32_t)(pc + j) * kBPiece;
            b0 = b0 >= ring_bytes ? b0 - ring_bytes : b0;
            b1 = b1 >= ring_bytes ? b1 - ring_bytes : b1;
            w0[j] = lds_ld16(ring + b0);
            w1[j] = lds_ld16(ring + b1);
            xv[j] = lds_ld16(xs + (pc + j) * kBPiece + lane * 16);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc0 = dot16<T>(w0[j], xv[j], acc0);
            acc1 = dot16<T>(w1[j], xv[j], acc1);
          }
        }
        for (; pc < np; ++pc) {
          if (pc * 64 + lane < nvec) {
            uint32_t b0 = a0 + (uint32_t)pc * kBPiece, b1 = a1 + (uint32_t)pc * kBPiece;
            b0 = b0 >= ring_bytes ? b0 - ring_bytes : b0;
            b1 = b1 >= ring_bytes ? b1 - ring_bytes : b1;
            const uint4 xv = lds_ld16(xs + pc * kBPiece + lane * 16);
            acc0 = dot16<T>(lds_ld16(ring + b0), xv, acc0);
            acc1 = dot16<T>(lds_ld16(ring + b1), xv, acc1);
          }
        }
        acc0 = wave_sum(acc0);
        acc1 = wave_sum(acc1);
        if (r == 0) {
          res0 = acc0;
          res1 = acc1;
        } else {
          res2 = acc0;
          res3 = acc1;
        }
      }
      if (s.dead) break;
      uint32_t lo, hi;
      if (pair) {
        const float g0 = Elem<T>::round(res0), u0 = Elem<T>::round(res1), g1 = Elem<T>::round(res2), u1 = Elem<T>::round(res3);
        lo = Elem<T>::from_f(Elem<T>::round(g0 / (1.0f + expf(-g0))) * u0);
        hi = Elem<T>::from_f(Elem<T>::round(g1 / (1.0f + expf(-g1))) * u1);
      } else {
        lo = Elem<T>::from_f(res0);
        hi = Elem<T>::from_f(res1);
      }
      if (lane == 0) {
        if (d.out) {
          const int n_out = pair ? d.N / 2 : d.N;
          if (2 * u + 1 < n_out) *(DL_GLOBAL uint32_t*)(d.out + 2 * (int64_t)u) = lo | (hi << 16);
          else d.out[2 * (int64_t)u] = (uint16_t)lo;
        } else {
          gr_store(out_region + u, tag, lo | (hi << 16));
        }
      }
    }
    // done with this phase's pieces: anything this wave claims later starts at the next phase
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) ctl[C_START + cw] = ph + 1 < p.n_phases ? base : 0x7fffffff;
    if (cw == 0) DL_BSTAMP(ph, 4);
  }
}

template <typename T>
__global__ __launch_bounds__(kBT, 1) void decode_block_kernel(const BParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  l8_t ring = (l8_t)smem;
  l8_t xs = ring + p.ring * kBPiece;
  l8_t hs = xs + p.x_bytes;
  lvi_t ctl = (lvi_t)(hs + p.h_bytes);
  const int tid = threadIdx.x;
  if (tid < kBCtl) ctl[tid] = 0;
  __syncthreads();
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.dbg & 2) {  // measurement mode: loader alone (no consumers, unlimited ring space): the raw rate of this loader loop
    if (tid < 3) ctl[C_START + tid] = 0x7fffffff;
    __syncthreads();
    if (wid == 0) loader_loop(p, ctl, (uint32_t)(uintptr_t)ring, tid & 63);
    return;
  }
  if (wid == 0) loader_loop(p, ctl, (uint32_t)(uintptr_t)ring, tid & 63);
  else consumer_loop<T>(p, ctl, ring, xs, hs, tid);
}

}  // namespace dl

using namespace dl;

extern "C" int64_t dl_decode_block_sync_bytes(int max_k) {
  if (max_k <= 0) return 0;
  const int64_t region = ((int64_t)(max_k + 1) / 2 + 15) / 16 * 16;
  return (int64_t)kBMaxPh * region * 8;
}

extern "C" int dl_decode_block(const dl_block_phase* phases, int n_phases, void* sync_buf, int64_t sync_bytes, const int32_t* pos_base, int call_tag,
                               float eps, int32_t* err_flag, int n_workgroups, int spin_limit, void* debug_stamps, int debug_mode, int dtype, void* stream) {
  DL_REQUIRE(phases && sync_buf && pos_base, "dl_decode_block: NULL pointer");
  DL_REQUIRE(n_phases >= 1 && n_phases <= kBMaxPh, "dl_decode_block: 1..%d phases", kBMaxPh);
  DL_REQUIRE(dtype == DL_BF16 || dtype == DL_F16, "dl_decode_block: 16-bit dtypes only");
  static int n_cu = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    (void)hipGetLastError();
  });
  const int G = n_workgroups > 0 ? n_workgroups : n_cu;
  DL_REQUIRE(G > 0 && (n_cu == 0 || G <= n_cu), "dl_decode_block: %d workgroups cannot all be resident on %d CUs", G, n_cu);
  BParams pp;
  memset(&pp, 0, sizeof(pp));
  int kmax = 0, hmax = 0;
  for (int i = 0; i < n_phases; ++i) {
    const dl_block_phase& d = phases[i];
    DL_REQUIRE(d.W && d.N > 0 && d.K > 0 && d.K % 8 == 0 && d.N % 2 == 0, "dl_decode_block: phase %d: bad shape N=%d K=%d", i, d.N, d.K);
    DL_REQUIRE(!(d.flags & DL_BLK_SILU_PAIR) || d.N % 4 == 0, "dl_decode_block: phase %d: gate|up needs N %% 4 == 0", i);
    DL_REQUIRE(i > 0 || d.x_in, "dl_decode_block: phase 0 reads its input vector from memory (x_in)");
    DL_REQUIRE(i == 0 || !d.x_in, "dl_decode_block: only phase 0 may read x_in");
    if (i > 0) {
      const dl_block_phase& q = phases[i - 1];
      const int n_prev = (q.flags & DL_BLK_SILU_PAIR) ? q.N / 2 : q.N;
      DL_REQUIRE(!q.out && n_prev == d.K, "dl_decode_block: phase %d consumes phase %d's output: K must be %d and phase %d must not write to memory", i,
                 i - 1, n_prev, i - 1);
    }
    DL_REQUIRE(i + 1 < n_phases || d.out, "dl_decode_block: the last phase writes to memory (out)");
    if (d.flags & DL_BLK_ADDNORM) {
      DL_REQUIRE(d.norm_w && d.K <= 256 * 4 * 8, "dl_decode_block: phase %d: add+norm needs norm_w and K <= 8192", i);
      DL_REQUIRE(d.h_in || hmax == d.K, "dl_decode_block: phase %d: no residual stream (h_in) for the first add+norm", i);
      hmax = d.K;
    }
    kmax = d.K > kmax ? d.K : kmax;
    memcpy(&pp.ph[i], &d, sizeof(d));
  }
  pp.n_phases = n_phases;
  pp.sync = reinterpret_cast<u64_t*>(sync_buf);
  pp.region_gr = (int)(dl_decode_block_sync_bytes(kmax) / 8 / kBMaxPh);
  DL_REQUIRE(sync_bytes >= dl_decode_block_sync_bytes(kmax), "dl_decode_block: sync buffer too small");
  pp.pos_base = pos_base;
  pp.call_tag = call_tag;
  pp.eps = eps;
  pp.x_bytes = (kmax * 2 + 1023) / 1024 * 1024;  // whole pieces: the consumers read x piece by piece
  pp.h_bytes = (hmax * 2 + 15) / 16 * 16;
  const int lds_total = 160 * 1024;
  pp.ring = (lds_total - pp.x_bytes - pp.h_bytes - kBCtl * 4) / kBPiece;
  pp.ring = pp.ring > 128 ? 128 : pp.ring;
  DL_REQUIRE(pp.ring >= 64, "dl_decode_block: K=%d leaves no room for the weight ring", kmax);
  for (int i = 0; i < n_phases; ++i) {
    const int np = (phases[i].K * 2 + kBPiece - 1) / kBPiece + 1;
    DL_REQUIRE(2 * np + 8 <= pp.ring, "dl_decode_block: a row pair of phase %d (2 x %d pieces) does not fit the ring", i, np);
    DL_REQUIRE(!(phases[i].flags & DL_BLK_SILU_PAIR) || (phases[i].K * 2) % kBPiece == 0, "dl_decode_block: gate|up rows must be whole KiB (K %% 512 == 0)");
  }
  pp.spin_limit = spin_limit > 0 ? spin_limit : (1 << 20);
  pp.lag = kBLag;
  pp.dbg = debug_mode;
  pp.err = err_flag;
  pp.stamps = reinterpret_cast<long long*>(debug_stamps);
  const size_t lds = (size_t)pp.ring * kBPiece + pp.x_bytes + pp.h_bytes + kBCtl * 4;
  hipStream_t st = as_stream(stream);
  auto go = [&](auto kfn) -> int {
    static std::once_flag attr_once;
    static bool attr_ok = false;
    std::call_once(attr_once, [&] {
      attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      if (!attr_ok) (void)hipGetLastError();
    });
    if (!attr_ok) {
      dl::set_error("dl_decode_block: cannot raise the dynamic LDS limit");
      return DL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(kBT), lds, st, pp);
    return DL_OK;
  };
  int rc;
  if (dtype == DL_BF16) rc = go(decode_block_kernel<bf16_t>);
  else rc = go(decode_block_kernel<f16_t>);
  if (rc != DL_OK) return rc;
  DL_CHECK_LAUNCH("dl_decode_block");
  return DL_OK;
}
