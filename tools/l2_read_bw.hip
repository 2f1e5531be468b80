// L2-hit read ceilings per CU on gfx950: every workgroup (one per CU) re-reads the same `bytes`-byte buffer (<= 4 MiB: resident in each XCD's L2)
//   mode 0: global_load_dwordx4 into VGPRs (16 B / lane, 1 KiB contiguous per wave instruction), 8 loads in flight per wave
//   mode 1: global_load_lds_dwordx4 (LDS-DMA) into a ring in LDS, 16 pieces in flight per wave
//   mode 2: both, half the waves each
//   mode 3: 2 DMA waves stream a 1 GiB buffer (HBM misses, non-temporal) while the other waves re-read the small buffer into VGPRs (L2 hits):
//           what the L2-hit side keeps of its ceiling beside an HBM stream through the same CU (dl_linear_packed's consumers beside its loaders)
// hipcc --offload-arch=gfx950 -O3 tools/l2_read_bw.hip -o /tmp/l2bw && /tmp/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ void dma_piece(const GLOBAL void* s_base, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(v_off), "s"(s_base), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void dma_piece_nt(const GLOBAL void* s_base, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(v_off), "s"(s_base), "s"(lds_dst) : "memory");
}

// mode 3
template <int UNR>
__global__ __launch_bounds__(1024) void mixed(const uint32_t* __restrict__ buf, uint32_t bytes, const char* __restrict__ big, int iters, int hbm_iters, uint32_t* out, long long* t_hit) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = (blockDim.x >> 6) - 2;
  if (w < 2) {  // HBM stream: this CU's own 2 x hbm_iters x 8 KiB slice of `big`
    const GLOBAL char* base = (const GLOBAL char*)big + ((size_t)blockIdx.x * 2 + w) * (size_t)hbm_iters * 8192;
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)w * 32768u;
    for (int it = 0; it < hbm_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) dma_piece_nt(base, (uint32_t)it * 8192u + u * 1024u + lane * 16u, ring + (uint32_t)__builtin_amdgcn_readfirstlane(((it & 3) * 8 + u)) * 1024u);
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const uint32_t pieces = bytes / 1024;
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ acc = {0, 0, 0, 0};
  uint32_t pc = ((w - 2) + blockIdx.x * 7) % pieces;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters * 8 / UNR; ++it) {
    u32x4_ v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      v[u] = *reinterpret_cast<const u32x4_*>(reinterpret_cast<const char*>(buf) + (size_t)pc * 1024 + lane * 16);
      pc += nw;
      pc = pc >= pieces ? pc - pieces : pc;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc ^= v[u];
  }
  if (lane == 0 && w == 2) t_hit[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
  if (acc.x == 0x12345678u) out[threadIdx.x] = acc.y ^ acc.z ^ acc.w;
}

// mode 4: like mode 3, but the L2-hit side is LDS-DMA too (waves 2 .. 2 + n_hit): what dl_linear_packed would get if X went through the loaders
__global__ __launch_bounds__(1024) void mixed_dma(const uint32_t* __restrict__ buf, uint32_t bytes, const char* __restrict__ big, int iters, int hbm_iters, long long* t_hit) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = (blockDim.x >> 6) - 2;
  const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)w * 16384u;
  if (w < 2) {
    const GLOBAL char* base = (const GLOBAL char*)big + ((size_t)blockIdx.x * 2 + w) * (size_t)hbm_iters * 8192;
    for (int it = 0; it < hbm_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) dma_piece_nt(base, (uint32_t)it * 8192u + u * 1024u + lane * 16u, ring + (uint32_t)__builtin_amdgcn_readfirstlane(((it & 1) * 8 + u)) * 1024u);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const uint32_t pieces = bytes / 1024;
  uint32_t pc = ((w - 2) + blockIdx.x * 7) % pieces;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dma_piece((const GLOBAL void*)buf, pc * 1024u + lane * 16u, ring + (uint32_t)__builtin_amdgcn_readfirstlane(((it & 1) * 8 + u)) * 1024u);
      pc += nw;
      pc = pc >= pieces ? pc - pieces : pc;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && w == 2) t_hit[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int MODE>
__global__ __launch_bounds__(1024) void l2bw(const uint32_t* __restrict__ buf, uint32_t bytes, int iters, uint32_t* out) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const uint32_t pieces = bytes / 1024;  // 1 KiB pieces
  u32x4 acc = {0, 0, 0, 0};
  const bool dma = MODE == 1 || (MODE == 2 && (w & 1));
  // wave w of workgroup b walks pieces w, w + nw, ... (offset by the block so that CUs are not in lock step)
  uint32_t pc = (w + blockIdx.x * 7) % pieces;
  if (!dma) {
    for (int it = 0; it < iters; ++it) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(buf) + (size_t)pc * 1024 + lane * 16);
        pc += nw;
        pc = pc >= pieces ? pc - pieces : pc;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
  } else {
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)w * 16384u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dma_piece((const GLOBAL void*)buf, pc * 1024u + lane * 16u, ring + (uint32_t)__builtin_amdgcn_readfirstlane((it & 1) * 8 + u) * 1024u);
        pc += nw;
        pc = pc >= pieces ? pc - pieces : pc;
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc.x == 0x12345678u) out[threadIdx.x] = acc.y ^ acc.z ^ acc.w;
}

int main() {
  const uint32_t bytes = 1536 * 1024;
  uint32_t *buf, *out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 4096 * 4);
  std::vector<uint32_t> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
  hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 400;
  auto run = [&](int mode, int waves) {
    const size_t smem = mode ? (size_t)waves * 16384 : 0;
    auto k = mode == 0 ? l2bw<0> : mode == 1 ? l2bw<1> : l2bw<2>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (smem > 160 * 1024) return;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), smem, 0, buf, bytes, iters, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), smem, 0, buf, bytes, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double total = 256.0 * waves * iters * 8 * 1024;
    printf("mode %d waves/CU %2d: %.1f us, %.2f TB/s aggregate, %.1f GB/s per CU (%.1f B/clk at 2.4 GHz)\n", mode, waves, ms * 1e3, total / ms / 1e9, total / 256 / ms / 1e6, total / 256 / (ms * 1e-3) / 2.4e9);
  };
  for (int mode = 0; mode < 3; ++mode)
    for (int waves : {2, 4, 6, 8, 16}) run(mode, waves);
  // mode 3: 4 L2-hit waves beside 2 HBM-streaming DMA waves
  char* big; hipMalloc(&big, (size_t)1 << 30);
  hipMemset(big, 1, (size_t)1 << 30);
  long long* t_hit; hipMalloc(&t_hit, 256 * 8);
  hipFuncSetAttribute((const void*)mixed<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)mixed<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int cfg_ = 0; cfg_ < 4; ++cfg_) {
    const int hit_waves = cfg_ == 0 ? 4 : cfg_ == 1 ? 8 : cfg_ == 2 ? 4 : 12, unr = cfg_ == 2 ? 16 : 8;
    auto mk = unr == 16 ? mixed<16> : mixed<8>;
    for (int hbm_iters : {0, 200}) {  // 2 waves x hbm_iters x 8 KiB per CU: 0 / 3.2 MiB per CU
      const int it_hit = 300;
      printf("[%d loads in flight per hit wave] ", unr);
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(mk, dim3(256), dim3((hit_waves + 2) * 64), 65536, 0, buf, bytes, big, it_hit, hbm_iters, out, t_hit);
      hipEventRecord(a);
      hipLaunchKernelGGL(mk, dim3(256), dim3((hit_waves + 2) * 64), 65536, 0, buf, bytes, big, it_hit, hbm_iters, out, t_hit);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<long long> th(256); hipMemcpy(th.data(), t_hit, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : th) avg += (double)v; avg /= 256;
      const double hit_bytes = (double)hit_waves * it_hit * 8 * 1024, hbm_bytes = 2.0 * hbm_iters * 8192;
      printf("mode 3 hit waves %d, HBM %.1f MiB/CU: launch %.1f us; hit side: %.0f memtime ticks per CU for %.0f KiB -> %.2f B/tick/CU; HBM side %.2f TB/s over the launch\n", hit_waves, hbm_bytes / 1048576.0, ms * 1e3, avg, hit_bytes / 1024, hit_bytes / avg, 256 * hbm_bytes / ms / 1e9);
    }
  }
  hipFuncSetAttribute((const void*)mixed_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int hit_waves : {1, 2, 4}) {
    for (int hbm_iters : {0, 200}) {
      const int it_hit = 600;
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(mixed_dma, dim3(256), dim3((hit_waves + 2) * 64), (hit_waves + 2) * 16384, 0, buf, bytes, big, it_hit, hbm_iters, t_hit);
      hipEventRecord(a);
      hipLaunchKernelGGL(mixed_dma, dim3(256), dim3((hit_waves + 2) * 64), (hit_waves + 2) * 16384, 0, buf, bytes, big, it_hit, hbm_iters, t_hit);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<long long> th(256); hipMemcpy(th.data(), t_hit, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : th) avg += (double)v; avg /= 256;
      const double hit_bytes = (double)hit_waves * it_hit * 8 * 1024, hbm_bytes = 2.0 * hbm_iters * 8192;
      printf("mode 4 (DMA hit side) hit waves %d, HBM %.1f MiB/CU: launch %.1f us; hit side: %.0f ticks per CU for %.0f KiB -> %.2f B/tick/CU; HBM side %.2f TB/s over the launch; total %.1f GB/s per CU\n", hit_waves,
             hbm_bytes / 1048576.0, ms * 1e3, avg, hit_bytes / 1024, hit_bytes / avg, 256 * hbm_bytes / ms / 1e9, (hit_bytes + hbm_bytes) / ms / 1e6);
    }
  }
  return 0;
}
