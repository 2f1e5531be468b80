// Does the ADDRESS PATTERN of dl_linear_packed's weight stream cost bandwidth?  (round 5, follow-up to profiles/r03_lds_dma_raw.txt: one LDS-DMA wave per CU
// streams contiguous 8 KiB groups at 7.3 TB/s, yet the packed kernel's two loader waves alone reach 5.0-5.4.)
// One workgroup per CU, 2 loader waves + 4 idle waves, per 64-k step every loader issues NU pieces of 1 KiB (global_load_lds_dwordx4 nt, M0 set per piece exactly
// like the product's lp_dma_piece), waits until at most (RD - 3) NU pieces are outstanding; optionally one __syncthreads per step (BAR).
//   pattern 0  unit-major (the product's layout): unit i of the set is its own stream, S KiB apart: piece (i, h) of step t at  i * S KiB + t * 2 KiB + h KiB
//   pattern 1  set-major: the workgroup's whole stream contiguous:                                 piece (i, h) of step t at  t * 2 NU KiB + i * 2 KiB + h KiB
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_stream_pattern.hip -o tools/_lds_dma_stream_pattern && tools/_lds_dma_stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DL_LDS __attribute__((address_space(3)))
#define DL_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ void dma_piece(const DL_GLOBAL void* s_base, uint32_t v_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(v_off), "s"(s_base), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NU, int RD, int PATTERN, bool BAR>
__global__ __launch_bounds__(384) void stream(const char* src, int S, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int steps = S / 2;
  const DL_GLOBAL char* wg = (const DL_GLOBAL char*)src + (int64_t)blockIdx.x * NU * S * 1024;
  if (w >= 2) {
    if (BAR) for (int t = 0; t <= steps; ++t) __syncthreads();
    return;
  }
  const int h = w;
  const DL_GLOBAL char* base[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) base[i] = PATTERN == 0 ? wg + ((int64_t)i * S + h) * 1024 : wg + (int64_t)(2 * i + h) * 1024;
  const uint32_t ring = (uint32_t)(uintptr_t)(DL_LDS unsigned char*)smem + h * 1024u;
  uint32_t voff = lane * 16u;
  constexpr uint32_t kStep = PATTERN == 0 ? 2048u : NU * 2048u;
  auto issue = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NU; ++i) dma_piece(base[i], voff, ring + (uint32_t)(slot * NU * 2048 + i * 2048));
    voff += kStep;
  };
  int slot = 0, issued = 0;
  for (; issued < RD - 1 && issued < steps; ++issued) { issue(slot); slot = slot + 1 == RD ? 0 : slot + 1; }
  for (int t = 0; t < steps; ++t) {
    if (issued - 2 - t >= RD - 3) wait_vmcnt<(RD - 3) * NU>(); else wait_vmcnt<0>();
    if (BAR) __syncthreads();
    if (issued < steps) { issue(slot); slot = slot + 1 == RD ? 0 : slot + 1; ++issued; }
  }
  if (BAR) __syncthreads();
  wait_vmcnt<0>();
  if (sink && lane == 0 && blockIdx.x == 100000) sink[0] = ((DL_LDS unsigned*)smem)[7];
}

template <typename K>
static void run(const char* name, K kfn, const char* src, int NU, int RD, int S, int G) {
  const size_t smem = (size_t)RD * NU * 2048;
  hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long long per_launch = (long long)G * NU * S * 1024;
  const int copies = (int)((3ll << 30) / per_launch);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < copies; ++i) hipLaunchKernelGGL(kfn, dim3(G), dim3(384), smem, 0, src + (long long)i * per_launch, S, (unsigned*)nullptr);
  hipEventRecord(a);
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < copies; ++i) hipLaunchKernelGGL(kfn, dim3(G), dim3(384), smem, 0, src + (long long)i * per_launch, S, (unsigned*)nullptr);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / (4.0 * copies);
  printf("%-72s %5.1f MB per launch: %7.2f us  %5.2f TB/s\n", name, per_launch / 1e6, us, per_launch / us / 1e6);
}

int main() {
  char* src; hipMalloc(&src, 3ll << 30); hipMemset(src, 1, 3ll << 30); hipDeviceSynchronize();
  const int G = 256;
#define RUN(NU, RD, P, B, S) run("NU=" #NU " RD=" #RD " " #P " barrier=" #B " S=" #S, stream<NU, RD, P, B>, src, NU, RD, S, G);
  // gate|up-like: 6 units x K = 4096 (S = 128): 768 KiB per CU;  q|k|v-like: 3 units; down-like: 1 unit x K = 11008 / 4 ranges ~ 4 units x S = 86
  RUN(6, 8, 0, false, 128) RUN(6, 8, 1, false, 128) RUN(6, 8, 0, true, 128) RUN(6, 8, 1, true, 128)
  RUN(3, 12, 0, false, 128) RUN(3, 12, 1, false, 128) RUN(3, 12, 0, true, 128) RUN(3, 12, 1, true, 128)
  RUN(4, 10, 0, true, 86) RUN(4, 10, 1, true, 86)
  RUN(6, 8, 0, true, 2048) RUN(6, 8, 1, true, 2048)
  return 0;
}
