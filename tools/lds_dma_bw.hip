// Raw LDS-DMA streaming rate of ONE loader wave per CU (gfx950): 8-piece groups into a 128 KiB ring, counted vmcnt waits, no consumers.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_lds_dma_bw tools/lds_dma_bw.hip && tools/_lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define DL_LDS __attribute__((address_space(3)))
#define DL_GLOBAL __attribute__((address_space(1)))

template <int LAG, bool NT, int WAVES, int CTL = 0>
__global__ __launch_bounds__(256, 1) void stream(const char* src, long long bytes_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  DL_LDS volatile int* ctl = (DL_LDS volatile int*)(smem + 128 * 1024);
  if (threadIdx.x < 8) ctl[threadIdx.x] = 0;
  __syncthreads();
  if (wid >= WAVES) {
    if (CTL & 4) {  // pollers: the other three waves spin on the landed word like the block's consumers
      for (int i = 0; i < 200000; ++i) {
        if (__builtin_amdgcn_readfirstlane(ctl[0]) >= (int)(bytes_per_wg / 1024)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    return;
  }
  int acc_flag = 0;
  const unsigned ring = (unsigned)(uintptr_t)(DL_LDS unsigned char*)smem + wid * (128 * 1024 / WAVES);
  const char* base = src + (long long)blockIdx.x * bytes_per_wg + (long long)wid * (bytes_per_wg / WAVES);
  const long long n8 = bytes_per_wg / WAVES / 8192;
  unsigned slot = 0;
  const unsigned nslot = 128 / WAVES / 8;
  for (long long i = 0; i < n8; ++i) {
    const unsigned long long ra = (unsigned long long)(base + i * 8192);
    const DL_GLOBAL char* row = (const DL_GLOBAL char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ra >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ra));
    unsigned keep, off2, voff = lane * 16;
    unsigned dst = __builtin_amdgcn_readfirstlane(ring + slot * 8192);
    if (NT)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_add_u32 %1, 0x1000, %2\n\t"
                   "global_load_lds_dwordx4 %2, %3 nt\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024 nt\n\tglobal_load_lds_dwordx4 %2, %3 offset:2048 nt\n\tglobal_load_lds_dwordx4 %2, %3 offset:3072 nt\n\t"
                   "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %3 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
                   "s_mov_b32 m0, %0" : "=&s"(keep), "=&v"(off2) : "v"(voff), "s"(row), "s"(dst) : "memory", "scc");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_add_u32 %1, 0x1000, %2\n\t"
                   "global_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\tglobal_load_lds_dwordx4 %2, %3 offset:2048\n\tglobal_load_lds_dwordx4 %2, %3 offset:3072\n\t"
                   "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                   "s_mov_b32 m0, %0" : "=&s"(keep), "=&v"(off2) : "v"(voff), "s"(row), "s"(dst) : "memory", "scc");
    slot = slot + 1 == nslot ? 0 : slot + 1;
    if (LAG == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (LAG == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (LAG == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    if (LAG == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    if (CTL & 1) acc_flag += __builtin_amdgcn_readfirstlane(ctl[1]);  // ds_read + s_waitcnt lgkmcnt(0)
    if ((CTL & 2) && lane == 0) ctl[0] = (int)(i * 8) - LAG;      // ds_write
  }
  if (CTL && lane == 0) ctl[0] = (int)(bytes_per_wg / 1024) + acc_flag;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lane == 0 && blockIdx.x == 100000) sink[0] = ((DL_LDS unsigned*)smem)[7];
}

__global__ void fill_random(unsigned* p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = (x & 0xbfffbfffu) | 0x30003000u;  // bf16-like pairs of moderate magnitude
  }
}

template <typename K>
static void run(const char* name, K kfn, const char* src, long long per_wg, int G) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kfn, dim3(G), dim3(256), 128 * 1024 + 64, 0, src, per_wg, (unsigned*)nullptr);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kfn, dim3(G), dim3(256), 128 * 1024 + 64, 0, src, per_wg, (unsigned*)nullptr);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-34s %7.1f us per launch  %6.2f TB/s  (%5.1f GB/s per CU)\n", name, ms * 200.0, (double)per_wg * G * 5 / ms / 1e9, (double)per_wg * 5 / ms / 1e6);
}

template <typename K>
static void run_small(const char* name, K kfn, const char* src, long long per_wg, int G, int copies) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < copies; ++i) hipLaunchKernelGGL(kfn, dim3(G), dim3(256), 128 * 1024 + 64, 0, src + (long long)i * per_wg * G, per_wg, (unsigned*)nullptr);
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r)
    for (int i = 0; i < copies; ++i) hipLaunchKernelGGL(kfn, dim3(G), dim3(256), 128 * 1024 + 64, 0, src + (long long)i * per_wg * G, per_wg, (unsigned*)nullptr);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / (10.0 * copies);
  printf("%-34s %4lld KiB per CU: %7.2f us per launch (eager, back to back)  %6.2f TB/s\n", name, per_wg / 1024, us, (double)per_wg * G / us / 1e6);
}

int main() {
  const int G = 256;
  const long long per_wg = 4ll << 20;  // 4 MiB per CU = 1 GiB per launch
  char* src;
  hipMalloc(&src, per_wg * G);
  if (getenv("RANDOM_DATA")) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)src, per_wg * G / 4);
  else hipMemset(src, 1, per_wg * G);
  hipDeviceSynchronize();
  printf("data: %s\n", getenv("RANDOM_DATA") ? "pseudo-random bf16-like" : "constant bytes");
  hipFuncSetAttribute((const void*)stream<32, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define RUN(LAG, NT, W)                                                                                                  \
  hipFuncSetAttribute((const void*)stream<LAG, NT, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);          \
  run("lag " #LAG " nt=" #NT " loader waves=" #W, stream<LAG, NT, W>, src, per_wg, G);
#define RUNC(LAG, C)                                                                                                    \
  hipFuncSetAttribute((const void*)stream<LAG, true, 1, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
  run("lag " #LAG " nt, control traffic mask " #C, stream<LAG, true, 1, C>, src, per_wg, G);
  RUNC(32, 1) RUNC(32, 2) RUNC(32, 3) RUNC(32, 7)
  RUN(8, true, 1) RUN(16, true, 1) RUN(32, true, 1) RUN(48, true, 1) RUN(32, false, 1) RUN(48, false, 1)
  RUN(16, true, 2) RUN(32, true, 2) RUN(16, true, 4) RUN(32, true, 4)
  for (long long kb : {128ll, 344ll, 384ll, 688ll, 2048ll}) {
    const int copies = (int)((1ll << 30) / (kb * 1024 * G));
    run_small("lag 32 nt, 1 loader wave", stream<32, true, 1>, src, kb * 1024, G, copies < 1 ? 1 : copies);
  }
  return 0;
}
