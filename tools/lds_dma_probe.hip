// What does global_load_lds_dwordx4 write where?  (gfx950)  Each lane's 16 source bytes carry (lane, dword) markers; the kernel dumps LDS.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_lds_dma_probe tools/lds_dma_probe.hip && tools/_lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define DL_LDS __attribute__((address_space(3)))
#define DL_GLOBAL __attribute__((address_space(1)))
__global__ void probe(const unsigned* src, unsigned* dump, int mode, int lds_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  DL_LDS unsigned* l = (DL_LDS unsigned*)smem;
  const int lane = threadIdx.x;
  for (int i = lane; i < 16384; i += 64) l[i] = 0xdead0000u | i;
  __syncthreads();
  const DL_GLOBAL char* g = (const DL_GLOBAL char*)src + lane * 16;
  unsigned dst = (unsigned)(uintptr_t)(DL_LDS unsigned char*)smem + lds_off;
  dst = __builtin_amdgcn_readfirstlane(dst);
  if (mode == 0) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  } else if (mode == 1) {
    __builtin_amdgcn_global_load_lds((const DL_GLOBAL void*)g, (DL_LDS void*)(smem + lds_off), 16, 0, 0);
  } else if (mode == 2) {  // saddr form + offset, m0 bumped as dma_8pieces does
    unsigned keep, off2;
    unsigned voff = lane * 16;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_add_u32 %1, 0x1000, %2\n\t"
        "global_load_lds_dwordx4 %2, %3 nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:1024 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&v"(off2) : "v"(voff), "s"(src), "s"(dst) : "memory", "scc");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 16384; i += 64) dump[i] = l[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = ((i / 4) << 8) | (i % 4);  // (16-byte chunk index << 8) | dword
  unsigned *src, *dump;
  hipMalloc(&src, 16384); hipMalloc(&dump, 65536);
  hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
  std::vector<unsigned> d(16384);
  for (int mode = 0; mode < 3; ++mode)
    for (int off : {0, 4096, 40000}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 65536, 0, src, dump, mode, off);
      hipMemcpy(d.data(), dump, 65536, hipMemcpyDeviceToHost);
      int n = 0, first = -1, last = -1;
      for (int i = 0; i < 16384; ++i) if (d[i] != (0xdead0000u | i)) { if (first < 0) first = i; last = i; ++n; }
      printf("mode %d lds_off %d: %d words changed, first word %d last %d;", mode, off, n, first, last);
      if (first >= 0) { printf(" words[first..+8]:"); for (int k = 0; k < 8; ++k) printf(" %x", d[first + k]); printf(" ... word[first+256..+4]:"); for (int k = 256; k < 260; ++k) printf(" %x", d[first + k]); }
      printf("\n");
    }
  return 0;
}
