// Can the ramp of decode launch n + 1 hide under the tail of launch n?  (one 1 GiB dl_gemv launch reads at 6.9 TB/s, the decode step's 129 launches
// average 5.9 -- the difference is what every launch boundary costs: drain, ~1.5 us of dispatch, a cold HBM round trip.)
//
// Probe: a chain of batch-1 weight-streaming launches with the 7B decoder's four shapes per layer (q|k|v 12288x4096, o 4096x4096, gate|up 22016x4096, down 4096x11008; weights
// rotate over 8 layers = 3.2 GB: cold), each launch's input the previous launch's output.
//   mode 0  one stream: launch n + 1 starts when launch n has drained (what a captured decode step does today)
//   mode 1  two streams, launches alternate: the STREAM order only says n + 2 after n; the true dependency n -> n + 1 is a device-side flag.  A workgroup of launch
//           n + 1 requests its first weight rows, THEN waits for launch n's flag (bounded spin), then reads x.  Launch n signals with a workgroup counter whose last
//           arrival publishes the flag; outputs are written through (sc1) and read past the caches (sc1), no fences.
// Every grid is 1024 workgroups of 256 threads holding 40 KiB of LDS (4 per CU, like the product's 128-VGPR GEMV kernels: the chip is full), so launch n + 1 only gets slots as
// launch n's workgroups retire -- and launch n is always fully dispatched before n + 1 is eligible (it became eligible when n - 1, same grid size, had retired completely).
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GI32 __attribute__((address_space(1))) int
#define GU32 __attribute__((address_space(1))) uint32_t

__device__ __forceinline__ float bf(uint32_t w, int hi) { return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16)); }

// y[n] = sum_k W[n][k] x[k] (bf16 weights, fp32 x and y), one row per wave per pass, K <= 11008
template <int OVERLAP, int DYN>
__global__ __launch_bounds__(256, 4) void stream_gemv(const uint16_t* __restrict__ W, const float* x, float* y, int N, int K, const int* wait_flag, int* counter, int* done_flag, int epoch,
                                                   int* err, unsigned long long* stamps, int done_tag) {
  __shared__ float xs_raw[10240];  // 40 KiB: four workgroups fill a CU's LDS; x kept as bf16 pairs in the first 22 KB
  uint16_t* xs = reinterpret_cast<uint16_t*>(xs_raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + w;
  if (threadIdx.x == 0) xs_raw[10239] = 0.f;
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();
  const int chunks = K / 512;  // 16-byte chunks per lane per row (K multiple of 512 or handled by the tail below)
  // first row's first chunks: requested before anything else
  int n = wave;
  u32x4 pre[4];
  const u32x4* row0 = reinterpret_cast<const u32x4*>(W + (size_t)(n < N ? n : 0) * K);
#pragma unroll
  for (int c = 0; c < 4; ++c) pre[c] = __builtin_nontemporal_load(row0 + (c < chunks ? c : 0) * 64 + lane);
  // DYN: rows past the first come from a ticket counter, one ticket requested a whole row ahead of its use
  int raw = 0;
  if (DYN && lane == 0) raw = __hip_atomic_fetch_add((GI32*)counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (OVERLAP == 1 && wait_flag) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load((const GI32*)wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) {
          atomicOr(err, 1);
          break;
        }
      }
    }
    __syncthreads();
  }
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 1] = wall_clock64();
  // x: past the caches (the previous launch may still be running on other CUs)
  if constexpr (OVERLAP != 2) {
    typedef __attribute__((address_space(1))) uint64_t GU64;
    uint64_t t[22];  // K / 2 / 256 <= 21.5: every request in flight before the first use
#pragma unroll
    for (int i = 0; i < 22; ++i) {
      const int idx = threadIdx.x + i * 256;
      t[i] = 0;
      if (idx * 2 < K) t[i] = OVERLAP ? __hip_atomic_load((const GU64*)(reinterpret_cast<const uint64_t*>(x) + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : reinterpret_cast<const uint64_t*>(x)[idx];
    }
#pragma unroll
    for (int i = 0; i < 22; ++i) {
      const int idx = threadIdx.x + i * 256;
      if (idx * 2 < K) reinterpret_cast<uint32_t*>(xs)[idx] = (uint32_t)((t[i] >> 16) & 0xffffu) | (uint32_t)((t[i] >> 32) & 0xffff0000u);
    }
  }
  if constexpr (OVERLAP == 2) {
    // granules: x[k] = {fp32 bits, tag of the launch that wrote it}; poll until every granule of this thread carries the producer's tag (wait_flag != NULL),
    // all requests in flight together, bounded
    typedef __attribute__((address_space(1))) uint64_t GU64;
    const uint64_t* xg = reinterpret_cast<const uint64_t*>(x);
    const uint32_t want = (uint32_t)epoch;
    int spins = 0;
    for (int c0 = 0; c0 < 44; c0 += 11) {  // 11 granules per thread and pass (registers), each pass polled until its granules carry the tag
      uint64_t t[11];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
          const int idx = threadIdx.x + (c0 + i) * 256;
          t[i] = (uint64_t)want << 32;
          if (idx < K) t[i] = __hip_atomic_load((const GU64*)(xg + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 11; ++i) ok = ok && (!wait_flag || (uint32_t)(t[i] >> 32) == want);
        if (ok) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 18)) {
          atomicOr(err, 2);
          break;
        }
      }
#pragma unroll
      for (int i = 0; i < 11; ++i) {
        const int idx = threadIdx.x + (c0 + i) * 256;
        if (idx < K) xs[idx] = (uint16_t)((uint32_t)t[i] >> 16);
      }
    }
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 2] = wall_clock64();
  bool first = true;
  while (n < N) {
    int n_next = n + n_waves;
    if (DYN) {
      n_next = n_waves + __builtin_amdgcn_readfirstlane(raw);
      if (lane == 0) raw = __hip_atomic_fetch_add((GI32*)counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u32x4* row = reinterpret_cast<const u32x4*>(W + (size_t)n * K);
    float acc = 0.f;
    for (int c0 = 0; c0 * 512 < K; c0 += 4) {
      u32x4 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cc = c0 + c;
        const bool ok = (cc * 512 + lane * 8) < K;
        v[c] = (c0 == 0 && first) ? pre[c] : (ok ? __builtin_nontemporal_load(row + cc * 64 + lane) : u32x4{0, 0, 0, 0});
        if (!ok) v[c] = u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k0 = (c0 + c) * 512 + lane * 8;
        if (k0 < K) {
          const uint32_t ww[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc += bf(ww[e], 0) * __uint_as_float((uint32_t)xs[k0 + 2 * e] << 16) + bf(ww[e], 1) * __uint_as_float((uint32_t)xs[k0 + 2 * e + 1] << 16);
        }
      }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
      const float r = acc * 0.02f;  // keep the chain's magnitude bounded
      if (OVERLAP == 2) {
        typedef __attribute__((address_space(1))) uint64_t GU64;
        __hip_atomic_store((GU64*)(reinterpret_cast<uint64_t*>(y) + n), ((uint64_t)(uint32_t)done_tag << 32) | __float_as_uint(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (OVERLAP)
        __hip_atomic_store((GU32*)(y + n), __float_as_uint(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        y[n] = r;
    }
    n = n_next;
    first = false;
  }
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 3] = wall_clock64();
  if (OVERLAP == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's outputs are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      const int old = __hip_atomic_fetch_add((GI32*)counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (int)gridDim.x - 1) {
        __hip_atomic_store((GI32*)counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((GI32*)done_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int H = 4096, I = 11008, L = 8;
  struct Shape { int N, K; } shp[4] = {{3 * H, H}, {H, H}, {2 * I, H}, {H, I}};
  std::vector<uint16_t*> W(L * 4);
  size_t total = 0;
  for (int l = 0; l < L; ++l)
    for (int s = 0; s < 4; ++s) {
      const size_t n = (size_t)shp[s].N * shp[s].K;
      CK(hipMalloc(&W[l * 4 + s], n * 2));
      std::vector<uint16_t> h(n);
      uint32_t st = 12345u + l * 4 + s;
      for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3c00u + ((st >> 20) & 0x3ffu)) ^ (uint16_t)((st >> 8) & 0x8000u); }  // +-[0.0078, 0.0156)
      CK(hipMemcpy(W[l * 4 + s], h.data(), n * 2, hipMemcpyHostToDevice));
      total += n * 2;
    }
  const int n_k = L * 4;
  float* buf[4][5];
  for (int m = 0; m < 4; ++m)
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&buf[m][i], 22016 * 8)); CK(hipMemset(buf[m][i], 0, 22016 * 8)); }
  std::vector<float> x0(22016);
  for (int i = 0; i < 22016; ++i) x0[i] = std::sin(0.37f * i);
  int *flags, *counters, *err;
  CK(hipMalloc(&flags, n_k * 64 * 4)); CK(hipMalloc(&counters, n_k * 64 * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(flags, 0, n_k * 64 * 4)); CK(hipMemset(counters, 0, n_k * 64 * 4)); CK(hipMemset(err, 0, 4));
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t ea, eb, fork, join;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  // chain: input of launch j = output of launch j - 1 (buffers rotate over 5); flags[j * 16]: launch j has finished in this epoch
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)n_k * 4096 * 8));
  bool stamp_now = false;
  auto chain = [&](int mode, int epoch) {
    if (mode == 2) hipMemsetAsync(counters, 0, n_k * 64 * 4, s0);
    for (int j = 0; j < n_k; ++j) {
      unsigned long long* st_ptr = stamp_now ? stamps + (size_t)j * 4096 : nullptr;
      const Shape sh = shp[j & 3];
      float* in = buf[mode][j % 5];
      float* out = buf[mode][(j + 1) % 5];
      hipStream_t st = ((mode == 1 || mode == 3) && (j & 1)) ? s1 : s0;
      const int* wf = (mode == 1 && j > 0) ? flags + (j - 1) * 16 : nullptr;
      if (mode == 3)  // granules: wanted tag = the producer's (launch j - 1 of this chain), own tag on the outputs
        hipLaunchKernelGGL((stream_gemv<2, 0>), dim3(1024), dim3(256), 0, st, W[j], in, out, sh.N, sh.K, j > 0 ? flags : (const int*)nullptr, (int*)nullptr, (int*)nullptr, epoch * 64 + j - 1, err,
                           st_ptr, epoch * 64 + j);
      else if (mode == 1)
        hipLaunchKernelGGL((stream_gemv<1, 0>), dim3(1024), dim3(256), 0, st, W[j], in, out, sh.N, sh.K, wf, counters + j * 16, flags + j * 16, epoch, err, st_ptr, 0);
      else if (mode == 2)
        hipLaunchKernelGGL((stream_gemv<0, 1>), dim3(1024), dim3(256), 0, st, W[j], in, out, sh.N, sh.K, (const int*)nullptr, counters + j * 16, (int*)nullptr, epoch, err, st_ptr, 0);
      else
        hipLaunchKernelGGL((stream_gemv<0, 0>), dim3(1024), dim3(256), 0, st, W[j], in, out, sh.N, sh.K, (const int*)nullptr, (int*)nullptr, (int*)nullptr, epoch, err, st_ptr, 0);
    }
  };
  int epoch = 0;
  for (int mode = 0; mode < 4; ++mode) {
    if (mode == 3) {
      std::vector<uint64_t> g0(22016);
      for (int i = 0; i < 22016; ++i) { uint32_t b; memcpy(&b, &x0[i], 4); g0[i] = b; }
      CK(hipMemcpy(buf[3][0], g0.data(), 22016 * 8, hipMemcpyHostToDevice));
    } else
      CK(hipMemcpy(buf[mode][0], x0.data(), 22016 * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {  // warm
      ++epoch;
      if (mode == 1 || mode == 3) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
      chain(mode, epoch);
      if (mode == 1 || mode == 3) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
    }
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(ea, s0));
    for (int rep = 0; rep < reps; ++rep) {
      ++epoch;
      if (mode == 1 || mode == 3) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
      chain(mode, epoch);
      if (mode == 1 || mode == 3) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
    }
    CK(hipEventRecord(eb, s0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    const double us_per_launch = ms * 1e3 / (reps * n_k);
    {
      std::vector<unsigned long long> h((size_t)n_k * 4096);
      stamp_now = true; ++epoch;
      if (mode == 1 || mode == 3) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
      chain(mode, epoch);
      if (mode == 1 || mode == 3) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
      CK(hipDeviceSynchronize());
      stamp_now = false;
      CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (int g = 0; g < 1024; ++g) t0 = h[8 * 4096 + g * 4] < t0 ? h[8 * 4096 + g * 4] : t0;
      printf("  launch:  wg start first/last | flag passed first/last | x in LDS first/last | wg done first/last   [us, 100 MHz clock, relative to launch 8's first workgroup]\n");
      for (int j = 8; j < 16; ++j) {
        printf("  %2d (%5dx%5d):", j, shp[j & 3].N, shp[j & 3].K);
        for (int q = 0; q < 4; ++q) {
          unsigned long long lo = ~0ull, hi = 0;
          for (int g = 0; g < 1024; ++g) { const unsigned long long t = h[(size_t)j * 4096 + g * 4 + q]; lo = t < lo ? t : lo; hi = t > hi ? t : hi; }
          printf("  %7.2f %7.2f", (double)(long long)(lo - t0) * 0.01, (double)(long long)(hi - t0) * 0.01);
        }
        printf("\n");
        if (mode != 1 && (j == 10 || j == 11)) {  // who finishes late?  (workgroup g runs on XCD g % 8)
          printf("      stream time (x in LDS -> done) per XCD, mean [min..max] us:");
          for (int xc = 0; xc < 8; ++xc) {
            double sum = 0, lo = 1e9, hi = 0;
            for (int g = xc; g < 1024; g += 8) {
              const double d = (double)(long long)(h[(size_t)j * 4096 + g * 4 + 3] - h[(size_t)j * 4096 + g * 4 + 2]) * 0.01;
              sum += d; lo = d < lo ? d : lo; hi = d > hi ? d : hi;
            }
            printf("  %.1f [%.1f..%.1f]", sum / 128, lo, hi);
          }
          printf("\n");
        }
      }
    }
    printf("mode %d (%s): %.1f us per chain of %d launches, %.2f us per launch, %.2f TB/s over %.2f GB\n", mode, mode == 3 ? "two streams + tagged 8-byte granules" : mode == 1 ? "two streams + device-side flags" : mode == 2 ? "one stream, rows from a ticket counter" : "one stream", ms * 1e3 / reps,
           n_k, us_per_launch, total * reps / (ms * 1e-3) / 1e12, total / 1e9);
  }
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  // both modes ran 13 chains from the same start: the rotating buffers must hold the same values
  std::vector<float> a(22016), b(22016);
  double maxd = 0, maxa = 0;
  for (int i = 0; i < 5; ++i) {
    CK(hipMemcpy(a.data(), buf[0][i], 22016 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), buf[1][i], 22016 * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < 22016; ++k) { maxd = std::fmax(maxd, std::fabs((double)a[k] - b[k])); maxa = std::fmax(maxa, std::fabs((double)a[k])); }
  }
  for (int i = 0; i < 5; ++i) {
    CK(hipMemcpy(a.data(), buf[0][i], 22016 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), buf[2][i], 22016 * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < 22016; ++k) maxd = std::fmax(maxd, std::fabs((double)a[k] - b[k]));
  }
  {
    std::vector<uint64_t> gb(22016);
    for (int i = 0; i < 5; ++i) {
      CK(hipMemcpy(a.data(), buf[0][i], 22016 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), buf[3][i], 22016 * 8, hipMemcpyDeviceToHost));
      for (int k = 0; k < 22016; ++k) { float v; uint32_t bits = (uint32_t)gb[k]; memcpy(&v, &bits, 4); maxd = std::fmax(maxd, std::fabs((double)a[k] - v)); }
    }
  }
  printf("error flag %d; max |serial - overlapped| over the chain's buffers %.3g (max |value| %.3g)\n", herr, maxd, maxa);
  return 0;
}
