// Register-only probe for the anomaly tools/tp_race_probe.py isolated: under GPU sharing with another process's model workload, the low half
// of v_pk_fma_f32 (as hipcc's SLP vectoriser emits it for two-neuron dot products: src1 broadcast through op_sel / op_sel_hi) was seen to
// differ from the scalar v_fma_f32 chain.  No memory traffic inside the timed loop: every wave runs a long chain of packed FMAs on values
// derived from its lane id, the same chain with scalar FMAs, and counts bitwise mismatches per form.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_pkfma_probe tools/pkfma_probe.hip && tools/_pkfma_probe [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float float2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void probe(unsigned long long* bad, int iters, int with_lds) {
  __shared__ float xs[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) xs[i] = 0.001f * (float)((i * 37 + blockIdx.x) % 997) - 0.4f;
  __syncthreads();
  float w0 = 0.25f + 0.001f * tid, w1 = -0.5f + 0.002f * tid;
  unsigned long long n_plain = 0, n_sel = 0, n_selhi = 0, n_mul = 0, n_add = 0, n_s0 = 0;
  for (int it = 0; it < iters; ++it) {
    float2_t acc_a = {0.f, 0.f}, acc_b = {0.f, 0.f}, acc_c = {0.f, 0.f};
    float r0 = 0.f, r1 = 0.f, s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
    float2_t pm = {1.f, 1.f}, pa = {0.f, 0.f}, acc_d = {0.f, 0.f};
    float m0 = 1.f, m1 = 1.f, a0 = 0.f, a1 = 0.f, u0 = 0.f, u1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 256; k += 2) {
      float x0, x1;
      if (with_lds) {
        x0 = xs[(tid * 8 + k) & 2047];
        x1 = xs[(tid * 8 + k + 1) & 2047];
      } else {
        x0 = 0.01f * (float)(k + it % 7) - 0.3f;
        x1 = 0.02f * (float)(k + 1) - 0.7f;
      }
      float2_t w = {w0 + 0.001f * k, w1 - 0.001f * k};
      float2_t x = {x0, x1};
      // (a) plain packed FMA: acc += w * x (lane-wise)
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc_a) : "v"(w), "v"(x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w.x), "v"(x0));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w.y), "v"(x1));
      // (b) op_sel:[0,1,0]: both halves multiply by the HIGH half of src1 (x1)
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc_b) : "v"(w), "v"(x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(w.x), "v"(x1));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(w.y), "v"(x1));
      // (c) op_sel_hi:[1,0,1]: both halves multiply by the LOW half of src1 (x0)
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc_c) : "v"(w), "v"(x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t0) : "v"(w.x), "v"(x0));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t1) : "v"(w.y), "v"(x0));
      // (d) v_pk_mul_f32 op_sel:[0,1]: both halves multiply by x1 (values near 1 so that the product stays finite)
      float2_t xm = {1.0f + 1e-3f * x0, 1.0f + 1e-3f * x1};
      asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(pm) : "v"(xm));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(m0) : "v"(xm.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(m1) : "v"(xm.y));
      // (e) v_pk_add_f32 op_sel:[0,1]
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(pa) : "v"(x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(x1));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(x1));
      // (f) v_pk_fma_f32 op_sel:[1,0,0]: the low half takes src0's HIGH half
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc_d) : "v"(w), "v"(x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(u0) : "v"(w.y), "v"(x0));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(u1) : "v"(w.y), "v"(x1));
    }
    n_mul += (__float_as_uint(pm.x) != __float_as_uint(m0)) + 2 * (__float_as_uint(pm.y) != __float_as_uint(m1));
    n_add += (__float_as_uint(pa.x) != __float_as_uint(a0)) + 2 * (__float_as_uint(pa.y) != __float_as_uint(a1));
    n_s0 += (__float_as_uint(acc_d.x) != __float_as_uint(u0)) + 2 * (__float_as_uint(acc_d.y) != __float_as_uint(u1));
    n_plain += (__float_as_uint(acc_a.x) != __float_as_uint(r0)) + 2 * (__float_as_uint(acc_a.y) != __float_as_uint(r1));
    n_sel += (__float_as_uint(acc_b.x) != __float_as_uint(s0)) + 2 * (__float_as_uint(acc_b.y) != __float_as_uint(s1));
    n_selhi += (__float_as_uint(acc_c.x) != __float_as_uint(t0)) + 2 * (__float_as_uint(acc_c.y) != __float_as_uint(t1));
    w0 += 1e-4f;
  }
  if (n_plain) atomicAdd(&bad[0], n_plain & 1 ? 1ull : 0ull), atomicAdd(&bad[1], n_plain >> 1 ? 1ull : 0ull);
  if (n_sel) atomicAdd(&bad[2], n_sel & 1 ? 1ull : 0ull), atomicAdd(&bad[3], n_sel >> 1 ? 1ull : 0ull);
  if (n_selhi) atomicAdd(&bad[4], n_selhi & 1 ? 1ull : 0ull), atomicAdd(&bad[5], n_selhi >> 1 ? 1ull : 0ull);
  if (n_mul) atomicAdd(&bad[8], 1ull);
  if (n_add) atomicAdd(&bad[9], 1ull);
  if (n_s0) atomicAdd(&bad[10], 1ull);
  atomicAdd(&bad[6], 1ull);
}

// Aggressors (same process, second stream): small kernels that co-reside with the probe's waves on the same SIMDs.
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ __launch_bounds__(256) void aggr_mfma(float* out, int iters) {  // v_mfma_f32_16x16x32_bf16 chain
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) a[i] = (short)(0x3f80 + threadIdx.x + i), b[i] = (short)(0x3f00 + i);
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  if (c[0] == 12345.f) out[threadIdx.x] = c[1];
}
__global__ __launch_bounds__(256) void aggr_dot2(float* out, int iters) {  // v_dot2c_f32_bf16 chain (what dl_gemv runs)
  float acc = 0.f;
  unsigned a = 0x3f803f80u + threadIdx.x, b = 0x3f003e80u;
  for (int i = 0; i < iters; ++i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
  if (acc == 12345.f) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void aggr_pkmul(float* out, int iters) {  // packed multiplies with source-select modifiers
  float2_t a = {1.0f + threadIdx.x, 2.0f}, b = {0.999f, 1.001f};
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a) : "v"(b));
  }
  if (a.x == 12345.f) out[threadIdx.x] = a.y;
}
__global__ __launch_bounds__(256) void aggr_valu(float* out, int iters) {  // plain VALU + transcendental
  float a = 1.0f + threadIdx.x;
  for (int i = 0; i < iters; ++i) a = __expf(a * 0.001f) + a * 0.5f;
  if (a == 12345.f) out[threadIdx.x] = a;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 10.0;
  const int aggr = argc > 2 ? atoi(argv[2]) : 0;
  hipStream_t s2;
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  float* sink;
  hipMalloc(&sink, 4096);
  unsigned long long* bad;
  hipMalloc(&bad, 16 * sizeof(*bad));
  hipMemset(bad, 0, 16 * sizeof(*bad));
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 50; ++i) {
      hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, bad, 8, i & 1);
      if (aggr == 1) hipLaunchKernelGGL(aggr_mfma, dim3(1024), dim3(256), 0, s2, sink, 4000);
      if (aggr == 2) hipLaunchKernelGGL(aggr_dot2, dim3(1024), dim3(256), 0, s2, sink, 20000);
      if (aggr == 3) hipLaunchKernelGGL(aggr_pkmul, dim3(1024), dim3(256), 0, s2, sink, 10000);
      if (aggr == 4) hipLaunchKernelGGL(aggr_valu, dim3(1024), dim3(256), 0, s2, sink, 5000);
      ++launches;
    }
    hipDeviceSynchronize();
  }
  unsigned long long h[16];
  hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  printf("{\"aggressor\": %d, \"launches\": %ld, \"threads_run\": %llu, \"threads_with_mismatch\": {\"plain_lo\": %llu, \"plain_hi\": %llu, \"op_sel_lo\": %llu, \"op_sel_hi_half\": %llu, "
         "\"op_sel_hi_lo\": %llu, \"op_sel_hi_hi\": %llu, \"pk_mul_op_sel\": %llu, \"pk_add_op_sel\": %llu, \"pk_fma_op_sel_src0\": %llu}}\n",
         aggr, launches, h[6], h[0], h[1], h[2], h[3], h[4], h[5], h[8], h[9], h[10]);
  return 0;
}
