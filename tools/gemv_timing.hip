// Phases of dl_gemv's workgroup 0 (100 MHz wall clock) + event time of the launch, cold weights.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDL_GEMV_TIMING -I include -I dynamic_llava_amd/csrc tools/gemv_timing.hip -o tools/_gemv_timing
#include "../dynamic_llava_amd/csrc/capi.hip"
#include "../dynamic_llava_amd/csrc/gemv.hip"

#include <vector>

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096, mode = argc > 3 ? atoi(argv[3]) : 0;
  const int NBUF = 4;
  std::vector<void*> ws(NBUF);
  const size_t wbytes = (size_t)N * K * 2;
  for (auto& w : ws) {
    hipMalloc(&w, wbytes);
    hipMemset(w, 0x3c, wbytes);
  }
  void *x, *h, *h2, *dl, *nw, *y, *flush;
  hipMalloc(&x, (size_t)2 * K * 2); hipMemset(x, 0x3c, (size_t)2 * K * 2);
  hipMalloc(&h, K * 2); hipMemset(h, 0x3c, K * 2);
  hipMalloc(&h2, K * 2);
  hipMalloc(&dl, K * 2); hipMemset(dl, 0x3c, K * 2);
  hipMalloc(&nw, K * 2); hipMemset(nw, 0x3c, K * 2);
  hipMalloc(&y, (size_t)N * 2);
  const size_t flush_bytes = 512ull << 20;
  hipMalloc(&flush, flush_bytes);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 10; ++it) {
    hipMemsetAsync(flush, it, flush_bytes, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    int rc = dl_gemv(mode, ws[it % NBUF], N, K, x, K, h, h2, dl, nw, 1e-5f, y, N, 1, DL_BF16, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    if (rc) {
      printf("error: %s\n", dl_last_error());
      return 1;
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long s[8];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(dl::g_gemv_stamps), sizeof(s));
    if (it >= 7)
      printf("N=%d K=%d mode=%d: event %.2f us | entry->x in LDS %.2f | ->first group streamed %.2f | ->end %.2f us (wg 0)\n", N, K, mode, ms * 1e3,
             (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01);
  }
  return 0;
}
