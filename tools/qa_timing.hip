// Timeline of ONE dl_gemv_qkv_attn launch (the product's q|k|v + attention launch of a batch-1 decode layer), cold weights and K/V:
// per-workgroup wall-clock stamps (100 MHz) -> when the streaming workgroups start / end, when q arrives at the attention workgroups, when the slab keys
// are merged, when the heads are done; next to the event time of the launch and of the plain q|k|v dl_gemv on the same weights.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DDL_QA_TIMING -I include -I dynamic_llava_amd/csrc tools/qa_timing.hip -o tools/_qa_timing
#include "../dynamic_llava_amd/csrc/capi.hip"
#include "../dynamic_llava_amd/csrc/gemv.hip"

#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 200, nH = argc > 2 ? atoi(argv[2]) : 32, D = 128, NS = argc > 3 ? atoi(argv[3]) : 1;
  const int H = nH * D, N = 3 * H, K = H, T_cap = T + 16, NBUF = 4;
  std::vector<void*> ws(NBUF), ks(NBUF), vs(NBUF);
  for (int i = 0; i < NBUF; ++i) {
    hipMalloc(&ws[i], (size_t)N * K * 2); hipMemset(ws[i], 0x3c, (size_t)N * K * 2);
    hipMalloc(&ks[i], (size_t)nH * T_cap * D * 2); hipMemset(ks[i], 0x3c, (size_t)nH * T_cap * D * 2);
    hipMalloc(&vs[i], (size_t)nH * T_cap * D * 2); hipMemset(vs[i], 0x3c, (size_t)nH * T_cap * D * 2);
  }
  void *h, *h2, *dl, *nw, *qkv, *out, *cs, *sn, *gran, *flush;
  int32_t *lens, *err;
  hipMalloc(&h, K * 2); hipMemset(h, 0x3c, K * 2);
  hipMalloc(&h2, K * 2);
  hipMalloc(&dl, K * 2); hipMemset(dl, 0x3c, K * 2);
  hipMalloc(&nw, K * 2); hipMemset(nw, 0x3c, K * 2);
  hipMalloc(&qkv, N * 2); hipMalloc(&out, H * 2);
  hipMalloc(&cs, (size_t)(T_cap + 8) * D * 2); hipMemset(cs, 0x3c, (size_t)(T_cap + 8) * D * 2);
  hipMalloc(&sn, (size_t)(T_cap + 8) * D * 2); hipMemset(sn, 0x3c, (size_t)(T_cap + 8) * D * 2);
  hipMalloc(&gran, (size_t)dl_gemv_qkv_attn_workspace_bytes(nH, nH, D)); hipMemset(gran, 0, (size_t)dl_gemv_qkv_attn_workspace_bytes(nH, nH, D));
  hipMalloc(&lens, 4); hipMalloc(&err, 4); hipMemset(err, 0, 4);
  const int tl = T - 1;
  hipMemcpy(lens, &tl, 4, hipMemcpyHostToDevice);
  const size_t flush_bytes = 512ull << 20;
  hipMalloc(&flush, flush_bytes);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  for (int it = 0; it < 10; ++it) {
    hipMemsetAsync(flush, it, flush_bytes, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    int rc = dl_gemv_qkv_attn(ws[it % NBUF], K, h, h2, dl, nw, 1e-5f, qkv, cs, sn, T_cap + 8, lens, lens, ks[it % NBUF], vs[it % NBUF], (int64_t)nH * T_cap * D, (int64_t)T_cap * D, T_cap,
                              out, gran, it + 1, err, NS, nH, nH, D, DL_BF16, 0, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    if (rc) { printf("error: %s\n", dl_last_error()); return 1; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> s(1200 * 8);
    hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(dl::g_qa_stamps), s.size() * 8);
    hipMemsetAsync(flush, it + 50, flush_bytes, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    dl_gemv(DL_GEMV_ADDNORM, ws[(it + 1) % NBUF], N, K, nullptr, 0, h, h2, dl, nw, 1e-5f, qkv, N, 1, DL_BF16, 0, st);
    hipEventRecord(e2, st);
    hipStreamSynchronize(st);
    float ms2 = 0; hipEventElapsedTime(&ms2, e0, e2);
    if (it < 7) continue;
    const int grid = 1024, n_gemv = grid - nH * NS;
    long long t0 = s[0];
    for (int b = 0; b < grid; ++b) t0 = std::min(t0, s[b * 8]);
    auto us = [&](long long x) { return (x - t0) * 0.01; };
    std::vector<double> start, end;
    for (int b = 0; b < n_gemv; ++b) { start.push_back(us(s[b * 8])); end.push_back(us(s[b * 8 + 1])); }
    std::sort(start.begin(), start.end()); std::sort(end.begin(), end.end());
    double a0 = 1e9, a0m = 0, q = 0, qm = 1e9, slab = 0, done = 0, donem = 1e9;
    for (int b = n_gemv; b < grid; ++b) {
      a0 = std::min(a0, us(s[b * 8])); a0m = std::max(a0m, us(s[b * 8]));
      q = std::max(q, us(s[b * 8 + 1])); qm = std::min(qm, us(s[b * 8 + 1]));
      slab = std::max(slab, us(s[b * 8 + 2]));
      done = std::max(done, us(s[b * 8 + 3])); donem = std::min(donem, us(s[b * 8 + 3]));
    }
    {
      std::vector<double> qa, dn;
      for (int b = n_gemv; b < grid; ++b) { qa.push_back(us(s[b * 8 + 1])); dn.push_back(us(s[b * 8 + 3])); }
      printf("  per head: q arrived -> rotated -> keys done -> merged -> done:");
      for (int b = n_gemv; b < grid; b += 5) printf("  [%.1f %.1f %.1f %.1f %.1f]", us(s[b * 8 + 1]), us(s[b * 8 + 4]), us(s[b * 8 + 5]), us(s[b * 8 + 2]), us(s[b * 8 + 3]));
      printf("\n  q arrival by head index:");
      for (double x : qa) printf(" %.1f", x);
      printf("\n  first-pass end of streaming workgroups 0, 32, 64, ...:");
      for (int b = 0; b < n_gemv; b += 32) printf(" %.1f", us(s[b * 8 + 2]));
      printf("\n");
      std::sort(qa.begin(), qa.end()); std::sort(dn.begin(), dn.end());
      printf("  q arrival per head (sorted):");
      for (double x : qa) printf(" %.1f", x);
      printf("\n  head done (sorted):");
      for (double x : dn) printf(" %.1f", x);
      printf("\n  streaming workgroup end, deciles:");
      for (int i = 0; i <= 10; ++i) printf(" %.1f", end[std::min(end.size() - 1, end.size() * i / 10)]);
      printf("\n");
    }
    printf("T=%d heads=%d splits=%d: launch %.2f us by events (plain q|k|v dl_gemv: %.2f) | streaming workgroups start %.2f..%.2f (median %.2f), end median %.2f, p90 %.2f, last %.2f | attention workgroups start %.2f..%.2f, q arrived %.2f..%.2f, "
           "slab keys merged by %.2f, heads done %.2f..%.2f us\n", T, nH, NS, ms * 1e3, ms2 * 1e3, start.front(), start.back(), start[start.size() / 2], end[end.size() / 2], end[end.size() * 9 / 10], end.back(), a0, a0m, qm, q, slab, donem, done);
  }
  return 0;
}
