// Phases of dl_attn_prefill's key-split kernel on the CLIP ViT-L/14-336 tower's shape (one image: 577 tokens, 16 heads x 64, non-causal), inputs cold.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDL_PF_TIMING -I include -I dynamic_llava_amd/csrc tools/pf_timing_clip.hip -o tools/_pf_timing_clip
#include "../dynamic_llava_amd/csrc/capi.hip"
#include "../dynamic_llava_amd/csrc/attn_prefill.hip"

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 577, nH = 16, d = 64, H = nH * d;
  void *qkv, *out, *flush;
  int32_t* cu;
  hipMalloc(&qkv, (size_t)T * 3 * H * 2);
  hipMemset(qkv, 0x3c, (size_t)T * 3 * H * 2);
  hipMalloc(&out, (size_t)T * H * 2);
  hipMalloc(&cu, 8);
  const int32_t h_cu[2] = {0, T};
  hipMemcpy(cu, h_cu, 8, hipMemcpyHostToDevice);
  const size_t flush_bytes = 512ull << 20;
  hipMalloc(&flush, flush_bytes);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* q = (const char*)qkv;
  for (int it = 0; it < 8; ++it) {
    if (!getenv("DL_PF_WARM")) hipMemsetAsync(flush, it, flush_bytes, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    int rc = dl_attn_prefill(q, q + (size_t)H * 2, q + (size_t)2 * H * 2, 3 * H, 3 * H, out, H, cu, 1, T, nH, nH, d, 0, DL_BF16, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    if (rc) {
      printf("error: %s\n", dl_last_error());
      return 1;
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long s[16];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(dl::g_pf_stamps), sizeof(s));
    if (it >= 5)
      printf("   whole-row kernel: K landed %.2f | V staged %.2f (since entry)\n", (s[8] - s[0]) * 0.01, (s[9] - s[0]) * 0.01);
    if (it >= 5 && getenv("DL_PF_WHOLE") && atoi(getenv("DL_PF_WHOLE")) == 0)
      printf("   whole-row kernel: K landed %.2f | V staged %.2f (since entry)\n", (s[8] - s[0]) * 0.01, (s[9] - s[0]) * 0.01);
    if (it >= 5 && getenv("DL_PF_WHOLE") && atoi(getenv("DL_PF_WHOLE")) == 0)
      printf("   round 1 (since the round's start = stamp 12 of round 0 is not taken; relative to entry): loads landed %.2f | compute done %.2f | barrier %.2f | stash %.2f | barrier %.2f us\n",
             (s[8] - s[0]) * 0.01, (s[9] - s[0]) * 0.01, (s[10] - s[0]) * 0.01, (s[11] - s[0]) * 0.01, (s[12] - s[0]) * 0.01);
    if (it >= 5)
      printf("T=%d: event %.2f us | last query tile of head 0, since its entry: first round staged %.2f | all rounds %.2f | partials exchanged %.2f | stores done %.2f us\n", T, ms * 1e3,
             (s[1] - s[0]) * 0.01, (s[2] - s[0]) * 0.01, (s[3] - s[0]) * 0.01, (s[7] - s[0]) * 0.01);
  }
  return 0;
}
