// Where does the batch-1 decode attention spend its ~11 us?  Phase stamps of workgroup (0,0,0) + event time of the launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDL_ATTN_TIMING -I include -I dynamic_llava_amd/csrc tools/attn_timing.hip -o /tmp/attn_timing
#include "../dynamic_llava_amd/csrc/capi.hip"
#include "../dynamic_llava_amd/csrc/attn_decode.hip"

#include <vector>

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 226, ns = argc > 2 ? atoi(argv[2]) : 1, kif = argc > 3 ? atoi(argv[3]) : 256,
            chunk = argc > 4 ? atoi(argv[4]) : 256, warm = argc > 5 ? atoi(argv[5]) : 0;  // warm: 1 = same slab every launch, no flush (L2/MALL/TLB hits)
  const int nH = 32, d = 128, H = nH * d, T_cap = T + 8, NB = 8;
  std::vector<void*> ks(NB), vs(NB);
  const size_t slab = (size_t)nH * T_cap * d * 2;
  for (int i = 0; i < NB; ++i) {
    hipMalloc(&ks[i], slab);
    hipMalloc(&vs[i], slab);
    hipMemset(ks[i], 0x3c, slab);
    hipMemset(vs[i], 0x3c, slab);
  }
  void *qkv, *cosb, *sinb, *out, *ws, *flush;
  int32_t* lens;
  hipMalloc(&qkv, 3 * H * 2);
  hipMemset(qkv, 0x3c, 3 * H * 2);
  hipMalloc(&cosb, (size_t)4096 * d * 2);
  hipMalloc(&sinb, (size_t)4096 * d * 2);
  hipMemset(cosb, 0x3c, (size_t)4096 * d * 2);
  hipMemset(sinb, 0x3c, (size_t)4096 * d * 2);
  hipMalloc(&out, H * 2);
  hipMalloc(&ws, dl_attn_decode_workspace_bytes(1, nH, d, 64) + 16);
  hipMalloc(&lens, 4);
  const int32_t len = T - 1;
  hipMemcpy(lens, &len, 4, hipMemcpyHostToDevice);
  const size_t flush_bytes = 512ull << 20;
  hipMalloc(&flush, flush_bytes);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 12; ++it) {
    if (!warm) hipMemsetAsync(flush, it, flush_bytes, st);  // push K/V out of L2 and the Infinity Cache, as 13 GB of weights do per step
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    int rc = dl_attn_decode_rope(qkv, 3 * H, cosb, sinb, 4096, lens, lens, ks[warm ? 0 : it % NB], vs[warm ? 0 : it % NB], slab / 2, (int64_t)T_cap * d, T_cap, out, H, ws,
                                 ns, kif, chunk, 1, nH, nH, d, DL_BF16, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    if (rc) {
      printf("error: %s\n", dl_last_error());
      return 1;
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long s[8];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(dl::g_attn_stamps), sizeof(s));
    if (it >= 8)
      printf("%s T=%d ns=%d kif=%d chunk=%d: event %.2f us | entry->loads landed %.2f | compute %.2f | wg merge %.2f | store ack %.2f us\n", warm ? "warm" : "cold", T, ns, kif,
             chunk, ms * 1e3, (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01, (s[4] - s[3]) * 0.01);
  }
  return 0;
}
