"""Long generations across decode paths (dl_gemv B=1/3, dl_gemm_smallm B=8/16, library GEMM B=32): hipGraph replay must equal eager
launches token for token; reports where identical rows of a batch leave each other (hipBLASLt prefill GEMMs are not row-position
invariant) and the peak memory.
    python tools/soak.py"""
import sys, time
sys.path.insert(0, ".")
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench
cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
feats = model.encode_images(images)
for B, n in ((1, 1500), (3, 300), (8, 300), (16, 200), (32, 100)):
    ids = prompt.expand(B, -1).contiguous(); f = feats.expand(B, -1, -1).contiguous()
    model.use_hip_graph = True
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = model.generate(ids, image_features=f, max_new_tokens=n, eos_token_id=None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lens = [t.clone() for t in model.last_cache[1]]
    model.use_hip_graph = False
    # (round 4: the decode schedule -- split-KV factor, attention workgroups per head of the fused launch -- follows the lengths observed chunk by
    # chunk, by a deterministic rule: the same request replays the same kernels with and without hipGraph)
    b = model.generate(ids, image_features=f, max_new_tokens=n, eos_token_id=None)
    same = torch.equal(a, b)
    first = [int((a[0] != a[i]).nonzero()[0]) if not torch.equal(a[0], a[i]) else -1 for i in range(B)]
    pl = model.last_prefill_logits.float()
    print(f"B={B} n={n}: {(t1 - t0) * 1e3:.1f} ms, graph==eager: {same}, first step where row i leaves row 0: {first}, prefill logit max diff between rows: {float((pl - pl[0:1]).abs().max()):.4f}, kv_len full/sparse: {int(lens[0][0])}/{int(lens[-1][0])}", flush=True)
    assert same
print("soak OK, peak memory GB:", round(torch.cuda.max_memory_allocated() / 1e9, 1))
