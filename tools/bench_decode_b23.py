#!/usr/bin/env python
"""Where does dl_gemv hand over to dl_gemm_smallm?  Decode ms/step at B = 2 / 3 on either path (round 4, final tree:
B=2 3.187 vs 3.521, B=3 3.495 vs 3.559, B=4 on dl_gemm_smallm 3.579: the hand-over after 3 rows stands)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench
cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
feats = model.encode_images(images)
def run(B, n_new):
    ids = prompt.expand(B, -1).contiguous(); f = feats.expand(B, -1, -1).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.generate(ids, image_features=f, max_new_tokens=n_new, eos_token_id=None)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for B in (2, 3, 4):
    for gmax in (3, 1):
        if B == 4 and gmax == 1: continue
        model.gemv_max_decode_batch = gmax
        model._dstate = None
        for _ in range(2):
            run(B, 33); run(B, 1)
        t = min(run(B, 65) for _ in range(3)) - min(run(B, 1) for _ in range(3))
        print(f"B={B} gemv_max_decode_batch={gmax}: {t / 64 * 1e3:6.3f} ms/step", flush=True)
