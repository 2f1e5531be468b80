#!/usr/bin/env python
"""Decode ms/step at small batch sizes: dl_gemv path (gemv_max_decode_batch >= B) vs the library-GEMM path.
    python tools/bench_decode_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench

cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
feats = model.encode_images(images)


def run(B, n_new):
    ids = prompt.expand(B, -1).contiguous()
    f = feats.expand(B, -1, -1).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.generate(ids, image_features=f, max_new_tokens=n_new, eos_token_id=None)
    torch.cuda.synchronize(); return time.perf_counter() - t0


for B in (4, 5, 8, 12, 16):
    for maxb in (4, 16):
        if B <= 4 and maxb == 16:
            continue
        model.smallm_max_decode_batch = maxb if maxb > 4 else 0
        model._dstate = None
        for _ in range(2):
            run(B, 33); run(B, 1)
        t = min(run(B, 65) for _ in range(3)) - min(run(B, 1) for _ in range(3))
        path = "dl_gemv" if B <= 4 else ("dl_gemm_smallm" if maxb > 4 else "library GEMM")
        print(f"B={B} ({path:12s}): {t / 64 * 1e3:6.3f} ms/step  {B * 64 / t:8.1f} tok/s", flush=True)
