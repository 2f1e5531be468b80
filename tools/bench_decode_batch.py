#!/usr/bin/env python
"""Decode ms/step per batch size: dl_gemv (B <= 3), dl_gemm_smallm (4..16, vs the library GEMM it replaces), library GEMM beyond.
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


batches = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 8, 16, 20, 24, 32]
for B in batches:
    # maxb 0: dl_gemm_smallm off (library GEMM past the dl_gemv range); lp_min: from which batch the MLP runs on dl_linear_packed (99: never)
    for maxb, lp_min, lpq in ((0, 4, 0), (0, 4, 1), (32, 4, 0), (32, 4, 1)):
        if B <= 3 and (maxb == 0 or lp_min != 99):
            continue
        model.smallm_max_decode_batch = maxb
        model.packed_decode_mlp_min_batch = lp_min
        model.packed_decode_qkv_min_batch = 4 if lpq else 99
        wide = 0
        model._dstate = None
        for _ in range(2):
            run(B, 33); run(B, 1)
        t = min(run(B, 65) for _ in range(3)) - min(run(B, 1) for _ in range(3))
        path = "dl_gemv" if B <= model.gemv_max_decode_batch else ("dl_gemm_smallm" if B <= maxb else "library GEMM")
        mlp = "MLP on dl_linear_packed" if (B > model.gemv_max_decode_batch and lp_min <= B <= 32) else "MLP on the same"
        print(f"B={B} (q|k|v, o: {path:14s} {mlp:24s}{', q|k|v too' if lpq else ''}): {t / 64 * 1e3:6.3f} ms/step  {B * 64 / t:8.1f} tok/s", flush=True)
