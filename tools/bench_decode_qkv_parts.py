#!/usr/bin/env python
"""Decode ms/step at 16 / 24 / 32 rows with the q|k|v projection's two k ranges (a) handed over inside dl_linear_packed's launch (round 5) and (b) left as fp32
partial sums that dl_attn_decode_rope_parts adds (round 6, `packed_decode_qkv_parts`).  Same box, interleaved.   python tools/bench_decode_qkv_parts.py [B ...]"""
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


for B in [int(a) for a in sys.argv[1:]] or [16, 24, 32]:
    res = {}
    for rep in range(2):
        for parts in (False, True):
            model.packed_decode_qkv_parts, model.packed_decode_qkv_parts_max_batch = parts, 32
            model._dstate = None
            for _ in range(2):
                run(B, 33); run(B, 1)
            t = min(run(B, 129) for _ in range(3)) - min(run(B, 1) for _ in range(3))
            res.setdefault(parts, []).append(t / 128 * 1e3)
    a, b = min(res[False]), min(res[True])
    print(f"B={B}: hand-over inside the projection {a:.3f} ms/step, partial sums added by the attention launch {b:.3f} ms/step ({b / a:.4f})", flush=True)
