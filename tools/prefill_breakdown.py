#!/usr/bin/env python
"""Where does steady-state prefill time go?  generate(max_new_tokens=1) wall time vs the captured prefill graph alone vs host pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench

cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
for _ in range(3):
    model.generate(prompt, images=images, max_new_tokens=1, eos_token_id=None)
torch.cuda.synchronize()


def wall(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print("generate(max_new_tokens=1) wall ms:", round(wall(lambda: model.generate(prompt, images=images, max_new_tokens=1, eos_token_id=None)), 3))
ent = next(iter(model._prefill_graphs.values()))
print("prefill graph replay only ms:", round(wall(lambda: ent["graph"].replay()), 3))
t0 = time.perf_counter()
for _ in range(50): lay = model._layout(prompt, None, None, 576)
print("_layout host ms:", round((time.perf_counter() - t0) / 50 * 1e3, 3))
print("CLIP+projector eager ms:", round(wall(lambda: model.encode_images(images)), 3))
# CLIP alone inside a graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): model.encode_images(images)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): f = model.encode_images(images)
print("CLIP+projector graph ms:", round(wall(lambda: g.replay()), 3))
