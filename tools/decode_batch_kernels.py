#!/usr/bin/env python
"""Kernel inventory of the batched decode step (default B=32, the configs[2] steady state) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d /tmp/prof_db -o db -- python tools/decode_batch_kernels.py [B]; python tools/prof_summary.py <db> 40 --after spin_kernel
One eager prefill + 33 decode steps after the marker (divide the decode kernels' call counts by 32 steps x 32 layers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
model.use_hip_graph = False
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
feats = model.encode_images(images)
ids = prompt.expand(B, -1).contiguous()
f = feats.expand(B, -1, -1).contiguous()
model.generate(ids, image_features=f, max_new_tokens=4, eos_token_id=None)
torch.cuda.synchronize()
torch.cuda._sleep(100000)  # marker kernel
torch.cuda.synchronize()
model.generate(ids, image_features=f, max_new_tokens=33, eos_token_id=None)
torch.cuda.synchronize()
