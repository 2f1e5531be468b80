#!/usr/bin/env python
"""13 eager prefills (generate, 1 new token, no hipGraph) for a rocprofv3 kernel trace: per-prefill kernel inventory.
    rocprofv3 --kernel-trace -d gpurun_out/prof_pf -o pf -- python tools/prefill_kernels.py; python tools/prof_summary.py <db> 60
(divide the call counts by 13)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench

cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
model.use_hip_graph = False
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
for _ in range(13):
    model.generate(prompt, images=images, max_new_tokens=1, eos_token_id=None)
torch.cuda.synchronize()
