#!/usr/bin/env python
"""13 eager prefills (generate, 1 new token, no hipGraph) for a rocprofv3 kernel trace: per-prefill kernel inventory.
    rocprofv3 --kernel-trace -d gpurun_out/prof_pf -o pf -- python tools/prefill_kernels.py; python tools/prof_summary.py <db> 60 --after spin_kernel
(divide the call counts by 13).  A marker kernel (torch.cuda._sleep -> `spin_kernel`) is launched after the model has been built and warmed
up: `prof_summary.py --after spin_kernel` drops everything before it (weight init, first-call lazy initialisation)."""
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
model.generate(prompt, images=images, max_new_tokens=1, eos_token_id=None)  # lazy library initialisation stays out of the inventory
torch.cuda.synchronize()
torch.cuda._sleep(100000)  # marker kernel
torch.cuda.synchronize()
for _ in range(13):
    model.generate(prompt, images=images, max_new_tokens=1, eos_token_id=None)
torch.cuda.synchronize()
