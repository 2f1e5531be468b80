#!/usr/bin/env python
"""BASELINE configs[2] -- 32 ragged requests in one packed batch -- five eager prefills (generate, 1 new token, no hipGraph) for rocprofv3: the MFMA-bound leg
of the path.
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d <dir> -o m -- python tools/prefill_kernels_c2.py; python tools/mfma_report.py <db>
A marker kernel (torch.cuda._sleep -> `spin_kernel`) is launched after build + warm-up: the reports drop everything before it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig

cfg = DynamicLlavaConfig()
dev, dtype = torch.device("cuda"), torch.bfloat16
model = build_random_model(cfg, dtype=dtype, device=dev, seed=0, predictor_gain=50.0)
model.use_hip_graph = False
g = torch.Generator().manual_seed(1)  # bench.py configs2_leg's inputs
B = 32
n_q = torch.randint(8, 65, (B,), generator=g).tolist()
W = 35 + 1 + max(n_q)
ids = torch.zeros(B, W, dtype=torch.long)
am = torch.zeros(B, W, dtype=torch.long)
for b in range(B):
    row = torch.cat([torch.tensor([1]), torch.randint(3, cfg.vocab_size, (34,), generator=g), torch.tensor([-200]), torch.randint(3, cfg.vocab_size, (n_q[b],), generator=g)])
    ids[b, : row.numel()] = row
    am[b, : row.numel()] = 1
images = torch.randn(B, 3, 336, 336, generator=g).to(dtype).to(dev)
ids, am = ids.to(dev), am.to(dev)
model.generate(ids, attention_mask=am, images=images, max_new_tokens=1, eos_token_id=None)
torch.cuda.synchronize()
torch.cuda._sleep(100000)  # marker kernel
torch.cuda.synchronize()
for _ in range(5):
    model.generate(ids, attention_mask=am, images=images, max_new_tokens=1, eos_token_id=None)
torch.cuda.synchronize()
