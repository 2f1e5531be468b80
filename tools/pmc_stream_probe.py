#!/usr/bin/env python
"""Launches for a PMC comparison of the two weight-streaming kernels on the same [12288,4096] / [22016,4096] weights: dl_gemv (B=1) and
dl_gemm_smallm (M = 8 / 32, partials left in the workspace).  Weights rotate over 6 copies (cold).  Run under
    rocprofv3 --kernel-trace --pmc <counters> -d <dir> -o s -- python tools/pmc_stream_probe.py
and summarise with tools/pmc_stream_report.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
for N, K in [(12288, 4096), (22016, 4096)]:
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(6)]
    x1 = torch.randn(1, K, device=dev, dtype=dt); y1 = torch.empty(1, N, device=dev, dtype=dt)
    for rep in range(2):
        for w in ws:
            ops.gemv(w, y1, x=x1, mode=ops.GEMV_PLAIN)
    for M in (8, 32):
        x = torch.randn(M, K, device=dev, dtype=dt)
        scratch = torch.empty(16 * M * N, device=dev, dtype=torch.float32)
        for rep in range(2):
            for w in ws:
                ops.gemm_smallm_parts(x, w, scratch, n_slices=0, variant=2)
    torch.cuda.synchronize()
    del ws
