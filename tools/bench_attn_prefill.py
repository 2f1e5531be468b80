#!/usr/bin/env python
"""dl_attn_prefill on the bench prompt shapes (B=1, T=170 / 631; 32 heads x 128), graph-timed.  DL_PF_NW=1|2|4 forces the waves/WG."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
for B, T, nH, d, causal in [(1, 170, 32, 128, True), (1, 117, 32, 128, True), (1, 631, 32, 128, True), (8, 170, 32, 128, True), (32, 215, 32, 128, True),
                            (32, 700, 32, 128, True), (1, 577, 16, 64, False), (1, 576, 8, 64, False), (2, 577, 16, 64, False)]:
    qkv = torch.randn(B * T, 3 * nH * d, device=dev, dtype=dt)
    out = torch.empty(B * T, nH * d, device=dev, dtype=dt)
    cu = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=dev)
    H = nH * d
    fn = lambda: ops.attn_prefill(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], out, cu, T, nH, nH, d, causal)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 100 * 1e3
    flops = (2 if causal else 4) * B * T * T * nH * d  # causal: half of 4*T^2*H
    print(f"B={B} T={T} heads={nH}x{d} {'causal' if causal else 'full'} NW={os.environ.get('DL_PF_NW','auto')}: {us:8.2f} us  {flops/us/1e6:7.1f} TFLOP/s")
