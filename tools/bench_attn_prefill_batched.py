#!/usr/bin/env python
"""The decoder's prefill attention over a BATCH of compacted requests (configs[2] / [3]: 32 requests x 32 heads x 158..214 rows, head_dim 128, causal): the whole-head
kernel (one workgroup per (request, head), late round 6) against the plain kernel it replaces (DL_PF_WHOLE128=0), graph-timed.
    python tools/bench_attn_prefill_batched.py; DL_PF_WHOLE128=0 python tools/bench_attn_prefill_batched.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
nH = nKV = 32; d = 128


def timed(fn, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


g = torch.Generator().manual_seed(1)
for B, lo, hi in ((32, 158, 215), (8, 158, 215), (4, 158, 215), (2, 158, 215), (1, 170, 171), (32, 170, 171), (64, 100, 257), (256, 158, 215)):
    lens = torch.randint(lo, hi, (B,), generator=g).tolist()
    total = sum(lens)
    qkv = torch.randn(total, 3 * nH * d, generator=g).to(dt).to(dev)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    out = torch.empty(total, nH * d, dtype=dt, device=dev)
    t = timed(lambda: ops.attn_prefill(qkv[:, : nH * d], qkv[:, nH * d : 2 * nH * d], qkv[:, 2 * nH * d :], out, cu, max(lens), nH, nKV, d, True))
    fl = sum(4 * nH * d * L * (L + 1) / 2 for L in lens)
    print(f"B={B:3d} rows {lo}..{hi - 1}: {t:8.2f} us  ({fl / t / 1e6:6.1f} TFLOP/s causal)   DL_PF_WHOLE128={os.environ.get('DL_PF_WHOLE128', '1')}", flush=True)

# the CLIP tower's attention (16 heads x 64, 577 tokens, non-causal) at 1..32 images: DL_PF_HEAD64_MIN=100000 keeps the per-64-row-block kernel at every batch size
nH2, d2 = 16, 64
for B in (32, 16, 8, 2, 1):
    total = 577 * B
    qkv = torch.randn(total, 3 * nH2 * d2, generator=g).to(dt).to(dev)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * 577
    out = torch.empty(total, nH2 * d2, dtype=dt, device=dev)
    t = timed(lambda: ops.attn_prefill(qkv[:, : nH2 * d2], qkv[:, nH2 * d2 : 2 * nH2 * d2], qkv[:, 2 * nH2 * d2 :], out, cu, 577, nH2, nH2, d2, False))
    fl = B * 4 * nH2 * d2 * 577 * 577
    print(f"CLIP B={B:3d} x 577: {t:8.2f} us  ({fl / t / 1e6:6.1f} TFLOP/s)   DL_PF_HEAD64_MIN={os.environ.get('DL_PF_HEAD64_MIN', '256')}", flush=True)
