#!/usr/bin/env python
"""Library GEMM time of the prefill's decoder projections as a function of the packed row count M (cold weights, hipGraph-timed): which row counts
the width buckets of the prefill should land on (model.prefill_width_bucket)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
dev, dt = "cuda", torch.bfloat16
NB = 6


def timed(fns, reps=2):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns: f()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps * len(fns)) * 1e3


Ms = [int(a) for a in sys.argv[1:]] or [128, 144, 160, 170, 176, 192, 208, 224, 240, 256]
for name, N, K in [("qkv", 12288, 4096), ("gate|up", 22016, 4096), ("o", 4096, 4096), ("down", 4096, 11008)]:
    ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(NB)]
    line = []
    for M in Ms:
        x = torch.randn(M, K, device=dev, dtype=dt)
        line.append(f"M={M}: {timed([lambda w=w: F.linear(x, w) for w in ws]):6.2f}")
    print(f"{name:8s} [{N},{K}] us: " + "  ".join(line), flush=True)
    del ws
