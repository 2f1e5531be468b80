#!/usr/bin/env python
"""dl_linear_splitk + its consumer (dl_add_rmsnorm_parts) as a PAIR, per split-K factor: more slices shorten the GEMM and lengthen the
consumer (it reads s x M x N fp32).  o_proj / down_proj of the prefill at M = 176 / 128 rows, cold weights (6 matrices in turn).
Round 4, final tree (us, GEMM + consumer = pair): M=176 o: s=4 31.3, s=6 29.3, s=8 26.4, s=12 32.2, s=16 36.2; down: s=4 56.4, s=6 49.7, s=8 43.3, s=12 47.4,
s=16 61.8 (M=128: o 24.2, down 38.1 at s=8, again the minimum): 8 slices = 512 workgroups = two per CU stands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
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


for M in (176, 128):
    for name, N, K in [("o", 4096, 4096), ("down", 4096, 11008)]:
        ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(NB)]
        x = torch.randn(M, K, device=dev, dtype=dt)
        h = torch.randn(M, N, device=dev, dtype=dt)
        nw = torch.ones(N, device=dev, dtype=dt)
        parts = torch.empty(16 * M * N, device=dev, dtype=torch.float32)
        line = []
        for s_ in (2, 4, 6, 8, 12, 16):
            t_g = timed([lambda w=w: ops.linear_splitk(x, w, parts, s_) for w in ws])
            t_p = timed([lambda w=w: ops.add_rmsnorm_parts(h, ops.linear_splitk(x, w, parts, s_), nw, 1e-5) for w in ws])
            line.append(f"s={s_}: {t_g:5.1f} + {t_p - t_g:4.1f} = {t_p:5.1f}")
        print(f"M={M} {name:5s}: " + " | ".join(line), flush=True)
        del ws
