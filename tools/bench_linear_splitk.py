#!/usr/bin/env python
"""dl_linear_splitk vs the library GEMM on the prefill's narrow projections (o_proj / down_proj, M = 117 / 170 rows), cold weights,
with and without the consumer (dl_add_rmsnorm_parts vs library + dl_add_rmsnorm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
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


for M in (170, 117):
    for name, N, K in [("o", 4096, 4096), ("down", 4096, 11008), ("qkv", 12288, 4096), ("gate|up", 22016, 4096)]:
        ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(NB)]
        x = torch.randn(M, K, device=dev, dtype=dt)
        parts = torch.empty(16 * M * N, device=dev, dtype=torch.float32)
        t_lib = timed([lambda w=w: F.linear(x, w) for w in ws])
        ref = F.linear(x.float(), ws[0].float())
        line = []
        for s_ in (1, 2, 4, 8):
            if s_ > K // 128: continue
            t = timed([lambda w=w: ops.linear_splitk(x, w, parts, s_) for w in ws])
            got = ops.linear_splitk(x, ws[0], parts, s_).sum(0)
            err = float((got - ref).abs().max() / ref.abs().max())
            line.append(f"s={s_}: {t:6.2f}us (err {err:.1e})")
        print(f"M={M} {name:8s} [{N},{K}]: library {t_lib:6.2f}us | splitk " + "  ".join(line), flush=True)
        del ws
