#!/usr/bin/env python
"""dl_linear on the VisionPredictor shapes (B=1: 576 image tokens), graph-timed, against F.linear (hipBLASLt) + the unfused epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16


def timed(fn, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


tot_m = tot_l = 0.0
for name, M, N, K, flags in [("in_conv+gelu", 576, 512, 4096, ops.EPI_GELU), ("qkv", 576, 1536, 512, 0), ("out+res", 576, 512, 512, ops.EPI_RESIDUAL),
                             ("ff1+gelu", 576, 2048, 512, ops.EPI_GELU), ("ff2+res", 576, 512, 2048, ops.EPI_RESIDUAL), ("out1+gelu", 576, 256, 512, ops.EPI_GELU),
                             ("out2+gelu", 576, 128, 256, ops.EPI_GELU)]:
    x = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * 0.02; b = torch.randn(N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev, dtype=dt); y = torch.empty(M, N, device=dev, dtype=dt)
    t_m = timed(lambda: ops.linear(x, w, b, out=y, residual=r if flags & ops.EPI_RESIDUAL else None, flags=flags))
    def lib():
        o = F.linear(x, w, b)
        if flags & ops.EPI_GELU: o = F.gelu(o)
        if flags & ops.EPI_RESIDUAL: o = r + o
        return o
    t_l = timed(lib)
    tot_m += t_m; tot_l += t_l
    print(f"{name:14s} [{M},{K}]x[{N},{K}]: dl_linear {t_m:6.2f} us ({2*M*N*K/t_m/1e6:6.1f} TFLOP/s) | library + unfused epilogue {t_l:6.2f} us")
print(f"sum: dl_linear {tot_m:.1f} us, library {tot_l:.1f} us")
