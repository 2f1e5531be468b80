#!/usr/bin/env python
"""The CLIP ViT-L/14-336 tower's and the projector's GEMMs at B=1 (M = 577 / 576): hipBLASLt (F.linear, bias in the epilogue) vs dl_linear,
graph-timed, weights rotated over 8 copies (23 layers stream 0.6 GB through a 256 MB cache in the real tower)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16


def timed(fn, reps=24):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


tot = [0.0, 0.0]
for name, M, N, K, gelu in [("clip q|k|v", 577, 3072, 1024, 0), ("clip out_proj", 577, 1024, 1024, 0), ("clip fc1 (+quick_gelu)", 577, 4096, 1024, 0), ("clip fc2", 577, 1024, 4096, 0),
                            ("projector 1 (+gelu)", 576, 4096, 1024, 1), ("projector 2", 576, 4096, 4096, 0)]:
    x = torch.randn(M, K, device=dev, dtype=dt)
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(8)]
    b = torch.randn(N, device=dev, dtype=dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    t_l = timed(lambda i: F.linear(x, ws[i % 8], b))
    t_m = timed(lambda i: ops.linear(x, ws[i % 8], b, out=y, flags=ops.EPI_GELU if gelu else 0))
    fl = 2 * M * N * K
    n = 23 if name.startswith("clip") else 1
    tot[0] += t_l * n; tot[1] += t_m * n
    print(f"{name:24s} [{M},{K}]x[{N},{K}]: hipBLASLt {t_l:6.2f} us ({fl / t_l / 1e6:6.1f} TFLOP/s) | dl_linear {t_m:6.2f} us ({fl / t_m / 1e6:6.1f} TFLOP/s)")
print(f"tower (23 layers) + projector: hipBLASLt {tot[0] / 1e3:.3f} ms, dl_linear {tot[1] / 1e3:.3f} ms")
