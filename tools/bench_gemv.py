#!/usr/bin/env python
"""Tuning sweep for dl_gemv on the LLaVA-1.5-7B decode shapes (B=1).  Each timed graph walks 8 distinct weight
buffers (> 256 MB Infinity Cache in total) so that, as in the real decode step, every weight byte comes from HBM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynamic_llava_amd import hip_ops as ops

dev, dt = "cuda", torch.bfloat16
SHAPES = [("qkv  addnorm", 12288, 4096, ops.GEMV_ADDNORM), ("o    plain", 4096, 4096, ops.GEMV_PLAIN), ("gu   addnorm+pair", 22016, 4096, ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR),
          ("down plain", 4096, 11008, ops.GEMV_PLAIN), ("gu   addnorm", 22016, 4096, ops.GEMV_ADDNORM), ("down silumul", 4096, 11008, ops.GEMV_SILUMUL), ("lm_head addnorm", 32000, 4096, ops.GEMV_ADDNORM)]
if os.environ.get("MODEL") == "13b":  # LLaVA-1.5-13B (configs[4]): MODEL=13b python tools/bench_gemv.py
    SHAPES = [("qkv  addnorm", 15360, 5120, ops.GEMV_ADDNORM), ("o    plain", 5120, 5120, ops.GEMV_PLAIN), ("gu   addnorm+pair", 27648, 5120, ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR),
              ("down plain", 5120, 13824, ops.GEMV_PLAIN)]
CAPS = [int(c) for c in os.environ.get("CAPS", "512,1024,2048,4096").split(",")]
B = int(os.environ.get("B", "1"))
NW = 8


def timed(fn_list, reps=3):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fn_list:
            f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fn_list:
                f()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps * len(fn_list)) * 1e3  # us per launch


for name, N, K, mode in SHAPES:
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(NW)]
    pair = bool(mode & ops.GEMV_OUT_SILU_PAIR)
    y = torch.empty(B, N // 2 if pair else N, device=dev, dtype=dt)
    x = torch.randn(B, 2 * K if (mode & 3) == ops.GEMV_SILUMUL else K, device=dev, dtype=dt)
    h, h2, dl = torch.randn(B, K, device=dev, dtype=dt), torch.empty(B, K, device=dev, dtype=dt), torch.randn(B, K, device=dev, dtype=dt)
    nw = torch.ones(K, device=dev, dtype=dt)
    best = None
    for variant in (0,):
        for cap in CAPS:
            if (mode & 3) == ops.GEMV_ADDNORM:
                fns = [lambda w=w: ops.gemv(w, y, mode=mode, h_in=h, h_out=h2, delta=dl, norm_w=nw, eps=1e-5, grid_cap=cap) for w in ws]
            else:
                fns = [lambda w=w: ops.gemv(w, y, x=x, mode=mode, grid_cap=cap) for w in ws]
            us = timed(fns)
            gbs = N * K * 2 / us / 1e3
            print(f"{name:20s} N={N:6d} K={K:6d} B={B} variant={variant} cap={cap:6d}  {us:8.2f} us  {gbs:8.1f} GB/s")
            if best is None or us < best[0]:
                best = (us, variant, cap, gbs)
    # torch / hipBLASLt reference on the same rotating buffers
    xx = torch.randn(B, K, device=dev, dtype=dt)
    us = timed([lambda w=w: torch.nn.functional.linear(xx, w) for w in ws])
    print(f"## {name:20s} best {best[0]:.2f} us variant={best[1]} cap={best[2]} {best[3]:.0f} GB/s | torch F.linear {us:.2f} us {N*K*2/us/1e3:.0f} GB/s")
