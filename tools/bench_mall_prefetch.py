#!/usr/bin/env python
"""Does a dl_gemv run faster when its weights were just pulled into the 256 MB Infinity Cache (MALL)?
Graph A: flush (360 MB of other weights) -> gemv.  Graph B: flush -> read W once -> gemv.  Each minus the same graph without the gemv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops

dev, dt = "cuda", torch.bfloat16


def timed(fns, reps=10):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


fl_w = [torch.randn(22016, 4096, device=dev, dtype=dt) * 0.02 for _ in range(2)]
fl_x = torch.randn(1, 4096, device=dev, dtype=dt)
fl_y = torch.empty(1, 22016, device=dev, dtype=dt)
flush = [lambda w=w: ops.gemv(w, fl_y, x=fl_x, mode=ops.GEMV_PLAIN) for w in fl_w]
for name, N, K in [("o", 4096, 4096), ("qkv", 12288, 4096), ("down", 4096, 11008), ("gate|up", 22016, 4096)]:
    W = torch.randn(N, K, device=dev, dtype=dt) * 0.02
    x = torch.randn(1, K, device=dev, dtype=dt)
    y = torch.empty(1, N, device=dev, dtype=dt)
    acc = torch.zeros(1, device=dev, dtype=torch.int32)
    Wi = W.view(torch.int32)
    pre = lambda: acc.copy_(Wi.sum().reshape(1))
    gv = lambda: ops.gemv(W, y, x=x, mode=ops.GEMV_PLAIN)
    t_f, t_fg = timed(flush), timed(flush + [gv])
    t_fp, t_fpg = timed(flush + [pre]), timed(flush + [pre, gv])
    print(f"{name:8s} [{N},{K}] {N*K*2/1e6:6.1f} MB: cold {t_fg - t_f:6.2f} us   after a read of W (MALL) {t_fpg - t_fp:6.2f} us   (the read itself {t_fp - t_f:6.2f} us)")
