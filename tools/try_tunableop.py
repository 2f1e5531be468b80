#!/usr/bin/env python
"""Does PyTorch TunableOp find faster hipBLASLt/rocBLAS solutions for the prefill GEMM shapes (M=170 / 631)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

dev, dt = "cuda", torch.bfloat16
shapes = [(170, 12288, 4096), (170, 4096, 4096), (170, 22016, 4096), (170, 4096, 11008), (631, 12288, 4096), (631, 22016, 4096), (631, 4096, 11008), (577, 3072, 1024), (577, 4096, 1024), (577, 1024, 4096)]


def timed(fns):
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
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / len(fns) * 1e3


def run(tag):
    for M, N, K in shapes:
        ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(8)]
        x = torch.randn(M, K, device=dev, dtype=dt)
        us = timed([lambda w=w: F.linear(x, w) for w in ws])
        print(f"{tag} M={M} N={N} K={K}: {us:8.2f} us  {N*K*2/us/1e3:7.0f} GB/s weights  {2*M*N*K/us/1e6:7.1f} TFLOP/s")


run("default ")
torch.cuda.tunable.enable(True)
torch.cuda.tunable.set_max_tuning_duration(200)
torch.cuda.tunable.set_max_tuning_iterations(20)
torch.cuda.tunable.set_filename("/tmp/tunableop.csv")
t0 = time.time()
for M, N, K in shapes:
    w = torch.randn(N, K, device=dev, dtype=dt); x = torch.randn(M, K, device=dev, dtype=dt)
    F.linear(x, w); torch.cuda.synchronize()
print("tuning took", round(time.time() - t0, 1), "s")
run("tunable ")
