#!/usr/bin/env python
"""TunableOp on the decode GEMM shapes at M = 25..32 (BASELINE configs[2]/[3] steady state) and the prefill shapes."""
import os, sys, time
import torch
import torch.nn.functional as F
dev, dt = "cuda", torch.bfloat16
Ms = [int(m) for m in os.environ.get("MS", "32").split(",")]
shapes = [(M, N, K) for M in Ms for (N, K) in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]]

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
    tot = {}
    for M, N, K in shapes:
        ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(8)]
        x = torch.randn(M, K, device=dev, dtype=dt)
        us = timed([lambda w=w: F.linear(x, w) for w in ws])
        tot[M] = tot.get(M, 0) + us
        print(f"{tag} M={M} N={N} K={K}: {us:8.2f} us  {N*K*2/us/1e3:7.0f} GB/s weights", flush=True)
    for M, t in tot.items(): print(f"{tag} M={M}: sum of the four GEMMs of a layer {t:.1f} us", flush=True)

run("default ")
torch.cuda.tunable.enable(True)
torch.cuda.tunable.set_max_tuning_duration(150)
torch.cuda.tunable.set_max_tuning_iterations(20)
torch.cuda.tunable.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tunableop_m32.csv"))
t0 = time.time()
for M, N, K in shapes:
    w = torch.randn(N, K, device=dev, dtype=dt); x = torch.randn(M, K, device=dev, dtype=dt)
    F.linear(x, w); torch.cuda.synchronize()
print("tuning took", round(time.time() - t0, 1), "s", flush=True)
run("tunable ")
# (results are written by TunableOp itself at exit)
