#!/usr/bin/env python
"""How much of the configs[2]-like decode attention (B=32 ragged, one workgroup per (row, head), all 1024 resident at once) is the idle tail of
CUs whose four resident workgroups happen to be long rows?  Same kernel, same lengths, only the ORDER of the rows in the batch changes:
random (the bench order), sorted ascending (co-resident workgroups = rows r, r+8, r+16, r+24 under round-robin dispatch: unbalanced) and a
snake order in which those four positions always sum to about the same length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
from oracle.ref_cpu import rope_table

dev, dt = "cuda", torch.bfloat16
nH, d = 32, 128
H = nH * d
cos, sin = (t.to(dev) for t in rope_table(d, 4096, 10000.0, dt))


def timed(fn, reps=40):
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


base = [200 + (i * 229) % 700 for i in range(32)]
srt = sorted(base, reverse=True)
snake = [0] * 32
for r in range(8):
    snake[r], snake[r + 8], snake[r + 16], snake[r + 24] = srt[r], srt[15 - r], srt[16 + r], srt[31 - r]
# variants of "which four rows share a CU": stride 8 (ids r + 256 s), stride 1 (4 consecutive rows), stride 2 ...
orders = {"bench order": base, "sorted descending": srt, "sorted ascending": srt[::-1], "snake (rows r, r+8, r+16, r+24 balanced)": snake}
snake4 = []
for r in range(8):
    snake4 += [srt[r], srt[15 - r], srt[16 + r], srt[31 - r]]
orders["snake (rows 4r..4r+3 balanced)"] = snake4
for name, Ts in orders.items():
    B = 32
    T_cap = max(Ts) + 1
    n_buf = 8
    ks = [torch.randn(B, nH, T_cap, d, device=dev, dtype=dt) for _ in range(n_buf)]
    vs = [torch.randn(B, nH, T_cap, d, device=dev, dtype=dt) for _ in range(n_buf)]
    qkv = torch.randn(B, 3 * H, device=dev, dtype=dt)
    out = torch.empty(B, H, device=dev, dtype=dt)
    lens = torch.tensor([t - 1 for t in Ts], dtype=torch.int32, device=dev)
    ws = ops.attn_decode_workspace(B, nH, d, 64, dev)
    nbytes = sum(2 * t * H * 2 for t in Ts)
    it = [0]

    def fused():
        i = it[0] = (it[0] + 1) % n_buf
        ops.attn_decode_rope(qkv, cos, sin, lens, lens, ks[i], vs[i], out, ws, 1, nH, nH, d)

    us = timed(fused)
    print(f"{name:45s}: {us:6.2f} us  {nbytes / us / 1e6:5.2f} TB/s  frac {nbytes / us / 1e6 / 8:.3f}", flush=True)
    del ks, vs
