#!/usr/bin/env python
"""Sweep of the decode-attention variants (fused vs rope+attn, workgroup size, split count) on bench-like shapes."""
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


for B, Ts in [(1, [226]), (1, [695]), (1, [2048]), (4, [226] * 4), (32, [200 + (i * 701) % 700 for i in range(32)]), (32, [2048] * 32)]:
    T_cap = max(Ts) + 1
    # 8 distinct slabs (> Infinity Cache for the big shapes; for small ones the decode step's slab is cold too)
    n_buf = 8 if B * T_cap * H * 2 * 2 < 200e6 else 2
    ks = [torch.randn(B, nH, T_cap, d, device=dev, dtype=dt) for _ in range(n_buf)]
    vs = [torch.randn(B, nH, T_cap, d, device=dev, dtype=dt) for _ in range(n_buf)]
    qkv = torch.randn(B, 3 * H, device=dev, dtype=dt)
    out = torch.empty(B, H, device=dev, dtype=dt)
    lens = torch.tensor([t - 1 for t in Ts], dtype=torch.int32, device=dev)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
    ws = ops.attn_decode_workspace(B, nH, d, 64, dev)
    nbytes = sum(2 * t * H * 2 for t in Ts)
    it = [0]
    res = []
    for ns in (1, 2, 4, 8, 16, 32):
        def unfused():
            i = it[0] = (it[0] + 1) % n_buf
            ops.rope_kv_write(qkv, cos, sin, cu, None, lens, lens, ks[i], vs[i], nH, nH, d)
            ops.attn_decode(qkv[:, :H], ks[i], vs[i], lens, 1, out, ws, ns, nH, nH, d)
        res.append(("rope+attn4", ns, timed(unfused)))
        for kif, chunk in ((64, 0), (64, 256), (128, 256), (256, 256)):
            def fused():
                i = it[0] = (it[0] + 1) % n_buf
                ops.attn_decode_rope(qkv, cos, sin, lens, lens, ks[i], vs[i], out, ws, ns, nH, nH, d, keys_in_flight=kif, chunk_keys=chunk)
            res.append((f"fused{kif}/{chunk}", ns, timed(fused)))
    print(f"B={B} T={Ts[0]}.. bytes={nbytes/1e6:.1f}MB")
    for name in ("rope+attn4", "fused64/0", "fused64/256", "fused128/256", "fused256/256"):
        print(f"   {name:13s} " + "  ".join(f"ns={ns}:{us:6.2f}us" for n, ns, us in res if n == name))
