#!/usr/bin/env python
"""dl_gemv_qkv_attn (one launch) vs dl_gemv(ADDNORM) + dl_attn_decode_rope (two launches) on the bench workload's sparse layers (7B, T ~ 200),
cold weights and K/V (rotated), hipGraph-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
from oracle.ref_cpu import rope_table

dev, dt = "cuda", torch.bfloat16
nH, d, H = (40, 128, 5120) if "--13b" in sys.argv else (32, 128, 4096)
sys.argv = [a for a in sys.argv if a != "--13b"]
N = 3 * H
NB = 8


def timed(fn, reps=NB * 2):
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


for T in ([int(a) for a in sys.argv[1:]] or [200, 250, 60]):
    T_cap = T + 8
    cos, sin = (t.to(dev) for t in rope_table(d, T_cap + 8, 10000.0, dt))
    ws = [torch.randn(N, H, device=dev, dtype=dt) * 0.02 for _ in range(NB)]
    ks = [torch.randn(1, nH, T_cap, d, device=dev, dtype=dt) for _ in range(NB)]
    vs = [torch.randn(1, nH, T_cap, d, device=dev, dtype=dt) for _ in range(NB)]
    nw = torch.ones(H, device=dev, dtype=dt)
    h0, delta, ho = torch.randn(1, H, device=dev, dtype=dt), torch.randn(1, H, device=dev, dtype=dt), torch.empty(1, H, device=dev, dtype=dt)
    qkv, out = torch.empty(1, N, device=dev, dtype=dt), torch.empty(1, H, device=dev, dtype=dt)
    lens = torch.tensor([T - 1], dtype=torch.int32, device=dev)
    gran = ops.gemv_qkv_attn_workspace(nH, nH, d, dev)
    it = [0]
    tagc = [0]

    def two(kif):
        def f():
            i = it[0] = (it[0] + 1) % NB
            ops.gemv(ws[i], qkv, mode=ops.GEMV_ADDNORM, h_in=h0, h_out=ho, delta=delta, norm_w=nw, eps=1e-5)
            ops.attn_decode_rope(qkv, cos, sin, lens, lens, ks[i], vs[i], out, None, 1, nH, nH, d, keys_in_flight=kif, chunk_keys=256)
        return f

    aws = ops.attn_decode_workspace(1, nH, d, 32, dev)

    def two_split(ns):
        def f():
            i = it[0] = (it[0] + 1) % NB
            tagc[0] = (tagc[0] + 1) % 251
            ops.gemv(ws[i], qkv, mode=ops.GEMV_ADDNORM, h_in=h0, h_out=ho, delta=delta, norm_w=nw, eps=1e-5)
            ops.attn_decode_rope(qkv, cos, sin, lens, lens, ks[i], vs[i], out, aws, ns, nH, nH, d, keys_in_flight=64, chunk_keys=0, call_tag=tagc[0])
        return f

    def gemv_only():
        i = it[0] = (it[0] + 1) % NB
        ops.gemv(ws[i], qkv, mode=ops.GEMV_ADDNORM, h_in=h0, h_out=ho, delta=delta, norm_w=nw, eps=1e-5)

    def fused(ns=1):
        def f():
            i = it[0] = (it[0] + 1) % NB
            tagc[0] = (tagc[0] + 1) % 251
            ops.gemv_qkv_attn(ws[i], qkv, h0, ho, delta, nw, 1e-5, cos, sin, lens, lens, ks[i], vs[i], out, gran, tagc[0], nH, nH, d, n_splits=ns)
        return f

    print(f"T={T}: q|k|v gemv alone {timed(gemv_only):6.2f} us | + attention (4 waves) {timed(two(64)):6.2f} us | + attention (8 waves) {timed(two(128)):6.2f} us | one launch, 1 / 2 / 3 / 4 attention workgroups per head {timed(fused(1)):6.2f} / {timed(fused(2)):6.2f} / {timed(fused(3)):6.2f} / {timed(fused(4)):6.2f} us | + split attention (n_splits = min(256 // nH, ceil(T / 64)) = {max(1, min(256 // nH, -(-T // 64)))}, in-kernel merge) {timed(two_split(max(1, min(256 // nH, -(-T // 64))))):6.2f} us", flush=True)
