#!/usr/bin/env python
"""dl_attn_decode_rope_oproj (one launch) vs dl_attn_decode_rope + dl_gemv (two launches) at the bench shapes; cold K/V and weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
from oracle.ref_cpu import rope_table

dev, dt = "cuda", torch.bfloat16
nH, d = 32, 128
H = nH * d
cos, sin = (t.to(dev) for t in rope_table(d, 4096, 10000.0, dt))


def timed(fn, reps=64, replays=5):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(replays): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * replays) * 1e3


for T, ns in ((234, 4), (695, 8)):
    NB = 16
    ks = [torch.randn(1, nH, T + 8, d, device=dev, dtype=dt) for _ in range(NB)]
    vs = [torch.randn_like(k) for k in ks]
    ws_ = [torch.randn(H, H, device=dev, dtype=dt) * 0.02 for _ in range(NB)]
    filler = [torch.randn(22016, 4096, device=dev, dtype=dt) * 0.02 for _ in range(4)]  # streamed between calls, like gate|up in the real step
    qkv = torch.randn(1, 3 * H, device=dev, dtype=dt)
    attn = torch.empty(1, H, device=dev, dtype=dt)
    y = torch.empty(1, H, device=dev, dtype=dt)
    yi = torch.empty(1, 11008, device=dev, dtype=dt)
    hh, h2, dl = (torch.randn(1, 4096, device=dev, dtype=dt) for _ in range(3))
    nw = torch.ones(4096, device=dev, dtype=dt)
    lens = torch.tensor([T], dtype=torch.int32, device=dev)
    wsb = ops.attn_decode_workspace(1, nH, d, 32, dev)

    def fill(i):
        ops.gemv(filler[i % 4], yi, mode=ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR, h_in=hh, h_out=h2, delta=dl, norm_w=nw, eps=1e-5)

    def two(i):
        fill(i)
        ops.attn_decode_rope(qkv, cos, sin, lens, lens, ks[i % NB], vs[i % NB], attn, wsb, ns, nH, nH, d, call_tag=i % 251)
        ops.gemv(ws_[i % NB], y, x=attn)

    def one(i):
        fill(i)
        ops.attn_decode_rope_oproj(qkv, cos, sin, lens, lens, ks[i % NB], vs[i % NB], attn, wsb, ns, i % 251, nH, nH, d, ws_[i % NB], y)

    t_fill = timed(fill)
    t2 = timed(two) - t_fill
    t1 = timed(one) - t_fill
    print(f"T={T} n_splits={ns}: attention + o_proj as two launches {t2:.2f} us, fused {t1:.2f} us (gate|up filler {t_fill:.2f} us subtracted)")
