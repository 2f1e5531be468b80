#!/usr/bin/env python
"""dl_linear_tiles (80- and 160-row tiles, 1 / 2 / 4 k ranges) against hipBLASLt on the DECODER's GEMM shapes at more than 256 rows (layers 0-1 of a one-image
prefill at M = 631, batched prefills).  Result (profiles/r06_linear_tiles_dense_shapes.txt): 0.76 x the library at down_proj M = 631 (70.8 vs 92.6 us), 0.79 x at
o_proj (27.7 vs 34.9); the library wins everywhere else (q|k|v 631, every shape from 1262 rows on).  Not wired in: two layers x ~20 us is 0.4 % of a prefill."""
import sys, os
sys.path.insert(0, "/root/repo")
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
for name, M, N, K in [("o_proj 631", 631, 4096, 4096), ("down 631", 631, 4096, 11008), ("down 640", 640, 4096, 11008), ("o_proj 1360", 1360, 4096, 4096), ("down 1360", 1360, 4096, 11008), ("down 1262", 1262, 4096, 11008),("down 5048", 5048, 4096, 11008), ("qkv 631", 631, 12288, 4096)]:
    x = torch.randn(M, K, device=dev, dtype=dt)
    xp = ops.pack_x_rows(x)
    NW = 4
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(NW)]
    wps = [ops.pack_weight_tiles(w) for w in ws]
    y = torch.empty(M, N, device=dev, dtype=dt)
    fl = 2 * M * N * K
    t_l = timed(lambda i: F.linear(x, ws[i % NW]))
    line = f"{name:12s}: hipBLASLt {t_l:6.2f} us ({fl / t_l / 1e6:5.0f} TF/s) |"
    for sh in (1042, 1041, 542):
        for ks in (1, 2, 4):
            if ks > 1:
                pbuf = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
                f = lambda i: ops.linear_tiles(xp, wps[i % NW], N, out=pbuf, epilogue=ops.LT_PARTS, x_packed_mk=(M, K), tile_shape=sh, k_split=ks)
            else:
                f = lambda i: ops.linear_tiles(xp, wps[i % NW], N, out=y, epilogue=ops.LT_BIAS, x_packed_mk=(M, K), tile_shape=sh)
            try:
                t = timed(f)
            except ops.HipOpsError as e:
                line += f" {sh}/k{ks} ERR"; continue
            line += f" {sh}/k{ks} {t:6.2f}"
    print(line, flush=True)
    tp = timed(lambda i: ops.pack_x_rows(x))
    print(f"      pack_x_rows {tp:.2f} us")
