"""Sweep of the partner-range trim of dl_linear_packed's in-kernel hand-over (partners shorter than an even k share so that their tiles are written through while the
reducer still multiplies).  1x MI355X, M = 170: q|k|v 41.2 (even) / 36.0 / 35.9 / 35.9 / 35.9 / 37.2 / 38.4 us at trim 0 / 12 / 24 / 36 / 48 / 64 / 80 of 256 -> default 24."""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd())
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
def timed(fns, reps=2):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns: f()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps * len(fns)) * 1e3
for M in (170, 117):
    for name, N, K, nu, ks in (("qkv", 12288, 4096, 6, 2), ("o", 4096, 4096, 2, 2)):
        ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(6)]
        wps = [ops.pack_weight_tiles(w) for w in ws]
        x = torch.randn(M, K, device=dev, dtype=dt); xp = ops.pack_x_tiles(x)
        out = torch.empty(M, N, device=dev, dtype=dt)
        wsb = ops.linear_packed_workspace(M, N, K, dev, 0, nu, ks)
        ref = F.linear(x.float(), ws[0].float())
        line = f"M={M} {name} nu={nu} ks={ks}:"
        for trim in (0, 12, 24, 36, 48, 64, 80):
            got = ops.linear_packed(xp, wps[0], N, units_per_workgroup=nu, k_split=ks, workspace=wsb, x_packed_mk=(M, K), _trim256=trim).float()
            err = float((got - ref).abs().max() / ref.abs().max())
            t = timed([lambda wp=wp: ops.linear_packed(xp, wp, N, out=out, units_per_workgroup=nu, k_split=ks, workspace=wsb, x_packed_mk=(M, K), _trim256=trim) for wp in wps])
            line += f" trim {trim}/256: {t:.2f}us (err {err:.0e}) |"
        print(line, flush=True)
