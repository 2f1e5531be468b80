#!/usr/bin/env python
"""dl_gemm_smallm vs the library GEMM (torch.mm -> hipBLASLt) on the decoder's four weight shapes at decode batch sizes, weights
rotated over 8 distinct copies (every launch streams cold weights, as in the layer chain).  Sweeps n_slices x wg_waves.
    python tools/bench_gemm_smallm.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops

dev, dt = "cuda", torch.bfloat16
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
    for _ in range(3): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * reps) * 1e3


Ms = [int(a) for a in sys.argv[1:]] or [8, 16, 32]
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate|up", 22016, 4096), ("down", 4096, 11008)]
for M in Ms:
    tot_lib = tot_best = tot_auto = 0.0
    for name, N, K in shapes:
        ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(NB)]
        x = torch.randn(M, K, device=dev, dtype=dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        scratch = torch.empty(16 * M * N, device=dev, dtype=torch.float32)
        it = [0]

        def lib():
            i = it[0] = (it[0] + 1) % NB
            torch.mm(x, ws[i].t(), out=y)

        t_lib = timed(lib)
        res = []
        for nw in (8, 2, 3):  # 8: direct fragments (8 waves); 2 / 3: the LDS-staged variants (8 waves x 256 k / 16 waves x 128 k)
            for sp in (1, 2, 3, 4, 6, 8):
                def mine():
                    i = it[0] = (it[0] + 1) % NB
                    ops.gemm_smallm(x, ws[i], out=y, workspace=scratch, n_slices=sp, wg_waves=0 if nw in (2, 3) else nw, variant=nw if nw in (2, 3) else 1)
                try:
                    res.append((timed(mine), nw, sp))
                except Exception as e:
                    pass

        def auto():
            i = it[0] = (it[0] + 1) % NB
            ops.gemm_smallm(x, ws[i], out=y, workspace=scratch)

        t_auto = timed(auto)
        res.sort()
        gb = N * K * 2 / 1e3
        tot_lib += t_lib; tot_best += res[0][0]; tot_auto += t_auto
        print(f"M={M:3d} {name:8s} [{N},{K}] lib {t_lib:6.2f}us ({gb / t_lib:5.0f} GB/s) | auto {t_auto:6.2f}us ({gb / t_auto:5.0f} GB/s) | best " + "  ".join(f"{t:6.2f}us({'st8' if nw == 2 else ('st16' if nw == 3 else 'w' + str(nw))},s{sp})" for t, nw, sp in res[:4]), flush=True)
        del ws
    print(f"M={M}: per-layer GEMM time lib {tot_lib:.1f}us, auto {tot_auto:.1f}us, best-of {tot_best:.1f}us")
