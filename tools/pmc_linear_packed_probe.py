#!/usr/bin/env python
"""Launches for a request-counter comparison ("take the TCP / TCC request-counter pass per variant") of the M = 170 prefill
projections on: the library GEMM, dl_linear_splitk, and dl_linear_packed in the variants of tools/bench_linear_packed.py (row-major / fragment-order
X, 1 / 2 k ranges).  Weights rotate over 4 copies (cold).  Each variant is bracketed by marker launches (dl_pack_x_tiles of a [16, 64] matrix: one
before its warm-up call, one before its counted launches) so that the report can attribute the library's kernels too.  Run under
    rocprofv3 --kernel-trace --pmc <counters> -d <dir> -o s -- python tools/pmc_linear_packed_probe.py
once per counter group, summarise with tools/pmc_linear_packed_report.py <db> [<db> ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
M = 170
variants = []
for name, N, K in [("qkv", 12288, 4096), ("gate|up", 22016, 4096)]:
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(4)]
    wps = [ops.pack_weight_tiles(w) for w in ws]
    x = torch.randn(M, K, device=dev, dtype=dt)
    xp = ops.pack_x_tiles(x)
    out = torch.empty(M, N, device=dev, dtype=dt)
    parts = torch.empty(8 * M * N, device=dev, dtype=torch.float32)
    nu1 = 3 if name == "qkv" else 6
    cases = [("library", lambda w, wp: F.linear(x, w)), ("splitk s=2", lambda w, wp: ops.linear_splitk(x, w, parts, 2)),
             (f"packed rowX nu={nu1} ks=1", lambda w, wp: ops.linear_packed(x, wp, N, out=out, units_per_workgroup=nu1)),
             (f"packed nu={nu1} ks=1", lambda w, wp: ops.linear_packed(xp, wp, N, out=out, units_per_workgroup=nu1, x_packed_mk=(M, K)))]
    if name == "qkv":
        wsb = ops.linear_packed_workspace(M, N, K, dev, 0, 6, 2)
        cases.append(("packed nu=6 ks=2", lambda w, wp: ops.linear_packed(xp, wp, N, out=out, units_per_workgroup=6, k_split=2, workspace=wsb, x_packed_mk=(M, K))))
    for label, fn in cases:
        marker = torch.zeros(16, 64, device=dev, dtype=dt)
        ops.pack_x_tiles(marker)  # odd marker: what follows is this variant's warm-up (library heuristics, LDS attribute): not counted
        fn(ws[0], wps[0])
        torch.cuda.synchronize()
        ops.pack_x_tiles(marker)  # even marker: the counted launches of this variant follow
        torch.cuda.synchronize()
        for rep in range(2):
            for w, wp in zip(ws, wps):
                fn(w, wp)
        torch.cuda.synchronize()
        variants.append(f"{name}: {label}")
    del ws, wps
print("VARIANTS " + json.dumps(variants))
