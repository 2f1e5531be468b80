#!/usr/bin/env python
"""Launches for a counter comparison of the CLIP tower's GEMM shapes (M = 577) on the library and on dl_linear_tiles (weights direct to registers / through the
LDS ring as well), cold weights (rotation over > 256 MB).  Same marker protocol as tools/pmc_linear_packed_probe.py (dl_pack_x_tiles of a [16, 64] matrix before a
variant's warm-up call and before its counted launches), so tools/pmc_linear_packed_report.py summarises it:
    rocprofv3 --kernel-trace --pmc <counters> -d <dir> -o s -- python tools/pmc_linear_tiles_probe.py > probe.log
    python tools/pmc_linear_packed_report.py probe.log <db> [<db> ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16
M = 577
variants = []
for name, N, K, epi in [("fc1", 4096, 1024, ops.LT_QGELU), ("qkv", 3072, 1024, ops.LT_BIAS), ("fc2", 1024, 4096, ops.LT_PARTS)]:
    NW = int(300e6 / (N * K * 2)) + 1
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(NW)]
    wps = [ops.pack_weight_tiles(w) for w in ws]
    x = torch.randn(M, K, device=dev, dtype=dt)
    xp = ops.pack_x_rows(x)
    b = torch.randn(N, device=dev, dtype=dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    parts = torch.empty(4, M, N, device=dev, dtype=torch.float32)
    if epi == ops.LT_PARTS:
        cases = [("library", lambda w, wp: F.linear(x, w, b)),
                 ("tiles 542 k4 parts", lambda w, wp: ops.linear_tiles(xp, wp, N, out=parts, epilogue=ops.LT_PARTS, x_packed_mk=(M, K), tile_shape=542, k_split=4)),
                 ("tiles 10542 k4 parts (W through LDS)", lambda w, wp: ops.linear_tiles(xp, wp, N, out=parts, epilogue=ops.LT_PARTS, x_packed_mk=(M, K), tile_shape=10542, k_split=4))]
    else:
        sh = 542 if N == 4096 else 532
        cases = [("library", lambda w, wp: F.linear(x, w, b)),
                 (f"tiles {sh} bias", lambda w, wp: ops.linear_tiles(xp, wp, N, bias=b, out=out, x_packed_mk=(M, K), tile_shape=sh)),
                 (f"tiles {10000 + sh} bias (W through LDS)", lambda w, wp: ops.linear_tiles(xp, wp, N, bias=b, out=out, x_packed_mk=(M, K), tile_shape=10000 + sh))]
        if epi == ops.LT_QGELU:
            cases.append((f"tiles {sh} bias + QuickGELU", lambda w, wp: ops.linear_tiles(xp, wp, N, bias=b, out=out, epilogue=epi, x_packed_mk=(M, K), tile_shape=sh)))
    for label, fn in cases:
        marker = torch.zeros(16, 64, device=dev, dtype=dt)
        try:
            fn(ws[0], wps[0])  # (the "W through LDS" variants exist only in a library built with HIPCC_EXTRA=-DDL_LT_MEASURE)
        except ops.HipOpsError:
            continue
        ops.pack_x_tiles(marker)  # odd marker: what follows is this variant's warm-up: not counted
        fn(ws[0], wps[0])
        torch.cuda.synchronize()
        ops.pack_x_tiles(marker)  # even marker: the counted launches of this variant follow
        torch.cuda.synchronize()
        for w, wp in zip(ws[:16], wps[:16]):
            fn(w, wp)
        torch.cuda.synchronize()
        variants.append(f"{name}: {label}")
    del ws, wps
print("VARIANTS " + json.dumps(variants))
