#!/usr/bin/env python
"""Verdict r5 item 5, measured: gate|up [22016, 4096] at M = 170 / 117->(128) rows as 12 units per workgroup x 2 k ranges of fp32 partial sums + dl_silu_mul_parts,
against the shipped launch (6 units, one k range, SiLU * up in the epilogue).  30 layers' weight copies in one graph (cold weights).
    HIPCC_EXTRA=-DDL_LP_MEASURE_12U python -m dynamic_llava_amd.build_ext --force; python tools/bench_gate_up_12units.py   (the 12-unit instance is not in the product library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16


def timed(fn, reps=30):
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


H, I2, L = 4096, 22016, 12
ws = [torch.randn(I2, H, device=dev, dtype=dt) * 0.02 for _ in range(L)]
wp_pair = [ops.pack_weight_tiles(w, gate_up_pairs=True) for w in ws]
wp_plain = [ops.pack_weight_tiles(w) for w in ws]
for M in (170, 192, 144):
    x = torch.randn(M, H, device=dev, dtype=dt)
    xp = ops.pack_x_tiles(x)
    act = torch.empty(M, I2 // 2, device=dev, dtype=dt)
    parts = torch.empty(2, M, I2, device=dev, dtype=torch.float32)
    t_ship = timed(lambda i: ops.linear_packed(xp, wp_pair[i % L], I2, epilogue=ops.LP_SILU_PAIR, units_per_workgroup=6, k_split=1, x_packed_mk=(M, H), y_packed=True))
    t_g12 = timed(lambda i: ops.linear_packed(xp, wp_plain[i % L], I2, out=parts.view(-1), epilogue=ops.LP_PARTS, units_per_workgroup=12, k_split=2, x_packed_mk=(M, H)))
    t_g6 = timed(lambda i: ops.linear_packed(xp, wp_plain[i % L], I2, out=parts.view(-1), epilogue=ops.LP_PARTS, units_per_workgroup=6, k_split=1, x_packed_mk=(M, H)))
    t_c = timed(lambda i: ops.silu_mul_parts(parts, act))

    def both(i):
        ops.linear_packed(xp, wp_plain[i % L], I2, out=parts.view(-1), epilogue=ops.LP_PARTS, units_per_workgroup=12, k_split=2, x_packed_mk=(M, H))
        ops.silu_mul_parts(parts, act)
    t_b = timed(both)
    # check: same values as the shipped launch up to the fp32 summation order
    ref = ops.linear_packed(xp, wp_pair[0], I2, epilogue=ops.LP_SILU_PAIR, units_per_workgroup=6, k_split=1, x_packed_mk=(M, H))
    ops.linear_packed(xp, wp_plain[0], I2, out=parts.view(-1), epilogue=ops.LP_PARTS, units_per_workgroup=12, k_split=2, x_packed_mk=(M, H)); ops.silu_mul_parts(parts, act)
    err = float((act.float() - ref.float()).abs().max()) / float(ref.float().abs().max())
    print(f"M={M}: shipped (6 units, SiLU*up epilogue) {t_ship:.2f} us | 12 units x 2 ranges -> partial sums {t_g12:.2f} us (6 units x 1 range as partial sums {t_g6:.2f}) "
          f"+ dl_silu_mul_parts {t_c:.2f} = pair in one graph {t_b:.2f} us ({t_b / t_ship:.3f} x); max rel diff of act {err:.1e}", flush=True)
