#!/usr/bin/env python
"""dl_linear_packed (weights pre-packed in matrix-core operand order, LDS-DMA loader waves + MFMA consumer waves, k ranges shared between
workgroups) against the library GEMM on the decoder projections of the post-compaction prefill layers (M = 170 / 117) and of a 32-row decode step:
cold weights (NB rotating copies), hipGraph-timed.
  python tools/bench_linear_packed.py [--m 170,117,32] [--sweep] [--ablate]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops

ap = argparse.ArgumentParser()
ap.add_argument("--m", default="170,117,32")
ap.add_argument("--ablate", action="store_true")
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--model", default="7b")
ap.add_argument("--json", default="")
args = ap.parse_args()
dev, dt = "cuda", torch.bfloat16
NB = 6
H, I = (4096, 11008) if args.model == "7b" else (5120, 13824)
SHAPES = [("qkv", 3 * H, H, ops.LP_STORE), ("o", H, H, ops.LP_STORE), ("gate|up", 2 * I, H, ops.LP_STORE), ("gate|up+silu", 2 * I, H, ops.LP_SILU_PAIR), ("down", H, I, ops.LP_STORE)]
# (units per workgroup, k_split) candidates per shape
CAND = {"qkv": [(3, 1), (6, 2), (4, 2)], "o": [(1, 1), (2, 2), (4, 4), (2, 4)], "gate|up": [(6, 1), (4, 1)], "gate|up+silu": [(6, 1), (4, 1)],
        "down": [(1, 1), (2, 2), (4, 4), (2, 4)]}


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


rows = []
err_flag = torch.zeros(1, dtype=torch.int32, device=dev)
for M in [int(v) for v in args.m.split(",")]:
    for name, N, K, epi in SHAPES:
        pair = epi == ops.LP_SILU_PAIR
        ws = [torch.randn(N, K, device=dev, dtype=dt) / K**0.5 for _ in range(NB)]
        wps = [ops.pack_weight_tiles(w, gate_up_pairs=pair) for w in ws]
        x = torch.randn(M, K, device=dev, dtype=dt)
        xpk = ops.pack_x_tiles(x)
        n_out = N // 2 if pair else N
        out = torch.empty(M, n_out, device=dev, dtype=dt)
        if pair:
            t_lib = timed([lambda w=w: ops.silu_mul(F.linear(x, w)) for w in ws])
            ref32 = F.linear(x.float(), ws[0].float())
            g, u = ref32[:, : N // 2].to(dt).float(), ref32[:, N // 2 :].to(dt).float()
            ref = (F.silu(g).to(dt).float() * u)
        else:
            t_lib = timed([lambda w=w: F.linear(x, w) for w in ws])
            ref = F.linear(x.float(), ws[0].float())
        mb = N * K * 2 / 1e6
        rec = dict(M=M, name=name, N=N, K=K, weight_MB=mb, library_us=t_lib)
        line = f"M={M:3d} {name:13s} [{N},{K}] {mb:6.1f} MB: library {t_lib:6.2f}us ({mb / t_lib:4.2f} TB/s)"
        cands = CAND[name] if args.sweep else CAND[name][:2]
        for nu, ksp in cands:
            for xp, xbc in (((1, 0), (0, 0)) if args.sweep else ((1, 0),)):
                wsb = ops.linear_packed_workspace(M, N, K, dev, epi, nu, ksp)
                kw = dict(epilogue=epi, units_per_workgroup=nu, k_split=ksp, workspace=wsb, err=err_flag)
                xin, mk = (xpk, (M, K)) if xp else (x, None)
                try:
                    got = ops.linear_packed(xin, wps[0], N, x_packed_mk=mk, **kw).float()
                    got2 = ops.linear_packed(xin, wps[0], N, x_packed_mk=mk, **kw).float()
                except ops.HipOpsError as e:
                    line += f" | nu={nu} ks={ksp}: {e}"
                    continue
                err = float((got - ref).abs().max() / ref.abs().max())
                same = bool(torch.equal(got, got2))
                t = timed([lambda wp=wp: ops.linear_packed(xin, wp, N, out=out, x_packed_mk=mk, **kw) for wp in wps])
                rec[f"nu{nu}_ks{ksp}_xp{xp}_xbc{xbc}_us"] = t
                line += f" | nu={nu} ks={ksp}{'' if xp else ' rowX'}{' v1' if xbc else ''}: {t:6.2f}us ({mb / t:4.2f} TB/s, err {err:.1e}{'' if same else ' NONDET'})"
        if name in ("o", "down"):  # the product's form for the narrow projections: fp32 partial sums per k range, consumed by dl_add_rmsnorm_parts
            h0 = torch.randn(M, N, device=dev, dtype=dt)
            nw = torch.ones(N, device=dev, dtype=dt)
            parts8 = torch.empty(8 * M * N, device=dev, dtype=torch.float32)
            t_sk = timed([lambda w=w: ops.linear_splitk(x, w, parts8, 8) for w in ws])
            t_skc = timed([lambda w=w: ops.add_rmsnorm_parts(h0, ops.linear_splitk(x, w, parts8, 8), nw, 1e-5) for w in ws])
            line += f" | splitk s=8: {t_sk:6.2f}us (+ add_rmsnorm_parts {t_skc:6.2f})"
            rec["splitk8_us"], rec["splitk8_with_consumer_us"] = t_sk, t_skc
            for nu, ksp in ((4, 4), (2, 4), (4, 2), (2, 8) if K // 64 >= 8 else (4, 4)):
                pbuf = torch.empty(ksp * M * N, device=dev, dtype=torch.float32)
                f = lambda wp: ops.linear_packed(xpk, wp, N, out=pbuf, epilogue=ops.LP_PARTS, units_per_workgroup=nu, k_split=ksp, x_packed_mk=(M, K))
                got = f(wps[0]).sum(0)
                err = float((got - ref).abs().max() / ref.abs().max())
                t = timed([lambda wp=wp: f(wp) for wp in wps])
                tc = timed([lambda wp=wp: ops.add_rmsnorm_parts(h0, f(wp), nw, 1e-5) for wp in wps])
                rec[f"parts_nu{nu}_ks{ksp}_us"], rec[f"parts_nu{nu}_ks{ksp}_with_consumer_us"] = t, tc
                line += f" | PARTS nu={nu} ks={ksp}: {t:6.2f}us (+ consumer {tc:6.2f}, err {err:.1e})"
        if args.ablate and not pair and 128 < M <= 192:
            for ab, lab in ((1, "noMFMA"), (2, "noX"), (7, "loader-only")):
                for nu, ksp in [c for c in CAND[name] if c[0] in (3, 6)][:2]:
                    wsb = ops.linear_packed_workspace(M, N, K, dev, epi, nu, ksp)
                    try:
                        t = timed([lambda wp=wp: ops.linear_packed(xpk, wp, N, out=out, x_packed_mk=(M, K), units_per_workgroup=nu, k_split=ksp, workspace=wsb, _ablate=ab) for wp in wps])
                        line += f" | {lab} nu={nu} ks={ksp}: {t:6.2f}"
                    except ops.HipOpsError:
                        line += f" | {lab} nu={nu}: ERR"
        print(line, flush=True)
        rows.append(rec)
        del ws, wps
print("err_flag", int(err_flag.item()))
if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
