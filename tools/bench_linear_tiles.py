#!/usr/bin/env python
"""dl_linear_tiles vs hipBLASLt on the CLIP tower's / projector's / vision predictor's GEMM shapes: graph-timed, weights rotated over 8 copies
(the 23 encoder layers stream 0.6 GB through a 256 MB cache in the real tower), per tile shape / k split.  --stamps: per-wave timeline of one launch."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dynamic_llava_amd import hip_ops as ops
dev, dt = "cuda", torch.bfloat16


def timed(fn, reps=48):
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


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--oproj", action="store_true")
args = ap.parse_args()
B = args.batch
T10 = [1042, 1032, 1041] if B > 1 else []
CASES = [("clip q|k|v", 577 * B, 3072, 1024, ops.LT_BIAS, [542, 532, 10532, 20532] + T10, [1]),
         ("clip out_proj", 577 * B, 1024, 1024, ops.LT_BIAS, [521, 10521, 20521, 522, 541, 20541] + T10, [1, 2]),
         ("clip fc1+qgelu", 577 * B, 4096, 1024, ops.LT_QGELU, [542, 10542, 20542] + T10, [1]),
         ("clip fc1 (bias only)", 577 * B, 4096, 1024, ops.LT_BIAS, [542, 10542, 20542] + T10, [1]),
         ("clip fc2", 577 * B, 1024, 4096, ops.LT_BIAS, [542, 10542, 20542, 522] + T10, [1, 2, 4]),
         ("projector 1+gelu", 576 * B, 4096, 1024, ops.LT_GELU, [542, 20542] + T10, [1]),
         ("projector 2", 576 * B, 4096, 4096, ops.LT_BIAS, [542, 10542, 20542] + T10, [1]),
         ("predictor in", 576 * B, 512, 4096, ops.LT_GELU, [521, 522], [2, 4]),
         ("predictor qkv", 576 * B, 1536, 512, ops.LT_BIAS, [522, 532], [1]),
         ("predictor fc1", 576 * B, 2048, 512, ops.LT_GELU, [522, 542], [1]),
         ("predictor fc2", 576 * B, 512, 2048, ops.LT_BIAS, [512, 521], [1, 2])]
for name, M, N, K, epi, shapes, splits in ([] if (args.stamps or args.oproj) else CASES):
    x = torch.randn(M, K, device=dev, dtype=dt)
    xp = ops.pack_x_rows(x)
    NW = max(8, int(300e6 / (N * K * 2)) + 1)  # more than the 256 MB cache: every launch streams its weights from HBM, as in the 23-layer tower
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(NW)]
    wps = [ops.pack_weight_tiles(w) for w in ws]
    b = torch.randn(N, device=dev, dtype=dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    fl = 2 * M * N * K
    t_l = timed(lambda i: F.linear(x, ws[i % NW], b))
    line = f"{name:18s} [{M},{K}]x[{N},{K}]: hipBLASLt {t_l:6.2f} us ({fl / t_l / 1e6:5.0f} TF/s) |"
    best = None
    for sh in shapes:
        for ks in splits:
            for packed in (1, 0):
                if ks > 1:
                    pbuf = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
                    f = lambda i: ops.linear_tiles(xp if packed else x, wps[i % NW], N, out=pbuf, epilogue=ops.LT_PARTS, x_packed_mk=(M, K) if packed else None, tile_shape=sh, k_split=ks)
                else:
                    f = lambda i: ops.linear_tiles(xp if packed else x, wps[i % NW], N, bias=b, out=y, epilogue=epi, x_packed_mk=(M, K) if packed else None, tile_shape=sh)
                try:
                    t = timed(f)
                except ops.HipOpsError as e:
                    continue
                line += f" {sh}/k{ks}/{'P' if packed else 'R'} {t:5.2f}"
                if best is None or t < best[0]: best = (t, sh, ks, packed)
    print(line + f" || best {best[0]:.2f} us = {fl / best[0] / 1e6:.0f} TF/s ({t_l / best[0]:.2f}x)", flush=True)

if args.stamps:
    # s_memtime counters of different XCDs are not synchronised: every wave is reported relative to ITS OWN entry stamp; ticks = shader clocks (~2.4 GHz)
    tick_us = 1 / 2400.0
    def timeline(name, M, N, K, epi, sh, ks, wrap=0, K_alloc=None):
        try:
            return _timeline(name, M, N, K, epi, sh, ks, wrap, K_alloc)
        except ops.HipOpsError as e:  # (the "W via LDS" shapes exist only in a library built with HIPCC_EXTRA=-DDL_LT_MEASURE)
            print(f"{name}: skipped ({str(e)[:120]})")

    def _timeline(name, M, N, K, epi, sh, ks, wrap=0, K_alloc=None):
        Ka = K_alloc or K
        x = torch.randn(M, Ka, device=dev, dtype=dt); xp = ops.pack_x_rows(x)
        w = torch.randn(N, Ka, device=dev, dtype=dt) * 0.02; wp = ops.pack_weight_tiles(w)
        b = torch.randn(N, device=dev, dtype=dt) if epi != ops.LT_PARTS else None
        junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        st = torch.zeros(1024 * 8 * 8, dtype=torch.int64, device=dev)
        junk.fill_(1)  # evict the weights
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear_tiles(xp, wp, N, bias=b, epilogue=epi, x_packed_mk=(M, K), tile_shape=sh, k_split=ks, stamps=st, _wrap=wrap)
        e1.record(); torch.cuda.synchronize()
        s_ = st.view(1024, 8, 8).cpu().double()
        live = s_[:, 4, 0] > 0
        s_ = s_[live]
        rel = (s_ - s_[:, :, :1]) * tick_us
        cons, load = rel[:, 4:4 + sh % 10000 // 10 % 10], rel[:, :4]
        print(f"{name} shape {sh} k{ks} wrap {wrap} ({int(live.sum())} workgroups; {e0.elapsed_time(e1) * 1e3:.1f} us between events): per wave from its own entry, us: "
              f"first step landed med {load[:, :, 1].median():.2f} max {load[:, :, 1].max():.2f} | loop start med {cons[:, :, 1].median():.2f} "
              f"half med {cons[:, :, 4].median():.2f} end med {cons[:, :, 2].median():.2f} max {cons[:, :, 2].max():.2f} | stores done med {cons[:, :, 3].median():.2f} max {cons[:, :, 3].max():.2f}", flush=True)
    timeline("fc1", 577, 4096, 1024, ops.LT_QGELU, 542, 1)
    timeline("fc1 bias", 577, 4096, 1024, ops.LT_BIAS, 542, 1)
    timeline("fc1 bias, K=2048 as two passes over K=1024 (second half: L2 hits)", 577, 4096, 2048, ops.LT_BIAS, 542, 1, wrap=16)
    timeline("fc1 bias, K=2048 as two passes, W via LDS", 577, 4096, 2048, ops.LT_BIAS, 10542, 1, wrap=16)
    timeline("fc1 bias, K=2048 real", 577, 4096, 2048, ops.LT_BIAS, 542, 1)
    timeline("fc1 bias, 80 rows only (32 workgroups)", 80, 4096, 1024, ops.LT_BIAS, 542, 1)
    timeline("fc1 bias, five steps of W ahead", 577, 4096, 1024, ops.LT_BIAS, 20542, 1)
    timeline("qkv", 577, 3072, 1024, ops.LT_BIAS, 532, 1)
    timeline("out", 577, 1024, 1024, ops.LT_BIAS, 521, 1)
    timeline("fc2", 577, 1024, 4096, ops.LT_PARTS, 542, 4)

if "--oproj" in sys.argv:
    # the decoder's o_proj [4096, 4096] at <= 256 rows: dl_linear_tiles' partial-sum form against what shipped before (dl_linear_splitk at prefill rows, dl_gemm_smallm at
    # decode batches), each WITH its residual-add / RMSNorm consumer, 32 layers' weights in one graph (cold)
    from dynamic_llava_amd.model import DynamicLlavaLlamaForCausalLM as M_
    H = 4096
    ws = [torch.randn(H, H, device=dev, dtype=dt) * 0.02 for _ in range(32)]
    wps = [ops.pack_weight_tiles(w) for w in ws]
    nw = torch.ones(H, device=dev, dtype=dt)
    for M in (170, 117, 192, 32, 16):
        x = torch.randn(M, H, device=dev, dtype=dt)
        h = torch.randn(M, H, device=dev, dtype=dt)
        out = torch.empty(M, H, device=dev, dtype=dt)
        buf = torch.empty(8 * 256 * H, device=dev, dtype=torch.float32)
        shp, ks = M_._tiles_o_config(M, H)
        res = {}
        def tiles(i):
            p_ = ops.linear_tiles(x, wps[i % 32], H, out=buf[: ks * M * H], epilogue=ops.LT_PARTS, tile_shape=shp, k_split=ks)
            ops.add_rmsnorm_parts(h, p_, nw, 1e-5, out=out)
        res[f"tiles {shp} k{ks}"] = timed(tiles, reps=32)
        res["tiles alone"] = timed(lambda i: ops.linear_tiles(x, wps[i % 32], H, out=buf[: ks * M * H], epilogue=ops.LT_PARTS, tile_shape=shp, k_split=ks), reps=32)
        if M <= 192:
            def splitk(i):
                p_ = ops.linear_splitk(x, ws[i % 32], buf, 8)
                ops.add_rmsnorm_parts(h, p_, nw, 1e-5, out=out)
            res["splitk s8"] = timed(splitk, reps=32)
        if M <= 32:
            wsm = torch.empty(8 * M * 32768, device=dev, dtype=torch.float32)
            def smallm(i):
                p_, _ = ops.gemm_smallm_parts(x, ws[i % 32], wsm)
                ops.add_rmsnorm_parts(h, p_, nw, 1e-5, out=out)
            res["gemm_smallm"] = timed(smallm, reps=32)
        def lib_(i):
            o = F.linear(x, ws[i % 32])
            ops.add_rmsnorm(h, o, nw, 1e-5, out=out)
        res["library"] = timed(lib_, reps=32)
        print(f"o_proj M={M:4d} (+ residual-add / RMSNorm consumer): " + "  ".join(f"{k} {v:6.2f} us" for k, v in res.items()), flush=True)
