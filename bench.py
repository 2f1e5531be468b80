#!/usr/bin/env python
"""bench.py -- BASELINE.json metric "prefill+decode tokens/s/GPU, LLaVA-1.5-7B @ vision_keep_rate=0.2, 1 img".

One "step" = one full `generate()` of the hot path on one request batch, inputs already resident in HBM:
CLIP ViT-L/14-336 + projector -> 32-layer LLaVA-1.5-7B sparsified prefill (576 image tokens -> 115 after layer 2)
-> T_new greedy decode tokens with output-text KV eviction.  Workload = BASELINE.json configs[1]: random-init
LLaVA-1.5-7B, bf16, B=1, one synthetic 336x336 image, prompt = 35 system tokens + <image> + 20 question tokens
(N = 631 -> N' = 170), vision_keep_rate=0.2.  tokens/step = N + T_new (the reference's own accounting counts all
576 image tokens: llava/dynamic_eval/bench_test/dynamic_llava_long_text_mem.py:317-323).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

    python bench.py --gpus N ...        (no launcher: bench.py starts the N ranks itself through torch.distributed.run)

N > 1: weak scaling, one identical request per rank (weights replicated), one RCCL all-gather (a single
all_gather_into_tensor carrying the last-token logits + generated ids + their shapes) per step, no collective in the decode loop.  The ranks' results must be identical (FATAL otherwise).
After the timed region an N > 1 run also executes BASELINE configs[3] once (32 ragged requests per rank, one all-gather) and checks
the gathered rows against a re-run of another rank's chunk.  Rank 0 prints ONE JSON line.

The output-text predictor is random-init here: its keep/evict bit would saturate one way, so its final bias is calibrated on the
workload itself (median of the keep-logit gap over a teacher-forced pass) until about half of the generated tokens are evicted from
layers >= 2 -- the regime `output_text_keep_rate=0.5` names; `phases.evicted/generated` reports what the timed run did.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
N_SYS, N_Q, N_IMG = 35, 20, 576


METRIC = "prefill+decode tokens/s/GPU, LLaVA-1.5-7B @ vision_keep_rate=0.2, 1 img"


class BenchAbort(Exception):
    """A run that must not report a number.  main() turns it into ONE JSON line with an "error" key (what a driver parses), then exits non-zero."""

    def __init__(self, msg, **detail):
        super().__init__(msg)
        self.detail = detail


def abort_line(msg, detail, partial):
    """The JSON line of an aborted run: same leading keys as a result line (so that a parser finds `metric` / `n_gpus`), `value` null, the reason, the
    per-rank detail and whatever was measured before the abort."""
    return json.dumps({"metric": METRIC, "value": None, "unit": "tokens/s", "error": msg, "detail": detail, "partial": partial})


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer decoder layers => the JSON is marked INVALID")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-gpu", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--predictor-gain", type=float, default=50.0, help="'trained-like' predictor scaling (no score ties); 1.0 = plain random init")
    ap.add_argument("--no-calibrate", action="store_true", help="leave the random-init output-text predictor as is (it then keeps or evicts everything)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the untimed configs[2] / configs[4] legs and the measured ceilings (N = 1 only; they add ~2 minutes)")
    return ap.parse_args()


def make_inputs(cfg, device, dtype):
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, cfg.vocab_size, (N_SYS + N_Q,), generator=g)
    prompt = torch.cat([torch.tensor([1]), ids[: N_SYS - 1], torch.tensor([-200]), ids[N_SYS:]]).long()[None]
    images = torch.randn((1, 3, 336, 336), generator=g).to(dtype)
    return prompt.to(device), images.to(device)


def event_time_ms(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def graph_time_ms(fn, reps=40, replays=8):
    """Device-side time of one `fn()` launch: `reps` launches captured into ONE hipGraph (no Python / ctypes / launch-API
    time inside the timed region), graph replayed `replays` times between HIP events on the launch stream.  Includes the
    ~1-2 us dependent-launch gap between consecutive kernels, so it is an upper bound on the pure kernel duration that
    rocprofv3 reports."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * replays)


def decode_attn_roofline(model, label, B, T_list, n_heads, head_dim):
    """dl_attn_decode_rope (the decode step's attention: fused RoPE + KV append + ragged split-KV attention; + the combine
    kernel when n_splits > 1) on a [B, nH, T_cap, d] slab with per-row lengths T_list (incl. the new token).
    Algorithmic bytes per launch: sum_b [2*T_b*H*E + 2*H*E] (+ the q|k|v row and the appended K/V), E = 2."""
    from dynamic_llava_amd import hip_ops as ops
    from dynamic_llava_amd.cache import KVSlabCache

    dev, dt = model.device, model.dtype
    H = n_heads * head_dim
    T_cap = max(T_list) + 1
    k = torch.randn((B, n_heads, T_cap, head_dim), device=dev, dtype=dt)
    v = torch.randn_like(k)
    qkv = torch.randn((B, 3 * H), device=dev, dtype=dt)
    out = torch.empty((B, H), device=dev, dtype=dt)
    lens = torch.tensor([t - 1 for t in T_list], dtype=torch.int32, device=dev)
    cos, sin = model._rope_tables(T_cap + 1)
    n_splits = 1 if T_cap <= 256 else max(1, min(32, max(1, 256 // (B * n_heads)), -(-T_cap // 64)))  # same rule as KVSlabCache.n_splits
    ws = ops.attn_decode_workspace(B, n_heads, head_dim, 32, dev)
    # rotate over several slabs so that, as in the real decode step (13 GB of weights streamed in between), K/V come from HBM
    n_buf = 8 if k.numel() * 4 < 200e6 else 1
    ks = [k] + [torch.randn_like(k) for _ in range(n_buf - 1)]
    vs = [v] + [torch.randn_like(v) for _ in range(n_buf - 1)]
    it = [0]

    tagc = [0]

    def launch():
        i = it[0] = (it[0] + 1) % n_buf
        tagc[0] = (tagc[0] + 1) % 251  # as in the decode step: consecutive launches sharing the workspace carry different call tags
        ops.attn_decode_rope(qkv, cos, sin, lens, lens, ks[i], vs[i], out, ws, n_splits, n_heads, n_heads, head_dim, chunk_keys=KVSlabCache.spec_chunk(n_splits), call_tag=tagc[0])

    ms = graph_time_ms(launch)
    nbytes = sum(2 * t * H * 2 + 2 * H * 2 for t in T_list)
    gbs = nbytes / (ms * 1e-3) / 1e9
    name = "dl_attn_decode_rope (attn_decode_split_kernel<bf16,128,4,fused,4" + (",in-kernel combine>)" if 1 < n_splits and n_splits * n_heads * B <= 1024 else ">)")
    return {"kernel": name, "shape": label, "n_splits": n_splits, "bytes": nbytes, "us": round(ms * 1e3, 3), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}


def linear_packed_roofline(model):
    """dl_linear_packed (round 5) on the post-compaction prefill shapes (M = 170) and the 32-row decode step's MLP, walking the 30 layers' operand-order weight
    copies in one hipGraph (cold weights), between events on the launch stream.  Algorithmic bytes: the weights + X + Y."""
    from dynamic_llava_amd import hip_ops as ops

    layers = [l for l in model.model.layers[2:] if getattr(l, "wp_qkv", None) is not None and getattr(l, "wp_down", None) is not None]
    if not layers:
        return []
    dev, dt = model.device, model.dtype
    H, I = model.config.hidden_size, model.config.intermediate_size
    Nq = layers[0].w_qkv.shape[0]
    out = []
    for M in (170, 32):
        x = torch.randn((M, H), device=dev, dtype=dt)
        xi = torch.randn((M, I), device=dev, dtype=dt) * 0.1
        xp, xip = ops.pack_x_tiles(x), ops.pack_x_tiles(xi)
        nu_q, ks_q = model._lp_config(Nq // 16, False)
        nu_g, ks_g = model._lp_config(2 * I // 16, True)
        nu_d, ks_d = model._lp_config_parts(H // 16, M)
        y_q = torch.empty((M, Nq), device=dev, dtype=dt)
        act = torch.empty(int(ops.lib().dl_packed_x_bytes(M, I)) // 2, device=dev, dtype=dt)
        parts = torch.empty(ks_d * M * H, device=dev, dtype=torch.float32)
        parts_q = torch.empty(ks_q * M * Nq, device=dev, dtype=torch.float32)
        cases = [
            (f"q|k|v ({nu_q} units x {ks_q} k ranges handed over in the launch: decode batches of 16..32 rows)", [Nq, H], Nq * H * 2 + M * H * 2 + M * Nq * 2,
             lambda l: ops.linear_packed(xp, l.wp_qkv, Nq, out=y_q, units_per_workgroup=nu_q, k_split=ks_q, workspace=model._lp_ws if ks_q > 1 else None, err=model._lp_err, x_packed_mk=(M, H))),
            (f"q|k|v, fp32 partial sums for the RoPE / KV-append launch ({nu_q} units x {ks_q} k ranges: the prefill's path)", [Nq, H], Nq * H * 2 + M * H * 2 + ks_q * M * Nq * 4,
             lambda l: ops.linear_packed(xp, l.wp_qkv, Nq, out=parts_q, epilogue=ops.LP_PARTS, units_per_workgroup=nu_q, k_split=ks_q, x_packed_mk=(M, H))),
            (f"gate|up + SiLU*up epilogue, act in fragment order ({nu_g} units)", [2 * I, H], 2 * I * H * 2 + M * H * 2 + M * I * 2,
             lambda l: ops.linear_packed(xp, l.wp_gu, 2 * I, out=act, epilogue=ops.LP_SILU_PAIR, units_per_workgroup=nu_g, k_split=ks_g, workspace=model._lp_ws if ks_g > 1 else None, err=model._lp_err,
                                         x_packed_mk=(M, H), y_packed=True)),
            (f"down_proj, fp32 partial sums ({nu_d} units x {ks_d} k ranges)", [H, I], H * I * 2 + M * I * 2 + ks_d * M * H * 4,
             lambda l: ops.linear_packed(xip, l.wp_down, H, out=parts, epilogue=ops.LP_PARTS, units_per_workgroup=nu_d, k_split=ks_d, x_packed_mk=(M, I))),
        ]
        for name, shape, nbytes, fn in cases:
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                fn(layers[0])
            torch.cuda.current_stream().wait_stream(s_)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for l in layers:
                    fn(l)
            g.replay()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(5):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / (5 * len(layers)) * 1e3
            out.append({"kernel": "dl_linear_packed " + name, "shape": f"M={M} x {shape} bf16, {len(layers)} layers' weight copies in one graph", "bytes": nbytes, "us": round(us, 3),
                        "achieved": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)})
    model.check_device_errors()
    return out


def linear_tiles_roofline(model):
    """dl_linear_tiles (round 6) on the CLIP tower's four projections at one image (M = 577), walking the 23 encoder layers' operand-order weight copies in one
    hipGraph (cold weights: 0.58 GB through a 256 MB cache), between events on the launch stream.  These launches are MFMA-shaped but small (1.2-4.8 GFLOP): the
    entry reports TFLOP/s against the dense bf16 MFMA peak AND the bytes every CU pulls through its vector-memory path (2 KiB per row tile / unit per 64-k step),
    which is what bounds the k loop (~29 cycles per 1-KiB wave request per CU, whether it is an LDS-DMA piece or a register load)."""
    from dynamic_llava_amd import hip_ops as ops

    vt = model.get_vision_tower()
    tiles = [t for t in (getattr(vt, "_tiles", None) or []) if t is not None] if vt is not None else []
    if not tiles:
        return []
    dev, dt = model.device, model.dtype
    c = model.config.clip
    C, I = c["hidden_size"], c["intermediate_size"]
    M = (c["image_size"] // c["patch_size"]) ** 2 + 1
    x = torch.randn((M, C), device=dev, dtype=dt)
    xi = torch.randn((M, I), device=dev, dtype=dt) * 0.1
    xp, xip = ops.pack_x_rows(x), ops.pack_x_rows(xi)
    bq, b1 = torch.zeros(3 * C, device=dev, dtype=dt), torch.zeros(I, device=dev, dtype=dt)
    y_q = torch.empty((M, 3 * C), device=dev, dtype=dt)
    g_pk = torch.empty(ops.tiles_x_numel(M, I), device=dev, dtype=dt)
    ks_o, ks_2 = vt.tiles_ksplit_out, vt.tiles_ksplit_fc2
    p_o = torch.empty((ks_o, M, C), device=dev, dtype=torch.float32)
    p_2 = torch.empty((ks_2, M, C), device=dev, dtype=torch.float32)
    row_tiles = (M + 15) // 16
    cases = [
        ("q|k|v + bias (5 row tiles x 6 units per workgroup)", 3 * C, C, 1, (5, 6), lambda t: ops.linear_tiles(xp, t[0], 3 * C, bias=bq, out=y_q, x_packed_mk=(M, C))),
        (f"out_proj, fp32 partial sums of {ks_o} k ranges (5 x 4)", C, C, ks_o, (5, 4), lambda t: ops.linear_tiles(x, t[1], C, out=p_o, epilogue=ops.LT_PARTS, k_split=ks_o)),
        ("fc1 + bias + QuickGELU epilogue, fragment-order output (5 x 8)", I, C, 1, (5, 8), lambda t: ops.linear_tiles(xp, t[2], I, bias=b1, out=g_pk, epilogue=ops.LT_QGELU, x_packed_mk=(M, C), y_packed=True)),
        (f"fc2, fp32 partial sums of {ks_2} k ranges (5 x 8)", C, I, ks_2, (5, 8), lambda t: ops.linear_tiles(xip, t[3], C, out=p_2, epilogue=ops.LT_PARTS, k_split=ks_2, x_packed_mk=(M, I))),
    ]
    out = []
    for name, N, K, ks, (tm, nu), fn in cases:
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            fn(tiles[0])
        torch.cuda.current_stream().wait_stream(s_)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in tiles:
                fn(t)
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / (5 * len(tiles)) * 1e3
        flops = 2 * M * N * K
        n_wg = -(-row_tiles // tm) * -(-(N // 16) // nu) * ks
        cu_bytes = n_wg * (tm + nu) * 2048 * (K // 64 // ks)  # what the workgroups pull through their CUs' vector-memory paths: X tiles + W units, 2 KiB each per step
        out.append({"kernel": "dl_linear_tiles " + name, "shape": f"M={M} x [{N}, {K}] bf16, {len(tiles)} layers' weight copies in one graph", "bound": "mfma", "flops": flops, "us": round(us, 3),
                    "achieved": round(flops / us / 1e6, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / us / 1e6 / 2500.0, 4), "workgroups": n_wg,
                    "bytes_into_the_CUs": cu_bytes, "cu_fill_GBps_over_the_whole_launch": round(cu_bytes / us / 1e3, 1),
                    "cu_fill_bytes_per_clk_per_cu": round(cu_bytes / min(n_wg, 256) / (us * 2400.0), 2), "hbm_bytes": N * K * 2 + M * K * 2 + (ks * M * N * 4 if ks > 1 or "partial" in name else M * N * 2)})
    return out


def gemv_roofline(model):
    """dl_gemv on the decode step's four weight shapes, walking all 32 layers' weights (13 GB >> 256 MB Infinity Cache)."""
    from dynamic_llava_amd import hip_ops as ops

    dev, dt = model.device, model.dtype
    cfg = model.config
    H, I = cfg.hidden_size, cfg.intermediate_size
    h, h2, dl = (torch.randn((1, H), device=dev, dtype=dt) for _ in range(3))
    x = torch.randn((1, H), device=dev, dtype=dt)
    xi = torch.randn((1, I), device=dev, dtype=dt)
    nw = torch.ones(H, device=dev, dtype=dt)
    res = []
    layers = model.model.layers
    y_qkv = torch.empty((1, layers[0].w_qkv.shape[0]), device=dev, dtype=dt)
    y_h = torch.empty((1, H), device=dev, dtype=dt)
    y_i = torch.empty((1, I), device=dev, dtype=dt)
    cases = [
        ("qkv (add+rmsnorm prologue)", lambda l: ops.gemv(l.w_qkv, y_qkv, mode=ops.GEMV_ADDNORM, h_in=h, h_out=h2, delta=dl, norm_w=nw, eps=1e-5), lambda l: l.w_qkv),
        ("o_proj", lambda l: ops.gemv(l.self_attn.o_proj.weight, y_h, x=x), lambda l: l.self_attn.o_proj.weight),
        ("gate|up (add+rmsnorm prologue, silu*up epilogue)", lambda l: ops.gemv(l.w_gu, y_i, mode=ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR, h_in=h, h_out=h2, delta=dl, norm_w=nw, eps=1e-5), lambda l: l.w_gu),
        ("down_proj", lambda l: ops.gemv(l.mlp.down_proj.weight, y_h, x=xi), lambda l: l.mlp.down_proj.weight),
    ]
    lm = model.lm_head.weight
    y_v = torch.empty((1, lm.shape[0]), device=dev, dtype=dt)

    def whole_step():  # the 129 weight-streaming launches of one decode step, in order, on the real weights
        for l in layers:
            for _, fn, _ in cases:
                fn(l)
        ops.gemv(lm, y_v, mode=ops.GEMV_ADDNORM, h_in=h, h_out=h2, delta=dl, norm_w=nw, eps=1e-5)

    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        whole_step()
    torch.cuda.current_stream().wait_stream(s_)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        whole_step()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    n_launch = 4 * len(layers) + 1
    step_us = a.elapsed_time(b) / 5 * 1e3
    step_bytes = sum(wsel(l).numel() * 2 for l in layers for _, _, wsel in cases) + lm.numel() * 2
    agg = {"kernel": "dl::gemv_kernel -- dl_gemv, all weight-streaming launches of one decode step (q|k|v, o, gate|up, down per layer + lm_head)",
           "launches_per_step": n_launch, "bytes": int(step_bytes / n_launch), "us": round(step_us / n_launch, 3),
           "achieved": round(step_bytes / step_us / 1e3, 1), "frac": round(step_bytes / step_us / 1e3 / HBM_PEAK_GBS, 4),
           "bytes_per_step": step_bytes, "us_per_step": round(step_us, 1)}
    for name, fn, wsel in cases:
        fns = [lambda l=l: fn(l) for l in layers]
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            fns[0]()
        torch.cuda.current_stream().wait_stream(s_)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for f in fns:
                f()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / (5 * len(fns)) * 1e3
        nb = wsel(layers[0]).numel() * 2
        res.append({"kernel": "dl_gemv " + name, "shape": f"{list(wsel(layers[0]).shape)} bf16, B=1", "bytes": nb, "us": round(us, 3), "achieved": round(nb / us / 1e3, 1), "frac": round(nb / us / 1e3 / HBM_PEAK_GBS, 4)})
    return agg, res


def fused_launch_roofline(model):
    """The PRODUCT's q|k|v launch of a batch-1 decode layer whose rows run one attention workgroup per head: dl_gemv_qkv_attn (projection with the
    residual-add + RMSNorm prologue AND RoPE / KV append / attention over the slab) on the real weights of every such layer and the K/V the last
    generate() left in the slab, captured in one hipGraph and timed between HIP events.  Algorithmic bytes per launch = the layer's q|k|v weights
    + 2 T H E of K/V + the appended row (SURVEY 8d's decode_attn bytes)."""
    from dynamic_llava_amd import hip_ops as ops

    st, cache = model._dstate, model.last_cache
    if st is None or cache is None or st.B != 1 or st.qa_gran is None:
        return None
    cfg = model.config
    nH, nKV, d, SL = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.sparse_config["sparse_layer"]
    cos, sin = model._rope
    layers = list(enumerate(model.model.layers))[SL:]
    T = int(cache.lens[1][0]) + 1
    if T > cache.single_split_max_keys or not layers:
        return None
    delta = torch.randn_like(st.h)

    def step():
        for i, layer in layers:
            ops.gemv_qkv_attn(layer.w_qkv, st.qkv, st.h, st.h2, delta, layer.input_layernorm.weight, cfg.rms_norm_eps, cos, sin, cache.len_full, cache.lens[1], cache.k[i], cache.v[i],
                              st.attn, st.qa_gran, i & 0xff, nH, nKV, d, err=st.blk_err, n_splits=cache.fused_attn_splits(i, model.fused_attn_max_splits))

    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        step()
    torch.cuda.current_stream().wait_stream(s_)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    model.check_device_errors()
    us = a.elapsed_time(b) / (5 * len(layers)) * 1e3
    H = nH * d
    nb = layers[0][1].w_qkv.numel() * 2 + 2 * T * nKV * d * 2 + 2 * H * 2
    return {"kernel": "dl_gemv_qkv_attn (gemv_qkv_attn_kernel: the product's q|k|v + attention launch of layers >= sparse_layer at batch 1)",
            "shape": f"{list(layers[0][1].w_qkv.shape)} bf16 + K/V of T={T} keys, {len(layers)} layers' weights and slabs in one graph", "bytes": nb, "us": round(us, 3),
            "achieved": round(nb / us / 1e3, 1), "frac": round(nb / us / 1e3 / HBM_PEAK_GBS, 4)}


def other_kernel_rooflines(model, n_tokens):
    """HBM-bound row kernels on the workload's prefill shape (algorithmic bytes per SURVEY 8d)."""
    from dynamic_llava_amd import hip_ops as ops

    dev, dt, H = model.device, model.dtype, model.config.hidden_size
    res = []
    x = torch.randn((n_tokens, H), device=dev, dtype=dt)
    w = torch.ones(H, device=dev, dtype=dt)
    out = torch.empty_like(x)
    ms = graph_time_ms(lambda: ops.rmsnorm(x, w, 1e-5, out=out))
    nb = 2 * n_tokens * H * 2 + H * 2
    res.append({"kernel": "dl_rmsnorm", "shape": f"[{n_tokens},{H}]", "bytes": nb, "us": round(ms * 1e3, 3), "achieved": round(nb / ms / 1e6, 1), "frac": round(nb / ms / 1e6 / HBM_PEAK_GBS, 4)})
    return res


def _pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]


def cpu_baseline(model, prompt, images, new_tokens):
    """The oracle (CPU restatement of the reference path, oracle/ref_cpu.py, bf16) on this box's host cores at FULL depth: the bench
    model's own 32 distinct layers (state dict copied to the host), the bench prompt.  Bounded sample, SURVEY 8d protocol scaled to
    ~25 s of CPU work: CLIP+projector+prefill 1 warm-up + 3 timed repetitions, decode 1 warm-up + 10 timed tokens; medians (p10 / p90
    reported), value = (N + T_new) / (median prefill + (T_new - 1) x median decode step)."""
    from oracle.ref_cpu import Oracle

    threads = torch.get_num_threads()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if "vision_tower" not in k}
    import copy as _copy

    clip = _copy.deepcopy(model.model.vision_tower.vision_tower).cpu()
    o = Oracle(model.config, sd, torch.bfloat16, clip=clip)
    ids, img = prompt.cpu(), images.cpu()
    pre, dec = [], []
    pkv = logits = None
    with torch.no_grad():
        for rep in range(4):
            t0 = time.perf_counter()
            feats = o.encode_images(img)
            logits, pkv = o.forward(ids, image_features=feats)
            if rep > 0:
                pre.append(time.perf_counter() - t0)
        for j in range(11):
            t0 = time.perf_counter()
            logits, pkv = o.forward(logits[:, -1:].argmax(-1), past_key_values=pkv)
            if j > 0:
                dec.append(time.perf_counter() - t0)
    n_prompt = N_SYS + N_IMG + N_Q
    p50, d50 = _pct(pre, 0.5), _pct(dec, 0.5)
    total = p50 + (new_tokens - 1) * d50
    return {
        "value": round((n_prompt + new_tokens) / total, 2), "unit": "tokens/s", "cores": threads, "kind": "port",
        "sample": f"oracle/ref_cpu.py bf16, the bench model's 32 distinct layers + CLIP on the host, bench prompt N={n_prompt}: CLIP+projector+prefill "
                  f"1 warm-up + {len(pre)} reps, decode 1 warm-up + {len(dec)} tokens; medians; step = prefill + {new_tokens - 1} decode steps",
        "prefill_s": {"median": round(p50, 3), "p10": round(_pct(pre, 0.1), 3), "p90": round(_pct(pre, 0.9), 3)},
        "decode_s_per_token": {"median": round(d50, 4), "p10": round(_pct(dec, 0.1), 4), "p90": round(_pct(dec, 0.9), 4)},
        "prefill_tokens_per_s": round(n_prompt / p50, 2), "decode_tokens_per_s": round(1.0 / d50, 3),
    }


def ref_gpu_path(model, prompt, images, new_tokens):
    """Stand-in for "the reference GPU path" (the reference's Python cannot travel to this box): the oracle -- the same
    eager op sequence as the reference (padded batch, torch.cat KV cache, per-layer host sync on the eviction
    decision, SDPA) -- run by PyTorch-ROCm on this GPU with this model's weights.  Reported baseline only."""
    from oracle.ref_cpu import Oracle

    sd = {k: v for k, v in model.state_dict().items() if "vision_tower" not in k}
    cfg = model.config
    o = Oracle(cfg, sd, model.dtype, device=str(model.device), clip=model.model.vision_tower.vision_tower)
    n_prompt = N_SYS + N_IMG + N_Q

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o.greedy(prompt, images=images, max_new_tokens=n, eos_token_id=None)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(2)
    t_full = min(run(new_tokens) for _ in range(2))
    t_pre = min(run(1) for _ in range(3))
    return {"value": round((n_prompt + new_tokens) / t_full, 1), "unit": "tokens/s", "kind": "oracle op sequence on PyTorch-ROCm eager (reference GPU-path stand-in)",
            "prefill_tokens_per_s": round(n_prompt / t_pre, 1), "decode_tokens_per_s": round((new_tokens - 1) / max(t_full - t_pre, 1e-9), 2), "ms_per_step": round(t_full * 1e3, 2)}


@torch.no_grad()
def calibrate_text_predictor(model, prompt, images, n_tokens):
    """Shift the final bias of the (random-init) output-text predictor so that about half of the tokens of THIS workload are evicted:
    gap_j = keep logit - evict logit of decode step j along the model's own greedy continuation; bias[0] -= median(gap).  A few
    rounds, because evicting tokens changes the later hidden states.  Deterministic (same weights / inputs on every rank)."""
    tp = getattr(model.model, "output_text_score_predictor", None)
    if tp is None:
        return None
    last = tp.output_mlp[7]
    frac = None
    for _ in range(4):
        model.debug_records = {}
        out = model(prompt, images=images)
        pkv = out.past_key_values
        tok = out.logits[:, -1].argmax(-1)
        gaps = []
        for _j in range(n_tokens - 1):
            out = model(tok[:, None], past_key_values=pkv)
            pkv = out.past_key_values
            tl = model.debug_records["text_logit"]
            gaps.append(float(tl[0, 0] - tl[0, 1]))
            tok = out.logits[:, -1].argmax(-1)
        model.debug_records = None
        frac = sum(g > 0 for g in gaps) / max(len(gaps), 1)
        if 0.4 <= frac <= 0.6:
            break
        med = sorted(gaps)[len(gaps) // 2]
        last.bias.data[0] -= torch.tensor(med, dtype=last.bias.dtype, device=last.bias.device)
    model._dstate = None  # predictor weights are read through cached pointers: nothing to rebuild, but captured graphs are re-made
    model._prefill_graphs = {}
    return frac


def bimg_prompt_phase(model, images, reps=20):
    """The reference's prefill harness (llava/dynamic_eval/bench_test/dynamic_llava_image_time_and_mem.py:124-151): input_ids
    [[1, -200, 1]] (N = 578 -> 117 after layer 2), generate(max_new_tokens=1, min_new_tokens=1), event pairs around each call."""
    ids = torch.tensor([[1, -200, 1]], device=images.device)
    ts = []
    for i in range(reps + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        model.generate(ids, images=images, do_sample=False, num_beams=1, use_cache=True, min_new_tokens=1, max_new_tokens=1)
        b.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b))
    return {"prompt": "[[1,-200,1]] (BIMG:124), N=578 -> 117", "reps": reps, "prefill_ms_median": round(_pct(ts, 0.5), 3), "p10": round(_pct(ts, 0.1), 3),
            "p90": round(_pct(ts, 0.9), 3), "prefill_tokens_per_s": round(578 / _pct(ts, 0.5) * 1e3, 1)}


def configs3_leg(model, cfg, dd, rank, world, device, dtype, new_tokens=32):
    """BASELINE configs[3]: 32 ragged requests per rank (256 over 8 GPUs), contiguous chunks (model_vqa_loader.py:30-38), one all-gather of
    last-token logits + ids.  Rank 0 then re-runs the LAST rank's chunk and compares it with the gathered rows: generated ids that differ
    are FATAL (round 3: the drift seen when two ranks shared one GPU was a gfx950 packed-fp32 instruction form that is mis-executed beside
    another kernel's MFMA waves -- DESIGN.md section 5 -- and the library no longer contains it); prefill logits may differ in the last bit
    when the ranks sit on different devices (the library's B=32 prefill GEMMs), which is reported in `rerun_detail`."""
    per = 32
    g = torch.Generator().manual_seed(1)
    n_req = per * world
    n_q = torch.randint(8, 65, (n_req,), generator=g).tolist()

    def chunk(r):
        idx = list(dd.get_chunk(list(range(n_req)), world, r))
        gi = torch.Generator().manual_seed(100 + r)
        W = max(35 + 1 + n_q[i] for i in idx)
        ids = torch.zeros(len(idx), W, dtype=torch.long)
        am = torch.zeros(len(idx), W, dtype=torch.long)
        for row, i in enumerate(idx):
            gq = torch.Generator().manual_seed(1000 + i)
            body = torch.randint(3, cfg.vocab_size, (35 + n_q[i],), generator=gq)
            p = torch.cat([torch.tensor([1]), body[:34], torch.tensor([-200]), body[35:]])
            ids[row, : p.numel()] = p
            am[row, : p.numel()] = 1
        imgs = torch.randn((len(idx), 3, 336, 336), generator=gi).to(dtype)
        return ids.to(device), am.to(device), imgs.to(device), sum(35 + 576 + n_q[i] for i in idx)

    ids, am, imgs, n_prompt_tok = chunk(rank)

    def run(i_, a_, im_):
        out = model.generate(i_, attention_mask=a_, images=im_, max_new_tokens=new_tokens, eos_token_id=None)
        return out, model.last_prefill_logits.float().clone()

    run(ids, am, imgs)  # warm-up (graph capture, allocator)
    dd.barrier()
    torch.cuda.synchronize()
    import torch.distributed as tdist

    n_coll = []
    real_ag = tdist.all_gather_into_tensor if tdist.is_initialized() else None
    if real_ag is not None:  # count what the batch's gather really issues (one collective per batch, asserted by the 8-rank test)
        tdist.all_gather_into_tensor = lambda *a, **k: (n_coll.append(1), real_ag(*a, **k))[1]
    coll = {}
    t0 = time.perf_counter()
    out, lg = run(ids, am, imgs)
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0  # this rank's generate() alone
    all_lg, all_ids = dd.gather_results(lg, out, max_rows=per, max_new_tokens=new_tokens, timing=coll)  # one all_gather_into_tensor: logits + ids + shapes
    if real_ag is not None:
        tdist.all_gather_into_tensor = real_ag
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    dd.barrier()
    el = dd.max_over_ranks(time.perf_counter() - t0, device)
    own_min, own_max = dd.min_max_over_ranks(own, device)
    cmp_min, cmp_max = dd.min_max_over_ranks(t_compute, device)
    coll_min, coll_max = dd.min_max_over_ranks(float(coll.get("collective_us") or 0.0), device)
    tot = dd.max_over_ranks(float(n_prompt_tok), device)  # not summed: report per-rank max and the global count below
    ok, detail = True, None
    if rank == 0 and world > 1:
        r = world - 1
        i2, a2, im2, _ = chunk(r)
        o2, l2 = run(i2, a2, im2)
        ids_r, lg_r = all_ids[r * per:(r + 1) * per], all_lg[r * per:(r + 1) * per]
        ok = bool(torch.equal(ids_r, o2) and torch.equal(lg_r, l2))
        detail = {"rows_with_different_ids": int((ids_r != o2).any(dim=1).sum()), "max_abs_prefill_logit_diff": float((lg_r - l2).abs().max())}
        if detail["rows_with_different_ids"] and detail["max_abs_prefill_logit_diff"] == 0.0:
            # identical prefill (every K/V row of every layer feeds those logits) but different tokens: the DECODE diverged between two devices
            # running the same kernels on the same data -- a bug (round 3's packed-fp32 anomaly looked exactly like this), never a rounding matter
            raise BenchAbort(f"configs[3]: rank {r}'s gathered rows differ from rank 0's re-run of the same chunk after a bit-identical prefill", **detail)
        # a last-bit difference of the prefill logits (another device's library GEMM picked another kernel) can legitimately flip a near-tied
        # argmax later: reported in rerun_detail, not fatal
    n_tok_all = sum(35 + 576 + q for q in n_q) + n_req * new_tokens
    return {"workload": f"BASELINE configs[3]: {n_req} ragged requests ({per} per rank, question lengths ~U[8,64]), {new_tokens} new tokens each, one all-gather",
            "tokens_per_s": round(n_tok_all / el, 1), "seconds": round(el, 4), "gathered_rows": int(all_ids.shape[0]), "collectives_per_batch": len(n_coll),
            "rows_per_rank": [len(dd.get_chunk(list(range(n_req)), world, r_)) for r_ in range(world)], "dp_equals_rerun_of_last_rank": ok,
            "rerun_detail": detail, "max_prompt_tokens_per_rank": int(tot),
            # the one collective by itself, and the per-rank spread -- to be read against SURVEY 8e's estimate for 4.1 MB per rank
            # (~27 us direct over the 7 xGMI links, ~190 us for a per-link-bound ring); a rank that arrives late makes the others' collective_us long
            "collective": {**coll, "collective_us_min_max_over_ranks": [round(coll_min, 1), round(coll_max, 1)],
                           "estimate_us": {"direct_one_message_per_xgmi_link": 27, "ring_bound_by_one_link": 190, "source": "SURVEY.md section 8e (153 GB/s per link)"}},
            "rank_spread": {"seconds_fastest_rank": round(own_min, 4), "seconds_slowest_rank": round(own_max, 4), "generate_seconds_min_max": [round(cmp_min, 4), round(cmp_max, 4)],
                            "note": "per rank: generate() + the gather, before the closing barrier; `seconds` is the max over ranks of the barrier-closed time"}}


def _wall(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def measured_ceilings(device):
    """SURVEY 8d / BASELINE.md section 4: the two ceilings every roofline fraction of this line is priced against, MEASURED on this box beside the
    datasheet values -- a float4 device-to-device copy (bytes read + bytes written over the time) and a large square bf16 GEMM on hipBLASLt."""
    n = 1 << 30  # 1 GiB source, 1 GiB destination: far past the 256 MB Infinity Cache
    src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    ms_copy = min(event_time_ms(lambda: dst.copy_(src), 10, 3) for _ in range(2))
    # a READ-ONLY stream beside it (what a weight-streaming kernel does): this package's own dl_gemv over a [2^18, 4096] bf16 matrix (1 GiB) -- the copy's
    # read + write mix is not the ceiling of a pure read stream (torch.sum over the same bytes: 0.9 TB/s, a slow library reduction, not reported)
    read = {}
    try:
        from dynamic_llava_amd import hip_ops as ops

        wbig = src.view(torch.bfloat16).view(-1, 4096)
        xv = torch.randn(1, 4096, device=device, dtype=torch.bfloat16)
        yv = torch.empty(1, wbig.shape[0], device=device, dtype=torch.bfloat16)
        ms_gemv = min(event_time_ms(lambda: ops.gemv(wbig, yv, x=xv), 10, 3) for _ in range(2))
        read = {"hbm_read_dl_gemv_GBps": round(n / ms_gemv / 1e6, 1), "hbm_read_frac_of_spec": round(n / ms_gemv / 1e6 / HBM_PEAK_GBS, 3),
                "hbm_read": "1 GiB read once by ONE launch: dl_gemv [262144, 4096] bf16 at batch 1 (what a weight stream reaches when the launch is long enough "
                            "for its ramp not to matter; the decode step's launches stream 33-262 MB each)"}
    except Exception as e:  # noqa: BLE001
        read = {"hbm_read_error": repr(e)}
    N = 8192
    a = torch.randn(N, N, device=device, dtype=torch.bfloat16)
    b = torch.randn(N, N, device=device, dtype=torch.bfloat16)
    c = torch.empty(N, N, device=device, dtype=torch.bfloat16)
    ms_gemm = min(event_time_ms(lambda: torch.mm(a, b, out=c), 10, 3) for _ in range(2))
    return {"hbm_copy_GBps": round(2 * n / ms_copy / 1e6, 1), "hbm_copy": "torch copy_ of 1 GiB (uint8, 16-byte accesses): (read + written bytes) / time",
            "hbm_spec_GBps": HBM_PEAK_GBS, "hbm_copy_frac_of_spec": round(2 * n / ms_copy / 1e6 / HBM_PEAK_GBS, 3), **read,
            "bf16_gemm_TFLOPs": round(2 * N**3 / ms_gemm / 1e9, 1), "bf16_gemm": f"torch.mm {N}x{N}x{N} bf16 (hipBLASLt)", "bf16_dense_spec_TFLOPs": 2500.0,
            "bf16_gemm_frac_of_spec": round(2 * N**3 / ms_gemm / 1e9 / 2500.0, 3)}


def _weight_stream_bytes(model):
    return sum(p.numel() * p.element_size() for n, p in model.named_parameters() if ".layers." in n or n.startswith("lm_head"))


def prefill_flops(cfg, n_tokens_per_row, clip_cfg=None):
    """Dense-arithmetic FLOPs of ONE prefill as this path executes it (SURVEY 8d accounting): decoder GEMMs 2 (4 H^2 + 3 H I) per token-layer over
    2 N + (L - 2) N' token-layers per row (N' = N - 461 after the compaction at layer `sparse_layer`), causal attention 2 T^2 H per layer, lm_head for the
    LAST token of every row only (the product computes one row of logits per request, DML:2709 computes all N'), and per image the CLIP ViT-L/14-336 tower
    (23 of 24 layers: select_layer = -2), the mlp2x_gelu projector and the vision predictor."""
    H, I, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    SL = cfg.sparse_config["sparse_layer"]
    drop = N_IMG - int(N_IMG * cfg.sparse_config["vision_keep_rate"])
    per_tok_layer = 2 * (4 * H * H + 3 * H * I)
    gemm = attn = 0
    for n in n_tokens_per_row:
        n2 = n - drop
        gemm += per_tok_layer * (min(SL, L) * n + max(L - SL, 0) * n2) + 2 * H * V
        attn += 2 * H * (min(SL, L) * n * n + max(L - SL, 0) * n2 * n2)
    c = clip_cfg or cfg.clip
    C, CI, CL, T = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"] - 1, (c["image_size"] // c["patch_size"]) ** 2 + 1
    clip = CL * (2 * T * (4 * C * C + 2 * C * CI) + 4 * T * T * C) + 2 * (T - 1) * C * 3 * c["patch_size"] ** 2
    proj = 2 * (T - 1) * (C * H + H * H)
    sc = cfg.sparse_config
    dm, ff, nl = sc["d_model"], sc["dim_feedforward"], sc["num_layers"]
    pred = 2 * N_IMG * (H * dm + nl * (4 * dm * dm + 2 * dm * ff) + dm * dm // 2) + nl * 4 * N_IMG * N_IMG * dm
    n_img = len(n_tokens_per_row)
    return {"decoder_gemm": gemm, "decoder_attention": attn, "clip_tower": n_img * clip, "projector": n_img * proj, "vision_predictor": n_img * pred,
            "total": gemm + attn + n_img * (clip + proj + pred)}


def configs2_leg(model, cfg, device, dtype, new_tokens=128):
    """BASELINE configs[2], untimed by the driver but IN the driver's line: the bench's own 7B model, 32 ragged requests in one packed batch (question
    lengths ~U[8,64], seed 1), 128 greedy tokens each.  whole_step: (all streamed weights + the K/V rows the batch's attention reads at the final
    lengths) per decode step over the measured time per step, against the HBM spec."""
    g = torch.Generator().manual_seed(1)
    B = 32
    n_q = torch.randint(8, 65, (B,), generator=g).tolist()
    W = 35 + 1 + max(n_q)
    ids = torch.zeros(B, W, dtype=torch.long)
    am = torch.zeros(B, W, dtype=torch.long)
    for b in range(B):
        row = torch.cat([torch.tensor([1]), torch.randint(3, cfg.vocab_size, (34,), generator=g), torch.tensor([-200]), torch.randint(3, cfg.vocab_size, (n_q[b],), generator=g)])
        ids[b, : row.numel()] = row
        am[b, : row.numel()] = 1
    images = torch.randn(B, 3, 336, 336, generator=g).to(dtype).to(device)
    ids, am = ids.to(device), am.to(device)
    n_prompt = sum(35 + N_IMG + q for q in n_q)
    t_full = _wall(lambda: model.generate(ids, attention_mask=am, images=images, max_new_tokens=new_tokens, eos_token_id=None))
    lens = model.last_cache.lens.cpu()
    t_pre = _wall(lambda: model.generate(ids, attention_mask=am, images=images, max_new_tokens=1, eos_token_id=None))
    model.check_device_errors()
    dec_ms = (t_full - t_pre) / (new_tokens - 1) * 1e3
    SL, L, H = cfg.sparse_config["sparse_layer"], cfg.num_hidden_layers, cfg.hidden_size
    kv = sum(2 * int(lens[0 if i < SL else 1][b]) * H * 2 for i in range(L) for b in range(B))
    step_bytes = _weight_stream_bytes(model) + kv
    fl = prefill_flops(cfg, [35 + N_IMG + q for q in n_q])
    pf_tflops = fl["total"] / t_pre / 1e12
    prefill_roofline = {"bound": "mfma", "achieved": round(pf_tflops, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(pf_tflops / 2500.0, 4),
                        "flops": {k: int(v) for k, v in fl.items()}, "prefill_ms": round(t_pre * 1e3, 2),
                        "note": "the MFMA-bound leg of the path (VERDICT r5 item 6): dense FLOPs of the 32 ragged prompts' prefill -- decoder GEMMs over 2 N + 30 N' token-layers, "
                                "causal attention, one lm_head row per request, CLIP tower + projector + vision predictor per image -- over prefill_ms (CLIP -> first token, "
                                "host loop included), against the dense bf16 MFMA spec; frac_of_measured_gemm (filled in by main) prices it against this box's own 8192^3 "
                                "hipBLASLt rate; per-GEMM MFMA-busy: profiles/r06_configs2_prefill_mfma_util.txt"}
    return {"prefill_roofline": prefill_roofline,
            "workload": f"BASELINE configs[2]: LLaVA-1.5-7B bf16, B={B} images in one packed ragged batch (question lengths ~U[8,64]), {new_tokens} greedy tokens per row, 1 GPU",
            "tokens_per_s": round((n_prompt + B * new_tokens) / t_full, 1), "step_ms": round(t_full * 1e3, 2), "prefill_ms": round(t_pre * 1e3, 2),
            "prefill_tokens_per_s": round(n_prompt / t_pre, 1), "decode_ms_per_step": round(dec_ms, 3), "decode_tokens_per_s": round(B * 1e3 / dec_ms, 1),
            "kv_len_full_max": int(lens[0].max()), "kv_len_sparse_min_max": [int(lens[1].min()), int(lens[1].max())],
            "whole_step": {"bytes_per_step": int(step_bytes), "achieved": round(step_bytes / dec_ms / 1e6, 1), "frac": round(step_bytes / dec_ms / 1e6 / HBM_PEAK_GBS, 4), "unit": "GB/s",
                           "note": "all decoder + lm_head weights + the K/V rows the 32 rows' attention reads (final lengths) per decode step / decode_ms_per_step"}}


def configs4_leg(device, dtype, predictor_gain):
    """BASELINE configs[4]: LLaVA-1.5-13B bf16 (40 layers, random init), B=1, prompt 35 + 576 + 29 = 640 tokens -> 179 after layer 2, decoded to a
    total length of 2048 (1408 steps) with output-text KV eviction, the predictor calibrated like the headline's (about half of the tokens evicted)."""
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40)
    model = build_random_model(cfg, dtype=dtype, device=device, seed=0, predictor_gain=predictor_gain)
    g = torch.Generator().manual_seed(2)
    ids = torch.cat([torch.tensor([1]), torch.randint(3, cfg.vocab_size, (34,), generator=g), torch.tensor([-200]), torch.randint(3, cfg.vocab_size, (29,), generator=g)])[None].to(device)
    images = torch.randn(1, 3, 336, 336, generator=g).to(dtype).to(device)
    T_new = 2048 - 640
    calib = calibrate_text_predictor(model, ids, images, 64)
    t_full = _wall(lambda: model.generate(ids, images=images, max_new_tokens=T_new, eos_token_id=None), 1)
    lens = model.last_cache.lens.cpu().tolist()
    t_pre = _wall(lambda: model.generate(ids, images=images, max_new_tokens=1, eos_token_id=None), 3)
    model.check_device_errors()
    dec_ms = (t_full - t_pre) / (T_new - 1) * 1e3
    SL, L, H = cfg.sparse_config["sparse_layer"], cfg.num_hidden_layers, cfg.hidden_size
    # K/V read per token, averaged over the run: layers < SL grow 640 -> 2047, the evicting layers 179 -> final
    kv_avg = sum(2 * ((640 + lens[0][0]) / 2 if i < SL else (179 + lens[1][0]) / 2) * H * 2 for i in range(L))
    step_bytes = _weight_stream_bytes(model) + kv_avg
    res = {"workload": "BASELINE configs[4]: LLaVA-1.5-13B bf16, B=1, prompt 640 (-> 179 after layer 2) + 1408 greedy tokens = 2048, output-text KV eviction on, 1 GPU",
           "tokens_per_s": round((640 + T_new) / t_full, 1), "step_ms": round(t_full * 1e3, 1), "prefill_ms": round(t_pre * 1e3, 2), "decode_ms_per_token": round(dec_ms, 4),
           "decode_tokens_per_s": round(1e3 / dec_ms, 1), "kv_len_layers_0_1": lens[0][0], "kv_len_layers_ge2": lens[1][0], "kept_of_generated": lens[1][0] - 179,
           "text_predictor_calibrated_keep_fraction": calib,
           "whole_step": {"bytes_per_token": int(step_bytes), "achieved": round(step_bytes / dec_ms / 1e6, 1), "frac": round(step_bytes / dec_ms / 1e6 / HBM_PEAK_GBS, 4), "unit": "GB/s",
                          "note": "all decoder + lm_head weights + the K/V rows one step's attention reads (run average) / decode_ms_per_token"}}
    del model
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    partial = {"n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "rank": int(os.environ.get("RANK", "0"))}
    try:
        _main(args, partial)
    except BenchAbort as e:
        print(abort_line(str(e), e.detail, partial), flush=True)
        sys.exit(2)


def _main(args, partial):
    from dynamic_llava_amd import dist as dd
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (the reference forks one process per GPU from a shell loop,
        # run/dynamic_eval/eval_for_vqav2.sh:11-21); rank 0 of the child job prints the JSON line on the inherited stdout
        try:
            rc = dd.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        except dd.LaunchError as e:
            raise BenchAbort(str(e), visible_gpus=(torch.cuda.device_count() if torch.cuda.is_available() else 0), gpus=args.gpus)
        if rc != 0:  # the child job has printed its own line if it got far enough to know why; this one says that the job as a whole failed
            raise BenchAbort(f"the {args.gpus}-rank child job exited with code {rc}", child_exit_code=rc)
        return
    rank, world, local = dd.init_distributed()
    partial.update(rank=rank, world_size=world, local_rank=local)
    if world != args.gpus:
        if rank == 0:
            raise BenchAbort(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)", world_size=world, gpus=args.gpus)
        sys.exit(2)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_view = dd.describe(device)  # backend / world size / device per rank as the process group reports them (one all_gather_object, untimed)
    partial.update(dist=dist_view)
    dtype = torch.bfloat16
    cfg = DynamicLlavaConfig(num_hidden_layers=args.layers)  # LLaVA-1.5-7B defaults, sparse_layer=2, keep 0.2
    model = build_random_model(cfg, dtype=dtype, device=device, seed=0, predictor_gain=args.predictor_gain)
    model.use_hip_graph = not args.no_graph
    model.tp_side_stream = os.environ.get("DL_TP_SIDE", "0") == "1"
    prompt, images = make_inputs(cfg, device, dtype)
    n_prompt = N_SYS + N_IMG + N_Q
    T_new = args.new_tokens
    calib = None if args.no_calibrate else calibrate_text_predictor(model, prompt, images, T_new)
    gathered = {}

    def step():
        out = model.generate(prompt, images=images, max_new_tokens=T_new, do_sample=False, num_beams=1, use_cache=True, eos_token_id=None)
        if world > 1:  # DP result = concatenation over ranks (logits of the prefill's last token + generated ids): ONE collective
            gathered["logits"], gathered["ids"] = dd.gather_results(model.last_prefill_logits, out, max_rows=1, max_new_tokens=T_new)
        return out

    for _ in range(args.warmup):
        step()
    dd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others
    dd.barrier()
    elapsed = dd.max_over_ranks(time.perf_counter() - t0, device)
    el_min, el_max = dd.min_max_over_ranks(own_elapsed, device)  # per-rank spread (one more all-reduce, outside the timed region): a slow rank explains a 1 -> N curve
    rank_spread = {"ms_per_step_fastest_rank": round(el_min * 1e3 / args.steps, 3), "ms_per_step_slowest_rank": round(el_max * 1e3 / args.steps, 3),
                   "note": "each rank's own time for the K steps (its collectives included), before the closing barrier; `ms_per_step` is the max over ranks of the barrier-closed time"}
    partial.update(ms_per_step=round(elapsed * 1e3 / args.steps, 3), value_if_valid=round(world * (n_prompt + T_new) * args.steps / elapsed, 2))
    model.check_device_errors()  # a launch with in-kernel hand-offs that gave up would have poisoned its output: never report such a run
    end_lens = model.last_cache.lens.cpu().tolist()  # KV lengths at the end of a full step (before the pooled slab is reset)
    dp_consistent = None
    c3 = None
    if world > 1:  # identical requests on every rank: the gathered token ids must be identical -- FATAL otherwise (a DP run whose ranks disagree
        # measured nothing); the gathered prefill logits must agree to the last bit too, or -- if a device's library GEMM picked another
        # kernel -- within the bf16 noise class (reported as dp_rows_identical: false with the difference; beyond it: fatal)
        ids_same = bool(gathered["ids"].shape[0] == world and all(torch.equal(gathered["ids"][0], gathered["ids"][r]) for r in range(world)))
        lg_diff = max(float((gathered["logits"][0].float() - gathered["logits"][r].float()).abs().max()) for r in range(world))
        lg_mag = float(gathered["logits"][0].float().abs().max())
        dp_consistent = bool(ids_same and lg_diff == 0.0)
        if not ids_same or lg_diff > 2e-2 * max(lg_mag, 1.0):
            per_rank = [{"rank": r, "ids_equal_rank0": bool(torch.equal(gathered["ids"][0], gathered["ids"][r])), "first_new_tokens": gathered["ids"][r][:8].tolist(),
                         "max_abs_logit_diff_vs_rank0": float((gathered["logits"][0].float() - gathered["logits"][r].float()).abs().max())} for r in range(gathered["ids"].shape[0])]
            if rank == 0:  # every rank holds the same gathered tensors: one line, from rank 0
                raise BenchAbort("data-parallel ranks produced different results for identical requests", ids_equal=ids_same, max_logit_diff=lg_diff, logit_magnitude=lg_mag, per_rank=per_rank)
            sys.exit(2)
        c3 = configs3_leg(model, cfg, dd, rank, world, device, dtype)

    if rank != 0:
        return
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * (n_prompt + T_new) * args.steps / elapsed
    # ---- phase split (outside the timed region) ----
    out = model.generate(prompt, images=images, max_new_tokens=T_new, do_sample=False, num_beams=1, use_cache=True, eos_token_id=None)  # fresh run: end_lens of a step
    end_lens = model.last_cache.lens.cpu().tolist()
    pre_ms = min(event_time_ms(lambda: model.generate(prompt, images=images, max_new_tokens=1, eos_token_id=None), 3, 1) for _ in range(2))
    dec_ms = (ms_per_step - pre_ms) / max(T_new - 1, 1)
    clip_ms = event_time_ms(lambda: model.encode_images(images), 5, 2)
    try:  # the same work as it runs in the product path: inside a captured graph (the eager number above is mostly ~190 host launches)
        clip_graph_ms = graph_time_ms(lambda: model.encode_images(images), reps=4, replays=6)
    except Exception as e:  # noqa: BLE001 -- a reported phase, never the metric
        clip_graph_ms = None
        print(f"clip graph timing skipped: {e!r}", file=sys.stderr)
    t_full, t_sparse = end_lens[0][0], end_lens[1][0]
    nH, d = cfg.num_attention_heads, cfg.head_dim
    roof_attn = decode_attn_roofline(model, f"bench workload, layers>=2 at the last decode step: B=1, T={t_sparse + 1} (170 prompt + kept decode tokens + the new one)", 1, [t_sparse + 1], nH, d)
    roof_main, gemv_shapes = gemv_roofline(model)
    model.generate(prompt, images=images, max_new_tokens=T_new, do_sample=False, num_beams=1, use_cache=True, eos_token_id=None)  # leave the slab at the final lengths
    roof_fused = fused_launch_roofline(model)
    extra = ([roof_fused] if roof_fused else []) + [
        roof_attn,
        decode_attn_roofline(model, f"bench workload, layers 0-1 at the last decode step: B=1, T={t_full + 1}", 1, [t_full + 1], nH, d),
        decode_attn_roofline(model, "configs[2]-like: B=32 ragged, T spread over [299, 887] (round 6: a real spread -- 200 + 229 i mod 700; up to round 5 the ladder 701 i mod 700 made this entry T = 200..231)", 32, [200 + (i * 229) % 700 for i in range(32)], nH, d),
        decode_attn_roofline(model, "configs[4]-like: 13B heads (40x128), B=1, T=2048", 1, [2048], 40, d),
        decode_attn_roofline(model, "B=32, T=2048", 32, [2048] * 32, nH, d),
    ] + gemv_shapes + linear_packed_roofline(model) + linear_tiles_roofline(model) + other_kernel_rooflines(model, n_prompt)
    traffic, traffic_src = None, None
    try:  # HBM bytes per launch from the committed PMC passes (tools/pmc_probe.py under rocprofv3 --pmc, see profiles/)
        import glob
        pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]  # the latest round's committed passes
        pmc_rel = os.path.relpath(pmc_file, ROOT)
        with open(pmc_file) as f:
            pmc = {r["case"]: r for r in json.load(f)}
        # a committed profile only speaks for the kernels it was taken on: the file records the digests of the sources its launches were
        # built from; a tree whose gemv / attention sources have changed since gets NO traffic figure instead of a stale one
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pmc_report import source_digests

        meta = pmc.get("_meta")
        if meta is None or meta.get("source_digests") != source_digests():
            stale = [k for k, v in source_digests().items() if (meta or {}).get("source_digests", {}).get(k) != v]
            traffic_src = f"REFUSED: {pmc_rel} was taken on other kernel sources (differs: {stale}); re-run tools/refresh_profiles.sh and commit profiles/rNN_pmc_traffic.json"
            raise LookupError(traffic_src)
        rs = [pmc[k] for k in ("gemv qkv", "gemv o", "gemv gate_up", "gemv down")]
        ratio = sum(r["fetch_bytes_corrected"] + r["write_bytes"] for r in rs) / sum(r["algorithmic_bytes"] for r in rs)
        traffic = int(ratio * roof_main["bytes"])
        traffic_src = (f"{pmc_rel}: rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes over the four "
                       f"per-layer dl_gemv shapes: measured traffic / algorithmic bytes = {ratio:.4f}, applied to this run's average launch")
        roof_attn["traffic_over_algorithmic_pmc"] = pmc["decode_attn B=1 T=226"]["traffic_over_algorithmic"]
        if roof_fused and "gemv_qkv_attn T=226" in pmc:
            roof_fused["traffic_over_algorithmic_pmc"] = pmc["gemv_qkv_attn T=226"]["traffic_over_algorithmic"]
    except LookupError:
        traffic = None  # traffic_src says why
    except Exception as e:  # noqa: BLE001 -- no committed profile at all
        traffic, traffic_src = None, f"no usable profiles/r*_pmc_traffic.json ({e!r})"
    # the WHOLE decode step of the product (every launch of the captured graph, idle gaps included): algorithmic bytes = all streamed weights +
    # the K/V rows the step's attention reads (2 T H E per layer) over the measured time per token
    H_ = cfg.hidden_size
    SL_ = cfg.sparse_config["sparse_layer"]
    kv_bytes = sum(2 * ((t_full if i < SL_ else t_sparse) + 1) * H_ * 2 for i in range(cfg.num_hidden_layers))
    step_bytes = roof_main["bytes_per_step"] + kv_bytes
    whole = {"bytes_per_token": int(step_bytes), "ms_per_token": round(dec_ms, 4), "achieved": round(step_bytes / dec_ms / 1e6, 1),
             "frac": round(step_bytes / dec_ms / 1e6 / HBM_PEAK_GBS, 4),
             "note": "all weights streamed by one decode step + the K/V rows its attention reads (at the final lengths), over decode_ms_per_token of the timed generate() "
                     "calls: includes every launch of the step's graph, launch boundaries and the host loop"}
    res = {
        "metric": METRIC,
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (random-init LLaVA-1.5-7B + CLIP ViT-L/14-336 weights, randn 336x336 image, random token ids)",
        "config": {"workload": "BASELINE configs[1]: LLaVA-1.5-7B bf16, B=1 per GPU, 1 image, prompt 35+576+20=631 tokens (170 after layer 2), "
                               f"vision_keep_rate=0.2, output-text KV eviction on, greedy {T_new} new tokens; step = CLIP+projector+prefill+decode",
                   "tokens_per_step_per_gpu": n_prompt + T_new, "parallelism": f"dp{world}", "dist": dist_view, "dp_rows_identical": dp_consistent, "dp_max_abs_logit_diff": (lg_diff if world > 1 else None), "rank_spread": rank_spread, "hip_graph_decode": model.use_hip_graph,
                   "predictor_gain": args.predictor_gain, "text_predictor_calibrated_keep_fraction": calib, "knobs": model.knobs(),
                   "parity_note": "ids / kept sets / KV lengths bit-exact vs the oracle; logits: 1e-3 asserted literally in fp32, bf16 held to the reference's own "
                                  "eager-bf16 noise class against an fp32 truth (DESIGN.md section 5)"},
        "phases": {"prefill_ms": round(pre_ms, 3), "clip_projector_ms": round(clip_ms, 3), "clip_projector_graph_ms": (None if clip_graph_ms is None else round(clip_graph_ms, 3)), "decode_ms_per_token": round(dec_ms, 4),
                   "prefill_tokens_per_s": round(n_prompt / pre_ms * 1e3, 1), "decode_tokens_per_s": round(1e3 / dec_ms, 1),
                   "kv_len_full": t_full, "kv_len_sparse": t_sparse,
                   "evicted/generated": f"{(N_SYS + 115 + N_Q + T_new - 1) - t_sparse}/{T_new - 1}",
                   "bimg_prompt": bimg_prompt_phase(model, images) if world == 1 else None,
                   "decode_weight_stream_GBps": round(sum(p.numel() * p.element_size() for n, p in model.named_parameters() if ".layers." in n or n.startswith("lm_head")) / dec_ms / 1e6, 1)},
        # dominant kernel of the step by time (~84 % of a decode step, rocprof: profiles/): the hand-written weight-streaming GEMV
        "roofline": {"bound": "hbm", "achieved": roof_main["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roof_main["frac"], "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": roof_main["kernel"], "launches_per_step": roof_main["launches_per_step"], "bytes_per_launch": roof_main["bytes"], "us_per_launch": roof_main["us"],
                     "bytes_per_step": roof_main["bytes_per_step"], "us_per_step": roof_main["us_per_step"], "whole_step": whole,
                     "note": "averaged over the 129 dl_gemv launches of one decode step replayed as one hipGraph between HIP events on the launch stream (includes inter-kernel gaps); the north_star's sparse-attention kernel is the first entry of roofline_kernels; in the product step the q|k|v launches of the single-split layers also carry that layer's attention workgroups (dl_gemv_qkv_attn) and one gate|up launch the text predictor (dl_gemv_gu_tp): same streaming body, same bytes"},
        "roofline_kernels": extra,
    }
    if c3 is not None:
        res["configs3"] = c3
    if args.layers != 32:
        res["INVALID"] = f"debug run with {args.layers} layers"
    if world == 1 and not args.no_ref_gpu:
        try:
            res["ref_gpu_path"] = ref_gpu_path(model, prompt, images, T_new)
            res["ref_gpu_path"]["speedup_total"] = round(value / res["ref_gpu_path"]["value"], 2)
            res["ref_gpu_path"]["speedup_prefill"] = round(res["phases"]["prefill_tokens_per_s"] / res["ref_gpu_path"]["prefill_tokens_per_s"], 2)
        except Exception as e:  # the baseline leg must never take the measurement down
            res["ref_gpu_path"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(model, prompt, images, T_new)
        except Exception as e:
            res["cpu_baseline"] = {"error": repr(e)}
    if world == 1 and not args.no_extra_legs and args.layers == 32:
        # the other single-GPU BASELINE configs and the box's measured ceilings, OUTSIDE the timed region of the headline: reported
        # in the driver-visible line, never part of `value`
        for key, leg in (("measured_ceilings", lambda: measured_ceilings(device)), ("configs2", lambda: configs2_leg(model, cfg, device, dtype)),
                         ("configs4", lambda: configs4_leg(device, dtype, args.predictor_gain))):
            try:
                res[key] = leg()
            except Exception as e:  # a reported leg must never take the measurement down
                res[key] = {"error": repr(e)}
        if isinstance(res.get("measured_ceilings"), dict) and "bf16_gemm_TFLOPs" in res["measured_ceilings"] and isinstance(res.get("configs2"), dict) and "prefill_roofline" in res["configs2"]:
            res["configs2"]["prefill_roofline"]["frac_of_measured_gemm"] = round(res["configs2"]["prefill_roofline"]["achieved"] / res["measured_ceilings"]["bf16_gemm_TFLOPs"], 4)
        if isinstance(res.get("measured_ceilings"), dict) and "hbm_copy_GBps" in res["measured_ceilings"]:
            m = res["measured_ceilings"]["hbm_copy_GBps"]
            res["roofline"]["frac_of_measured_copy"] = round(res["roofline"]["achieved"] / m, 4)
            res["roofline"]["whole_step"]["frac_of_measured_copy"] = round(whole["achieved"] / m, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
