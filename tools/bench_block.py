#!/usr/bin/env python
"""dl_decode_block (csrc/decode_block.hip, the LDS-DMA engine) against the chain of dl_gemv launches it replaces: bit-exactness and time.
    python tools/bench_block.py [--h 4096 --i 11008 --n-qkv 12288] [--layers 6]
Single phases first (each GEMV shape of the decode layer as a one-phase block vs dl_gemv), then the whole block o -> gate|up -> down ->
next q|k|v vs the four launches; weights are rotated over `--layers` distinct copies so that every launch streams from HBM."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dynamic_llava_amd import hip_ops as ops  # noqa: E402


def graph_us(fn, n_inner, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * n_inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=4096)
    ap.add_argument("--i", type=int, default=11008)
    ap.add_argument("--n-qkv", type=int, default=12288)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--mode", type=int, default=0, help="dl_decode_block debug_mode: 1 = consumers skip the arithmetic, 2 = loader wave alone")
    args = ap.parse_args()
    ops.require_gpu()
    dev, dt = torch.device("cuda"), getattr(torch, args.dtype)
    H, I, NQ, L = args.h, args.i, args.n_qkv, args.layers
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *shape, s=0.02: (torch.randn(*shape, device=dev, generator=g) * s).to(dt)
    Wo = [rnd(H, H) for _ in range(L)]
    Wgu = [rnd(2 * I, H) for _ in range(L)]
    Wd = [rnd(H, I) for _ in range(L)]
    Wq = [rnd(NQ, H) for _ in range(L)]
    nw1 = [(1 + rnd(H, s=0.1)) for _ in range(L)]
    nw2 = [(1 + rnd(H, s=0.1)) for _ in range(L)]
    attn = rnd(1, H, s=1.0)
    h0 = rnd(1, H, s=1.0)
    eps = 1e-5
    pos = torch.tensor([77], dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    sync = ops.decode_block_sync(max(H, I), dev)
    A, P = ops.BLK_ADDNORM, ops.BLK_SILU_PAIR

    # ---- reference: the launch path ----
    def launch_layer(l, attn_x, h_in, h_mid, h_out, o, act, dn, qkv):
        ops.gemv(Wo[l], o, x=attn_x)
        ops.gemv(Wgu[l], act, mode=ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR, h_in=h_in, h_out=h_mid, delta=o, norm_w=nw1[l], eps=eps)
        ops.gemv(Wd[l], dn, x=act)
        ops.gemv(Wq[l], qkv, mode=ops.GEMV_ADDNORM, h_in=h_mid, h_out=h_out, delta=dn, norm_w=nw2[l], eps=eps)

    mk = lambda n: torch.zeros(1, n, dtype=dt, device=dev)
    o_r, act_r, dn_r, qkv_r, hm_r, ho_r = mk(H), mk(I), mk(H), mk(NQ), mk(H), mk(H)
    launch_layer(0, attn, h0, hm_r, ho_r, o_r, act_r, dn_r, qkv_r)
    torch.cuda.synchronize()

    # ---- single phases ----
    def one(spec, call_tag):
        ops.decode_block(ops.block_phases([spec]), sync, pos, call_tag, eps, dt, err=err, debug_mode=args.mode)

    res = {}
    y = mk(H)
    one(dict(W=Wo[0], x_in=attn, out=y), 1)
    res["o_proj"] = bool(torch.equal(y, o_r))
    y2 = mk(I)
    hm = mk(H)
    one(dict(W=Wgu[0], x_in=o_r, h_in=h0, h_out=hm, norm_w=nw1[0], out=y2, flags=A | P), 2)
    res["gate|up"] = bool(torch.equal(y2, act_r)) and bool(torch.equal(hm, hm_r))
    y3 = mk(H)
    one(dict(W=Wd[0], x_in=act_r, out=y3), 3)
    res["down"] = bool(torch.equal(y3, dn_r))
    y4 = mk(NQ)
    ho = mk(H)
    one(dict(W=Wq[0], x_in=dn_r, h_in=hm_r, h_out=ho, norm_w=nw2[0], out=y4, flags=A), 4)
    res["qkv"] = bool(torch.equal(y4, qkv_r)) and bool(torch.equal(ho, ho_r))
    torch.cuda.synchronize()
    print("single-phase blocks bit-identical to dl_gemv:", res, "err flag", int(err.item()))
    if not res["o_proj"]:
        d = (y.float() - o_r.float()).abs()
        print("  o_proj max diff", float(d.max()), "n_diff", int((y != o_r).sum()), "first", torch.nonzero((y != o_r)[0])[:8].flatten().tolist())

    # ---- the whole block ----
    def block_layer(l, attn_x, h_in, h_out, qkv, tag):
        ph = ops.block_phases([
            dict(W=Wo[l], x_in=attn_x),
            dict(W=Wgu[l], h_in=h_in, norm_w=nw1[l], flags=A | P),
            dict(W=Wd[l]),
            dict(W=Wq[l], norm_w=nw2[l], h_out=h_out, out=qkv, flags=A),
        ])
        ops.decode_block(ph, sync, pos, tag, eps, dt, err=err, debug_mode=args.mode)

    qkv_b, ho_b = mk(NQ), mk(H)
    block_layer(0, attn, h0, ho_b, qkv_b, 9)
    torch.cuda.synchronize()
    print("4-phase block: qkv identical", bool(torch.equal(qkv_b, qkv_r)), "h identical", bool(torch.equal(ho_b, ho_r)), "err flag", int(err.item()))
    # repeated calls with other tags / positions: no stale granule may be consumed
    okr = True
    for rep in range(20):
        pos.fill_(100 + rep)
        qkv_b.zero_(); ho_b.zero_()
        block_layer(0, attn, h0, ho_b, qkv_b, rep & 0xff)
        okr &= bool(torch.equal(qkv_b, qkv_r)) and bool(torch.equal(ho_b, ho_r))
    print("20 repeats identical:", okr, "err flag", int(err.item()))

    # ---- timing: L layers back to back (distinct weights), as the decode step chains them ----
    bufs = [mk(H), mk(I), mk(H), mk(NQ), mk(H), mk(H)]

    def chain_launch():
        for l in range(L):
            launch_layer(l, attn, h0, bufs[4], bufs[5], bufs[0], bufs[1], bufs[2], bufs[3])

    def chain_block():
        for l in range(L):
            block_layer(l, attn, h0, bufs[5], bufs[3], l)

    t_l = graph_us(chain_launch, L)
    t_b = graph_us(chain_block, L)
    mb = (H * H + 2 * I * H + H * I + NQ * H) * 2 / 1e6
    print(f"per layer ({mb:.1f} MB of weights): 4 dl_gemv launches {t_l:.2f} us ({mb / t_l:.2f} TB/s)   dl_decode_block {t_b:.2f} us ({mb / t_b:.2f} TB/s)   ratio {t_b / t_l:.3f}")
    for name, spec_fn, mbs in (("o_proj", lambda l: dict(W=Wo[l], x_in=attn, out=bufs[0]), H * H * 2 / 1e6),
                               ("gate|up", lambda l: dict(W=Wgu[l], x_in=o_r, h_in=h0, h_out=bufs[4], norm_w=nw1[l], out=bufs[1], flags=A | P), 2 * I * H * 2 / 1e6),
                               ("down", lambda l: dict(W=Wd[l], x_in=act_r, out=bufs[2]), H * I * 2 / 1e6),
                               ("qkv", lambda l: dict(W=Wq[l], x_in=dn_r, h_in=hm_r, h_out=bufs[5], norm_w=nw2[l], out=bufs[3], flags=A), NQ * H * 2 / 1e6)):
        def chain_one():
            for l in range(L):
                one(spec_fn(l), l)
        t = graph_us(chain_one, L)
        print(f"  one-phase block {name:8s}: {t:6.2f} us  {mbs / t:6.2f} TB/s")
    if args.stamps:
        n_ph = 4
        st = torch.zeros(256 * n_ph * 8, dtype=torch.int64, device=dev)
        ph = ops.block_phases([dict(W=Wo[1], x_in=attn), dict(W=Wgu[1], h_in=h0, norm_w=nw1[1], flags=A | P), dict(W=Wd[1]), dict(W=Wq[1], norm_w=nw2[1], h_out=bufs[5], out=bufs[3], flags=A)])
        ops.decode_block(ph, sync, pos, 200, eps, dt, err=err, stamps=st, debug_mode=args.mode)
        torch.cuda.synchronize()
        s = st.view(256, n_ph, 8).cpu().double()
        t0 = s[:, 0, 0].min()
        names = ["phase top", "sample ok", "input in LDS", "x ready", "units done"]
        for p_ in range(n_ph):
            print(f"  phase {p_}: loader issued last piece {((s[:, p_, 5] - t0) / 100).median():7.2f}/{((s[:, p_, 5] - t0) / 100).max():7.2f}, blocked {s[:, p_, 6].median():.0f} times for {(s[:, p_, 7] / 100).median():.2f} us (median)")
            print(f"  phase {p_}: " + "  ".join(f"{names[k]} {((s[:, p_, k] - t0) / 100).median():7.2f}/{((s[:, p_, k] - t0) / 100).max():7.2f}" for k in range(5) if s[:, p_, k].max() > 0) + "  (us median/max over CUs, 100 MHz clock)")


if __name__ == "__main__":
    main()
