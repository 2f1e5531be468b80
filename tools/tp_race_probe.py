#!/usr/bin/env python
"""Isolates the configs[3] divergence found by tools/diverge_probe.py: `dl_text_predictor_decide` returns different logits for the SAME
input while another process shares the GPU.  This probe runs only that op (B=32, H=4096, d_model=512, bf16) on a fixed input, alone and
under several kinds of load from a second process, and reports for every mismatching call which stage's output differs first (h1 =
stage 1, a1 = stage 2a, logits = stage 2b), how many elements, where, and expected vs observed values (stale data vs garbage).

    python tools/tp_race_probe.py [--iters 3000] [--stress gemm,stream,tp]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def stress(kind, ready, stop):
    dev = torch.device("cuda", 0)
    if kind == "gemm":
        a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        fn = lambda: a @ b
    elif kind == "stream":
        a = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
        b = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
        fn = lambda: b.copy_(a)
    elif kind == "small":  # many short kernels: keeps the dispatcher and the caches busy
        a = torch.randn(64, 4096, device=dev, dtype=torch.bfloat16)
        fn = lambda: torch.nn.functional.gelu(a)
    else:
        raise SystemExit(kind)
    fn()
    torch.cuda.synchronize()
    open(ready, "w").write("1")
    while not os.path.exists(stop):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--stress", default="gemm,stream,small")
    ap.add_argument("--role", default="main")
    ap.add_argument("--kind", default="gemm")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--ready", default="/tmp/tp_ready")
    ap.add_argument("--stop", default="/tmp/tp_stop")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tp_race_probe.json"))
    args = ap.parse_args()
    torch.cuda.set_device(0)
    if args.role == "stress" and args.kind != "tpself":
        return stress(args.kind, args.ready, args.stop)
    from dynamic_llava_amd import hip_ops as ops
    from dynamic_llava_amd.model import TextPredictor

    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.manual_seed(0)
    B, H, D = args.B, 4096, 512
    tp = TextPredictor(input_dim=H, d_model=D).to(dev, dt)
    xs = [torch.randn(B, H, device=dev, dtype=dt) * s for s in (1.0, 0.5, 2.0)]  # alternate inputs: a stale read then shows ANOTHER call's value
    ws = ops.text_predictor_workspace(B, D, dev)
    lg = torch.zeros(B, 2, device=dev, dtype=torch.float32)
    dec = torch.zeros(B, dtype=torch.int32, device=dev)

    addr = {"l7_bias": tp.output_mlp[7].bias.data_ptr(), "l1_w": tp.output_mlp[1].weight.data_ptr(), "x0": xs[0].data_ptr(), "ws": ws.data_ptr()}
    if args.role == "stress":  # kind == "tpself": same addresses (identical allocation sequence), other bias values, same kernels
        tp.output_mlp[7].bias.data += 8.0
        tp.output_mlp[5].bias.data += 1.0
        tp._w = None
        json.dump({k: hex(v) for k, v in addr.items()}, open(args.ready + ".addr", "w"))
        tp.decide(xs[0], ws, lg, dec)
        torch.cuda.synchronize()
        open(args.ready, "w").write("1")
        while not os.path.exists(args.stop):
            for i in range(200):
                tp.decide(xs[i % 3], ws, lg, dec)
            torch.cuda.synchronize()
        return

    def call(i):
        ws.fill_(777.0)
        tp.decide(xs[i % 3], ws, lg, dec)
        return ws[: B * D].clone(), ws[B * D : B * D + B * D // 2].clone(), lg.clone()

    ref = [tuple(t.clone() for t in call(i)) for i in range(3)]
    torch.cuda.synchronize()

    def sweep(n):
        bad = []
        for i in range(n):
            h1, a1, l_ = call(i)
            r = ref[i % 3]
            e = [not torch.equal(h1, r[0]), not torch.equal(a1, r[1]), not torch.equal(l_, r[2])]
            if any(e) and len(bad) < 12:
                rec = {"iter": i, "h1_differs": e[0], "a1_differs": e[1], "logits_differ": e[2]}
                for name, got, exp, width in (("h1", h1, r[0], D), ("a1", a1, r[1], D // 2)):
                    ne = torch.nonzero(got != exp).flatten()
                    if ne.numel():
                        idx = ne[:8].tolist()
                        prev = ref[(i - 1) % 3][0 if name == "h1" else 1]
                        rec[name] = {"n_diff": int(ne.numel()), "n_sentinel": int((got == 777.0).sum()), "first_(row,col)": [(j // width, j % width) for j in idx], "got": [float(got[j]) for j in idx],
                                     "expected": [float(exp[j]) for j in idx], "previous_call_value": [float(prev[j]) for j in idx],
                                     "rows_touched": sorted(set((ne // width).tolist()))[:16], "cols_touched_mod8": sorted(set((ne % width % 8).tolist()))}
                bad.append(rec)
                if e[2]:
                    ne = torch.nonzero((l_ != r[2]).any(dim=1)).flatten()[:6].tolist()
                    rec["logits"] = {"rows": ne, "got": [[float(v) for v in l_[j]] for j in ne], "expected": [[float(v) for v in r[2][j]] for j in ne]}
            elif any(e):
                bad.append({"iter": i})
        return bad

    rep = {"B": B, "iters": args.iters, "addresses_main": {k: hex(v) for k, v in addr.items()}}
    solo = sweep(args.iters)
    rep["solo_mismatching_calls"] = len(solo)
    rep["solo_examples"] = solo[:3]
    for kind in [k for k in args.stress.split(",") if k]:
        for f in (args.ready, args.stop):
            if os.path.exists(f):
                os.remove(f)
        if kind.startswith("model"):  # the other rank's real workload: tools/diverge_probe.py's stress role (B=32 generate loop); modelN[:mode] = N layers
            nl, _, mode = kind[5:].partition(":")
            nl = nl or "32"
            os.environ["DP_STRESS_MODE"] = mode
            p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "diverge_probe.py"), "--role", "stress", "--stress", "model", "--layers", nl, "--new", "32",
                                  "--ready", args.ready, "--stop", args.stop])
        else:
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--role", "stress", "--kind", kind, "--ready", args.ready, "--stop", args.stop])
        t0 = time.time()
        while not os.path.exists(args.ready) and time.time() - t0 < 300 and p.poll() is None:
            time.sleep(0.2)
        try:
            bad = sweep(args.iters)
        finally:
            open(args.stop, "w").write("1")
            try:
                p.wait(timeout=60)
            except Exception:
                p.kill()
        if os.path.exists(args.ready + ".addr"):
            rep[f"addresses_{kind}"] = json.load(open(args.ready + ".addr"))
        rep[f"under_{kind}_mismatching_calls"] = len(bad)
        rep[f"under_{kind}_examples"] = [b for b in bad if len(b) > 1][:6]
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(rep, open(args.out, "w"), indent=1)
    json.dump(rep, open(args.out, "w"), indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
