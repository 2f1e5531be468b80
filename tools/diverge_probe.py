#!/usr/bin/env python
"""Root-cause probe for the configs[3] divergence: the B=32 ragged chunk of bench.py's configs3 leg is run
(i) alone, several times -- first call (graph capture) vs replays vs after ANOTHER chunk went through the pooled slab / graph caches --
and (ii) while a second process keeps the same GPU busy.  Every run records the generated ids, the prefill logits, a position-weighted
checksum of every valid K/V row after the prefill, and -- in the eager trace mode -- a checksum of every layer's output at every decode
step, so that the first differing (step, layer, tensor) is named.  Writes gpurun_out/diverge_probe.json.

    python tools/diverge_probe.py [--layers 32] [--new 32] [--stress model|gemm|none] [--reps 3]
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
import torch.nn.functional as F  # noqa: E402


def chunk(cfg, r, world, device, dtype, per=32):
    """bench.py::configs3_leg's request generator, verbatim semantics (seeds included)."""
    from dynamic_llava_amd import dist as dd

    g = torch.Generator().manual_seed(1)
    n_req = per * world
    n_q = torch.randint(8, 65, (n_req,), generator=g).tolist()
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
    return ids.to(device), am.to(device), imgs.to(device)


def csum(t):
    """Order-sensitive 64-bit checksum of a tensor's bits (device side, one small host copy)."""
    v = t.contiguous().view(-1)
    if v.element_size() == 2:
        v = v.view(torch.int16).to(torch.int64) & 0xFFFF
    elif v.element_size() == 4:
        v = v.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    else:
        v = v.to(torch.int64)
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
    return int((v * w).sum().item())


def kv_sums(cache):
    """Per layer: checksum over the VALID K and V rows of every request (rows beyond the length are garbage of earlier requests)."""
    lens = cache.lens.cpu().tolist()
    out = []
    for i in range(cache.n_layers):
        ln = lens[cache.group(i)]
        s = 0
        for b, n in enumerate(ln):
            s ^= csum(cache.k[i][b, :, :n]) ^ (csum(cache.v[i][b, :, :n]) << 1)
        out.append(s)
    return out


def run_once(model, ids, am, imgs, new, scores=False):
    kw = dict(attention_mask=am, images=imgs, max_new_tokens=new, eos_token_id=None)
    if scores:
        kw.update(return_dict_in_generate=True, output_scores=True)
    out = model.generate(ids, **kw)
    rec = {}
    if scores:
        rec["scores"] = [csum(s) for s in out["scores"]]
        out = out["sequences"]
    rec["ids"] = out.cpu()
    rec["prefill_logits"] = csum(model.last_prefill_logits.float())
    rec["lens"] = model.last_cache.lens.cpu().tolist()
    return rec


def prefill_kv(model, ids, am, imgs):
    model.generate(ids, attention_mask=am, images=imgs, max_new_tokens=1, eos_token_id=None)
    return kv_sums(model.last_cache)


def diff(a, b):
    d = {"ids_equal": bool(torch.equal(a["ids"], b["ids"])), "prefill_logits_equal": a["prefill_logits"] == b["prefill_logits"], "lens_equal": a["lens"] == b["lens"]}
    if not d["ids_equal"]:
        ne = a["ids"] != b["ids"]
        d["rows_differ"] = int(ne.any(dim=1).sum())
        d["first_step_differs"] = int(ne.any(dim=0).int().argmax())
        d["rows"] = torch.nonzero(ne.any(dim=1)).flatten().tolist()
    if "scores" in a and "scores" in b:
        bad = [i for i, (x, y) in enumerate(zip(a["scores"], b["scores"])) if x != y]
        d["first_score_step_differs"] = bad[0] if bad else None
    return d


class Tracer:
    """Eager decode with a checksum after every kernel group of every layer (monkeypatches the ops the B=32 step calls)."""

    def __init__(self, model):
        from dynamic_llava_amd import hip_ops as ops

        self.model, self.ops, self.log, self.on = model, ops, [], False
        self.tp_calls = []  # per dl_text_predictor_decide call: (input x, h1, a1, logits) clones
        self._orig = {}
        for name in ("attn_decode_rope", "add_rmsnorm", "silu_mul", "gemm_smallm", "add_rmsnorm_parts", "silu_mul_parts", "rmsnorm", "decode_advance", "text_predictor_decide"):
            self._orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name, self._orig[name]))
        self._lin = F.linear

        def lin(x, w, b=None):
            y = self._lin(x, w, b)
            if self.on:
                self.log.append(("linear", tuple(y.shape), csum(y)))
            return y

        import dynamic_llava_amd.model as M

        M.F.linear = lin
        self._mm = torch.matmul

    def _wrap(self, name, fn):
        def w(*a, **k):
            if self.on and name == "text_predictor_decide":
                x_before = a[0].clone()
            r = fn(*a, **k)
            if self.on and name == "text_predictor_decide":
                B_, D_ = a[0].shape[0], a[2]
                ws_ = a[3]
                self.tp_calls.append((x_before, a[0].clone(), ws_[: B_ * D_].clone().view(B_, D_), ws_[B_ * D_ : B_ * D_ + B_ * D_ // 2].clone().view(B_, D_ // 2), a[4].clone()))
            if self.on:
                outs = []
                if name == "attn_decode_rope":
                    outs = [a[7]]
                elif name in ("add_rmsnorm", "add_rmsnorm_parts"):
                    outs = [a[0], k.get("out", r if torch.is_tensor(r) else None)]
                elif name in ("silu_mul", "rmsnorm"):
                    outs = [k.get("out", r if torch.is_tensor(r) else None)]
                elif name == "silu_mul_parts":
                    outs = [a[1]]
                elif name == "gemm_smallm":
                    outs = [k.get("out", r if torch.is_tensor(r) else None)]
                elif name == "decode_advance":
                    outs = [a[1]]
                elif name == "text_predictor_decide":
                    outs = [a[4], a[5]]
                self.log.append((name,) + tuple(csum(o) for o in outs if o is not None))
            return r

        return w

    def restore(self):
        import dynamic_llava_amd.model as M

        for n, f in self._orig.items():
            setattr(self.ops, n, f)
        M.F.linear = self._lin


def traced_decode(model, tr, ids, am, imgs, new):
    """Prefill through the graph path as generate() does, decode steps eagerly with the tracer on."""
    g = model.use_hip_graph
    tr.log, tr.on, tr.tp_calls = [], False, []
    model.use_hip_graph = False
    try:
        tr.on = True
        out = model.generate(ids, attention_mask=am, images=imgs, max_new_tokens=new, eos_token_id=None)
    finally:
        tr.on = False
        model.use_hip_graph = g
    return out.cpu(), list(tr.log)


def first_log_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return {"index": i, "a": str(x), "b": str(y), "prev": str(a[i - 1]) if i else None}
    return None


def tp_detail(a_calls, b_calls):
    """First text-predictor call whose stage outputs differ between two traced runs: which stage, where, and whether the observed value is
    the one the PREVIOUS call left in the same workspace slot (a stale read) or something else."""
    for ci, (ca, cb) in enumerate(zip(a_calls, b_calls)):
        names = ("x_before", "x_after", "h1", "a1", "logits")
        eq = {n: bool(torch.equal(u, v)) for n, u, v in zip(names, ca, cb)}
        if all(eq.values()):
            continue
        rec = {"call": ci, "equal": eq}
        for n, u, v, k in (("h1", ca[2], cb[2], 2), ("a1", ca[3], cb[3], 3), ("logits", ca[4], cb[4], 4), ("x_before", ca[0].float(), cb[0].float(), 0)):
            ne = torch.nonzero(u != v)
            if ne.numel():
                idx = ne[:6].tolist()
                prev = a_calls[ci - 1][k] if ci > 0 else None
                rec[n] = {"n_diff": int(ne.shape[0]), "rows": sorted(set(ne[:, 0].tolist()))[:16], "cols": sorted(set(ne[:, 1].tolist()))[:24], "where": idx,
                          "solo": [float(u[i, j]) for i, j in idx], "contended": [float(v[i, j]) for i, j in idx],
                          "solo_previous_call": None if prev is None else [float(prev.float()[i, j]) for i, j in idx]}
        return rec
    return None


def stress_main(args):
    """Second process: keeps the GPU busy the way another rank would."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.stress == "gemm":
        a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        open(args.ready, "w").write("1")
        while not os.path.exists(args.stop):
            for _ in range(50):
                a @ b
            torch.cuda.synchronize()
        return
    if args.stress == "stream":
        a = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
        b = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
        open(args.ready, "w").write("1")
        while not os.path.exists(args.stop):
            for _ in range(50):
                b.copy_(a)
            torch.cuda.synchronize()
        return
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig(num_hidden_layers=args.layers)
    model = build_random_model(cfg, dtype=torch.bfloat16, device=dev, seed=0, predictor_gain=50.0)
    ids, am, imgs = chunk(cfg, 0, 2, dev, torch.bfloat16)
    mode = os.environ.get("DP_STRESS_MODE", "")  # bisecting WHAT in the other rank's workload disturbs this one
    new = args.new
    if "nograph" in mode:
        model.use_hip_graph = False
    if "notp" in mode:
        model.config.sparse_config["use_output_text_predictor"] = False
    if "novp" in mode:
        model.config.sparse_config["use_vision_predictor"] = False
    if "prefill" in mode:
        new = 1
    if "b1" in mode:
        ids, am, imgs = ids[:1, :56].contiguous(), None, imgs[:1].contiguous()
    kw = dict(attention_mask=am, images=imgs, max_new_tokens=new, eos_token_id=None)
    if "feats" in mode:  # no CLIP tower in the loop
        kw.pop("images")
        kw["image_features"] = model.encode_images(imgs)
    model.generate(ids, **kw)
    torch.cuda.synchronize()
    open(args.ready, "w").write("1")
    while not os.path.exists(args.stop):
        model.generate(ids, **kw)
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--new", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--stress", default="model")
    ap.add_argument("--role", default="main")
    ap.add_argument("--ready", default="/tmp/dp_ready")
    ap.add_argument("--stop", default="/tmp/dp_stop")
    ap.add_argument("--no-calibrate", action="store_true")
    ap.add_argument("--tp-detail", action="store_true", help="only: eager traced runs alone and under each --stress kind (comma list), text-predictor tensors compared")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diverge_probe.json"))
    args = ap.parse_args()
    if args.role == "stress":
        return stress_main(args)

    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dtype = torch.bfloat16
    cfg = DynamicLlavaConfig(num_hidden_layers=args.layers)
    model = build_random_model(cfg, dtype=dtype, device=dev, seed=0, predictor_gain=50.0)
    rep = {"layers": args.layers, "new": args.new, "stress": args.stress}
    if not args.no_calibrate:  # bench.py calibrates the text predictor before the configs3 leg: same decisions-at-the-boundary regime
        sys.path.insert(0, ROOT)
        import bench

        p1, i1 = bench.make_inputs(cfg, dev, dtype)
        rep["calibrated_keep_fraction"] = bench.calibrate_text_predictor(model, p1, i1, 64)
    c1 = chunk(cfg, 1, 2, dev, dtype)
    c0 = chunk(cfg, 0, 2, dev, dtype)

    def save():
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1, default=str)

    if args.tp_detail:
        tr = Tracer(model)
        ids_a, log_a = traced_decode(model, tr, *c1, args.new)
        calls_a = tr.tp_calls
        ids_b, log_b = traced_decode(model, tr, *c1, args.new)
        rep["solo_repeat"] = {"ids_equal": bool(torch.equal(ids_a, ids_b)), "first_trace_diff": first_log_diff(log_a, log_b), "tp": tp_detail(calls_a, tr.tp_calls)}
        save()
        for kind in [k for k in args.stress.split(",") if k and k != "none"]:
            for f in (args.ready, args.stop):
                if os.path.exists(f):
                    os.remove(f)
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--role", "stress", "--stress", kind, "--layers", str(args.layers), "--new", "32", "--ready", args.ready, "--stop", args.stop])
            t0 = time.time()
            while not os.path.exists(args.ready) and time.time() - t0 < 600 and p.poll() is None:
                time.sleep(0.5)
            res = []
            try:
                for _ in range(args.reps):
                    ids_c, log_c = traced_decode(model, tr, *c1, args.new)
                    res.append({"ids_equal": bool(torch.equal(ids_a, ids_c)), "first_trace_diff": first_log_diff(log_a, log_c), "tp": tp_detail(calls_a, tr.tp_calls)})
            finally:
                open(args.stop, "w").write("1")
                try:
                    p.wait(timeout=120)
                except Exception:
                    p.kill()
            rep[f"under_{kind}"] = res
            save()
        tr.restore()
        print(json.dumps(rep, indent=1, default=str))
        return
    # ---- A. alone: first call (capture) vs replays ----
    solo = [run_once(model, *c1, args.new, scores=False) for _ in range(args.reps)]
    rep["A_solo_first_vs_replays"] = [diff(solo[0], s) for s in solo[1:]]
    kv_ref = prefill_kv(model, *c1)
    rep["A_kv_after_prefill_repeatable"] = kv_ref == prefill_kv(model, *c1)
    save()
    # ---- B. alone: after the other chunk went through the pooled slab and the graph caches ----
    run_once(model, *c0, args.new)
    after = [run_once(model, *c1, args.new) for _ in range(2)]
    rep["B_solo_after_other_chunk"] = [diff(solo[0], s) for s in after]
    rep["B_kv_after_prefill_equal"] = kv_ref == prefill_kv(model, *c1)
    # fresh caches (what another RANK has: its own first call on this chunk)
    model._dstate, model._prefill_graphs, model._cache_pool = None, {}, None
    fresh = run_once(model, *c1, args.new)
    rep["B_solo_fresh_state"] = diff(solo[0], fresh)
    save()
    # ---- C. eager trace alone (reference log for the contention run) ----
    tr = Tracer(model)
    ids_e, log_a = traced_decode(model, tr, *c1, args.new)
    ids_e2, log_b = traced_decode(model, tr, *c1, args.new)
    rep["C_eager_equals_graph"] = bool(torch.equal(ids_e, solo[0]["ids"]))
    rep["C_eager_trace_repeatable"] = first_log_diff(log_a, log_b) is None
    rep["C_trace_entries"] = len(log_a)
    save()
    # ---- D. with a second process on the GPU ----
    if args.stress != "none":
        for f in (args.ready, args.stop):
            if os.path.exists(f):
                os.remove(f)
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--role", "stress", "--stress", args.stress, "--layers", str(args.layers), "--new", str(args.new),
                              "--ready", args.ready, "--stop", args.stop])
        t0 = time.time()
        while not os.path.exists(args.ready) and time.time() - t0 < 600 and p.poll() is None:
            time.sleep(0.5)
        rep["D_stress_started"] = os.path.exists(args.ready)
        try:
            cont = [run_once(model, *c1, args.new) for _ in range(max(args.reps, 4))]
            rep["D_contended_vs_solo"] = [diff(solo[0], s) for s in cont]
            rep["D_kv_after_prefill_equal"] = [kv_ref == prefill_kv(model, *c1) for _ in range(3)]
            save()
            sc = [run_once(model, *c1, args.new, scores=True) for _ in range(3)]
            rep["D_contended_scores_mode"] = [diff(sc[0], s) for s in sc[1:]]
            tl = []
            for _ in range(3):
                ids_c, log_c = traced_decode(model, tr, *c1, args.new)
                tl.append({"ids_equal_solo_eager": bool(torch.equal(ids_c, ids_e)), "first_trace_diff": first_log_diff(log_a, log_c)})
            rep["D_contended_eager_trace"] = tl
            save()
            # component swaps under contention (graph path)
            variants = {}
            for name, setup in (("smallm32", lambda m: setattr(m, "smallm_max_decode_batch", 32)), ("no_inkernel_combine", lambda m: setattr(m, "attn_inkernel_combine", False)),
                                ("no_text_predictor", lambda m: m.config.sparse_config.__setitem__("use_output_text_predictor", False))):
                keep = (model.smallm_max_decode_batch, model.attn_inkernel_combine, model.config.sparse_config["use_output_text_predictor"])
                setup(model)
                model._dstate, model._prefill_graphs = None, {}
                runs = [run_once(model, *c1, args.new) for _ in range(4)]
                variants[name] = [diff(runs[0], s) for s in runs[1:]]
                model.smallm_max_decode_batch, model.attn_inkernel_combine = keep[0], keep[1]
                model.config.sparse_config["use_output_text_predictor"] = keep[2]
                model._dstate, model._prefill_graphs = None, {}
            rep["D_variants_self_consistency_under_contention"] = variants
        finally:
            open(args.stop, "w").write("1")
            try:
                p.wait(timeout=120)
            except Exception:
                p.kill()
    tr.restore()
    save()
    print(json.dumps(rep, indent=1, default=str))


if __name__ == "__main__":
    main()
