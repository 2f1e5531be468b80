#!/usr/bin/env python
"""Throughput on a VARIABLE-SHAPE request stream -- what the reference's eval loop actually produces (llava/dynamic_eval/model_vqa_loader.py:123-196:
one question per call, batch size 1, a new prompt width nearly every call, max_new_tokens=128, EOS enabled, greedy).

200 requests (LLaVA-1.5-7B random init, bf16): 35 system tokens + <image> + a question of ~U[8,64] tokens, a fresh image each, greedy, max_new_tokens=128,
eos_token_id=2.  Per request: HIP events around the prefill part (CLIP + projector + 32 layers + first token) and the decode part of generate(), and which
path served the prefill (first sighting of a shape = one eager run; second = capture; then replay).  Prints one JSON object; committed under profiles/.

    python tools/bench_varlen_stream.py [--requests 200] [--new-tokens 128] [--fixed]    (--fixed: the same stream with ONE width, the bench.py prompt)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig

ap = argparse.ArgumentParser()
ap.add_argument("--requests", type=int, default=200)
ap.add_argument("--new-tokens", type=int, default=128)
ap.add_argument("--fixed", action="store_true")
ap.add_argument("--layers", type=int, default=32)
args = ap.parse_args()

dev, dt = torch.device("cuda"), torch.bfloat16
cfg = DynamicLlavaConfig(num_hidden_layers=args.layers)
model = build_random_model(cfg, dtype=dt, device=dev, seed=0, predictor_gain=50.0)
model.record_timing = True
g = torch.Generator().manual_seed(7)
n_q = [20] * args.requests if args.fixed else torch.randint(8, 65, (args.requests,), generator=g).tolist()
reqs = []
for q in n_q:
    body = torch.randint(3, cfg.vocab_size, (35 + q,), generator=g)
    reqs.append(torch.cat([torch.tensor([1]), body[:34], torch.tensor([-200]), body[35:]])[None].to(dev))
images = [torch.randn((1, 3, 336, 336), generator=g).to(dt).to(dev) for _ in range(8)]

# the stream: nothing is warmed up on purpose except the library itself (one throw-away request of a width outside the stream's range)
warm = torch.cat([torch.tensor([1]), torch.randint(3, cfg.vocab_size, (34,), generator=g), torch.tensor([-200]), torch.randint(3, cfg.vocab_size, (160,), generator=g)])[None].to(dev)
model.generate(warm, images=images[0], max_new_tokens=4, eos_token_id=2)
torch.cuda.synchronize()
recs = []
t0 = time.perf_counter()
for i, ids in enumerate(reqs):
    out = model.generate(ids, images=images[i % 8], max_new_tokens=args.new_tokens, do_sample=False, num_beams=1, use_cache=True, eos_token_id=2)
    n_out = int(out.shape[1])  # the harness decodes the ids on the host right away (VQAL:177): same synchronisation point
    recs.append((model.last_timing, ids.shape[1], n_out))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
model.check_device_errors()
rows = []
for tm, W, n_out in recs:
    e = tm["ev"]
    rows.append(dict(path=tm["path"], width=W, new_tokens=n_out, prefill_ms=e[0].elapsed_time(e[1]), decode_ms=e[1].elapsed_time(e[2])))


def stats(xs):
    xs = sorted(xs)
    return None if not xs else {"n": len(xs), "mean": round(sum(xs) / len(xs), 3), "p50": round(xs[len(xs) // 2], 3), "p90": round(xs[int(0.9 * (len(xs) - 1))], 3), "max": round(xs[-1], 3)}


n_prompt = sum(r["width"] - 1 + 576 for r in rows)
n_new = sum(r["new_tokens"] for r in rows)
res = {
    "workload": f"{args.requests} requests, B=1, 35 system tokens + <image> + question ~U[8,64] tokens" + (" (FIXED: 20 tokens)" if args.fixed else "") + f", fresh image per request, greedy, max_new_tokens={args.new_tokens}, eos_token_id=2 (VQAL:123-196)",
    "distinct_widths": len({r["width"] for r in rows}),
    "tokens_per_s": round((n_prompt + n_new) / wall, 1), "requests_per_s": round(args.requests / wall, 3), "wall_s": round(wall, 2),
    "prompt_tokens": n_prompt, "new_tokens": n_new,
    "prefill_ms": {"all": stats([r["prefill_ms"] for r in rows]), **{p: stats([r["prefill_ms"] for r in rows if r["path"] == p]) for p in ("eager", "graph-capture", "graph-replay")}},
    "decode_ms_per_token": stats([r["decode_ms"] / max(1, r["new_tokens"] - 1) for r in rows]),
    "prefill_graph_entries": len(model._prefill_graphs), "decode_graphs": len(model._dstate.graphs),
    "prefill_policy": "shape cache: first sighting eager (once), second sighting captured into a shared-pool hipGraph, then replay; prompt widths bucketed" if getattr(model, "prefill_width_bucket", 0) else "shape cache: first sighting eager (once), second sighting captured, then replay",
}
print(json.dumps(res))
