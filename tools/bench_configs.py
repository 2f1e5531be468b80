#!/usr/bin/env python
"""Informational numbers for the BASELINE configs that are NOT the bench line (bench.py measures configs[1]):
  configs[2]  LLaVA-1.5-7B bf16, batch=32 images, ragged prompts (question lengths ~U[8,64]), 128 decode steps, 1 GPU
  configs[4]  LLaVA-1.5-13B bf16, B=1, total length 2048 (prompt 640 + 1408 decode steps) with output-text KV eviction
Prints one JSON object; the eager PyTorch-ROCm restatement of the reference op sequence (oracle on the GPU) is timed beside
the prefill (its B>1 decode is not a supported mode of the reference: model_vqa_loader.py:109 asserts batch_size == 1)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig


def wall(fn, n=3):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out


res = {}
dev, dt = torch.device("cuda"), torch.bfloat16
which = sys.argv[1:] or ["c2", "c4"]
if "c2" in which:
    cfg = DynamicLlavaConfig()
    model = build_random_model(cfg, dtype=dt, device=dev, seed=0, predictor_gain=50.0)
    g = torch.Generator().manual_seed(1)
    B, T_new = 32, 128
    n_q = torch.randint(8, 65, (B,), generator=g).tolist()
    W = 35 + 1 + max(n_q)
    ids = torch.zeros(B, W, dtype=torch.long); am = torch.zeros(B, W, dtype=torch.long)
    for b in range(B):
        row = torch.cat([torch.tensor([1]), torch.randint(3, 32000, (34,), generator=g), torch.tensor([-200]), torch.randint(3, 32000, (n_q[b],), generator=g)])
        ids[b, : row.numel()] = row; am[b, : row.numel()] = 1
    images = torch.randn(B, 3, 336, 336, generator=g).to(dt).to(dev)
    ids, am = ids.to(dev), am.to(dev)
    n_prompt = sum(35 + 576 + q for q in n_q)
    t_full, out = wall(lambda: model.generate(ids, attention_mask=am, images=images, max_new_tokens=T_new, eos_token_id=None))
    t_pre, _ = wall(lambda: model.generate(ids, attention_mask=am, images=images, max_new_tokens=1, eos_token_id=None))
    lens = model.last_cache.lens.cpu()
    res["configs[2]"] = {"B": B, "prompt_tokens": n_prompt, "new_tokens_per_row": T_new, "step_ms": round(t_full * 1e3, 2), "prefill_ms": round(t_pre * 1e3, 2),
                         "tokens_per_s": round((n_prompt + B * T_new) / t_full, 1), "prefill_tokens_per_s": round(n_prompt / t_pre, 1),
                         "decode_tokens_per_s": round(B * (T_new - 1) / (t_full - t_pre), 1), "decode_ms_per_step": round((t_full - t_pre) / (T_new - 1) * 1e3, 3)}
    try:
        from oracle.ref_cpu import Oracle
        o = Oracle(cfg, {k: v for k, v in model.state_dict().items() if "vision_tower" not in k}, dt, device="cuda", clip=model.model.vision_tower.vision_tower)
        t_ref, _ = wall(lambda: o.forward(ids, attention_mask=am, images=images), 2)
        res["configs[2]"]["ref_gpu_prefill_ms"] = round(t_ref * 1e3, 2)
        res["configs[2]"]["prefill_speedup_vs_ref_gpu"] = round(t_ref / t_pre, 2)
    except Exception as e:
        res["configs[2]"]["ref_gpu_error"] = repr(e)
    del model
    torch.cuda.empty_cache()
if "c4" in which or "c4cal" in which:
    cfg = DynamicLlavaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40)
    model = build_random_model(cfg, dtype=dt, device=dev, seed=0, predictor_gain=50.0)
    g = torch.Generator().manual_seed(2)
    ids = torch.cat([torch.tensor([1]), torch.randint(3, 32000, (34,), generator=g), torch.tensor([-200]), torch.randint(3, 32000, (29,), generator=g)])[None].to(dev)
    images = torch.randn(1, 3, 336, 336, generator=g).to(dt).to(dev)
    T_new = 2048 - 640
    calib = None
    if "c4cal" in which:  # the random-init output-text predictor evicts nearly everything; calibrated like bench.py does (about half kept)
        import bench as _bench
        calib = _bench.calibrate_text_predictor(model, ids, images, 64)
    t_full, out = wall(lambda: model.generate(ids, images=images, max_new_tokens=T_new, eos_token_id=None), 1)
    lens = model.last_cache.lens.cpu().tolist()
    t_pre, _ = wall(lambda: model.generate(ids, images=images, max_new_tokens=1, eos_token_id=None))
    wbytes = sum(p.numel() * 2 for n, p in model.named_parameters() if ".layers." in n or n.startswith("lm_head"))
    dec_ms = (t_full - t_pre) / (T_new - 1) * 1e3
    res["configs[4]"] = {"model": "LLaVA-1.5-13B random init", "prompt_tokens": 640, "new_tokens": T_new, "step_ms": round(t_full * 1e3, 1), "prefill_ms": round(t_pre * 1e3, 2),
                         "decode_ms_per_token": round(dec_ms, 4), "decode_tokens_per_s": round(1e3 / dec_ms, 1), "tokens_per_s": round((640 + T_new) / t_full, 1),
                         "kv_len_layers_0_1": lens[0][0], "kv_len_layers_ge2": lens[1][0], "kept_of_generated": lens[1][0] - 179, "decode_weight_stream_GBps": round(wbytes / dec_ms / 1e6, 1),
                         "text_predictor_calibrated_keep_fraction": calib, "decode_graphs_captured": len(model._dstate.graphs),
                         "split_factors_seen": sorted({k[2] for k in model._dstate.graphs})}
print(json.dumps(res))
