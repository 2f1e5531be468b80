"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by importing and RUNNING THE REFERENCE
(Osilly/dynamic_llava under /root/reference) in the build container.

    python -m oracle.make_golden            # rewrites tests/golden/

The reference cannot travel to the GPU box, so the vectors are committed (inputs are regenerated from
seeds by oracle/fixtures.py; the files hold the reference's outputs).  The driver loop is the
reference's own hand-rolled greedy loop (llava/dynamic_eval/bench_test/dynamic_llava_long_text_mem.py:
310-337) because HF `generate()` does not exist on transformers 5.x models.
"""
from __future__ import annotations

import copy
import os
import sys
import tempfile

import numpy as np
import torch

from oracle import fixtures as fx
from oracle._ref_import import import_reference

SD_SEED = 2  # chosen so that the output-text predictor yields a keep/evict mix (seeds 0,1,3 are one-sided)
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> dict(dtype, sparse overrides, prompts [(n_sys, n_q)], steps, predictor_gain)
CASES = {
    "tiny_fp32_b1": dict(dtype="float32", sparse={}, prompts=[(5, 7)], steps=12, gain=1.0),
    "tiny_fp32_b1_greedy": dict(dtype="float32", sparse={}, prompts=[(5, 7)], steps=12, gain=50.0, greedy=True),
    "tiny_fp32_b1_gain50": dict(dtype="float32", sparse={}, prompts=[(5, 7)], steps=12, gain=50.0),
    "tiny_bf16_b1_gain50": dict(dtype="bfloat16", sparse={}, prompts=[(5, 7)], steps=12, gain=50.0),
    "tiny_bf16_b1_ties": dict(dtype="bfloat16", sparse={}, prompts=[(5, 7)], steps=6, gain=1.0),
    "tiny_fp16_b1_gain50": dict(dtype="float16", sparse={}, prompts=[(5, 7)], steps=8, gain=50.0),
    "tiny_fp32_dense": dict(dtype="float32", sparse=dict(vision_keep_rate=1.0, use_text_predictor=False, use_output_text_predictor=False), prompts=[(5, 7)], steps=6, gain=1.0),
    "tiny_fp32_keep50": dict(dtype="float32", sparse=dict(vision_keep_rate=0.5), prompts=[(3, 11)], steps=8, gain=50.0),
    "tiny_fp32_benchprompt": dict(dtype="float32", sparse=dict(use_text_predictor=False, use_output_text_predictor=False), prompts=[(1, 1)], steps=4, gain=50.0),
    "tiny_fp32_b3_same": dict(dtype="float32", sparse={}, prompts=[(5, 7)] * 3, steps=8, gain=50.0),
    # instruct predictor on (the training default, train_sparse.py:156) -> prefill drops instruct tokens too
    "tiny_fp32_instruct": dict(dtype="float32", sparse=dict(use_instruct_predictor=True), prompts=[(5, 23)], steps=8, gain=50.0),
    "tiny_bf16_instruct": dict(dtype="bfloat16", sparse=dict(use_instruct_predictor=True), prompts=[(5, 23)], steps=8, gain=50.0),
    # decode WITHOUT KV cache (use_cache=False, DML:2393-2504), driven like
    # llava/dynamic_eval/bench_test/dynamic_llava_long_text_time_with_no_cache.py:319-343 (whole sequence re-run per step)
    # multi-round dialogue -- a new instruct chunk (T > 1) on a non-empty cache (DML:2506-2521); with the
    # instruct predictor off the same call is plain chunked prefill on a cache
    "tiny_fp32_multiround": dict(dtype="float32", sparse=dict(use_instruct_predictor=True), prompts=[(5, 9)], steps=0, gain=50.0, rounds=[3, ("chunk", 8), 4, ("chunk", 5), 3]),
    "tiny_fp32_chunked": dict(dtype="float32", sparse={}, prompts=[(5, 9)], steps=0, gain=50.0, rounds=[2, ("chunk", 7), 3]),
    # ARCH:418-454 pinned against the reference itself: vocab >= 29902 so the "USER:" ids (11889, 29901) exist; two matches inside the
    # instruct span, the instruct predictor (on) only drops tokens after the LAST one
    "tiny_fp32_userprompt": dict(dtype="float32", sparse=dict(use_instruct_predictor=True), prompts=[(5, 26)], steps=3, gain=50.0, vocab=30000, user_at=[3, 14]),
    "tiny_fp32_nocache": dict(dtype="float32", sparse={}, prompts=[(5, 7)], steps=10, gain=50.0, nocache=True),
    "tiny_fp32_nocache_b2": dict(dtype="float32", sparse={}, prompts=[(5, 7), (5, 7)], steps=6, gain=50.0, nocache=True),
}


def case_config(c):
    """The config of a golden case (tiny model; `vocab` overrides the 320-entry test vocabulary)."""
    cfg = fx.tiny_config(**c["sparse"])
    if "vocab" in c:
        cfg.vocab_size = c["vocab"]
    return cfg


def case_prompts(c, cfg):
    """Seeded prompts of a golden case; `user_at`: offsets inside the question where the "USER:" id pair (ARCH:36) is planted."""
    prompts = [fx.make_prompt(cfg, ns, nq, seed=i) for i, (ns, nq) in enumerate(c["prompts"])]
    for off in c.get("user_at", []):
        for p, (ns, _) in zip(prompts, c["prompts"]):
            p[ns + 1 + off] = 11889
            p[ns + 2 + off] = 29901
    return prompts


def build_reference_model(dll, cfg, sd, clip, dtype):
    """Instantiates the reference DynamicLlavaLlamaForCausalLM on `cfg` and loads `sd` + `clip`."""
    from transformers import CLIPImageProcessor

    rcfg = dll.DynamicLlavaConfig(
        hidden_size=cfg.hidden_size,
        intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_key_value_heads,
        vocab_size=cfg.vocab_size,
        max_position_embeddings=cfg.max_position_embeddings,
        rms_norm_eps=cfg.rms_norm_eps,
    )
    rcfg.rope_theta = cfg.rope_theta
    rcfg.rope_scaling = None
    rcfg.sparse_config = copy.deepcopy(cfg.sparse_config)
    rcfg.mm_projector_type = cfg.mm_projector_type
    rcfg.mm_hidden_size = cfg.mm_hidden_size
    rcfg.mm_vision_select_layer = cfg.mm_vision_select_layer
    rcfg.mm_vision_select_feature = cfg.mm_vision_select_feature
    tmp = tempfile.mkdtemp(prefix="dl_clip_")
    clip.float().save_pretrained(tmp)
    CLIPImageProcessor(size={"shortest_edge": cfg.clip["image_size"]}, crop_size=cfg.clip["image_size"]).save_pretrained(tmp)
    rcfg.mm_vision_tower = tmp
    model = dll.DynamicLlavaLlamaForCausalLM(rcfg)
    missing, unexpected = model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    missing = [m for m in missing if "vision_tower" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    vt = model.get_vision_tower()
    vt.load_model()
    model = model.to(dtype).eval()
    return model


def run_reference(dll, cfg, sd, clip, dtype, input_ids, images, steps, forced=None):
    """Greedy loop over reference forward() with hooks capturing layer-`sparse_layer` inputs."""
    model = build_reference_model(dll, cfg, sd, clip, dtype)
    cap = {}
    L = cfg.sparse_config["sparse_layer"]

    def pre(mod, args, kwargs):
        cap["position_ids"] = kwargs.get("position_ids")
        cap["text_decision"] = kwargs.get("text_decision")

    model.model.layers[L].register_forward_pre_hook(pre, with_kwargs=True)
    if hasattr(model.model, "image_score_predictor"):
        model.model.image_score_predictor.register_forward_hook(lambda m, i, o: cap.__setitem__("vision_logit", o))
    out = dict(step_logits=[], text_decision=[], len_first=[], len_last=[], kv_len_last=[], kv_len_first=[], ids=[])
    pkv = None
    cur = input_ids
    imgs = images.to(dtype)
    B = input_ids.shape[0]
    with torch.inference_mode():
        img_feat = model.encode_images(imgs)
        for j in range(steps + 1):
            cap.clear()
            o = model(cur, images=imgs if j == 0 else None, past_key_values=pkv)
            pkv = o.past_key_values
            logits = o.logits[:, -1, :].float()
            out["step_logits"].append(logits.numpy().copy())
            if j == 0:
                out["prefill_logits_shape"] = np.array(o.logits.shape)
                out["position_ids"] = cap["position_ids"].numpy().copy()
                if "vision_logit" in cap:
                    out["vision_logit"] = cap["vision_logit"].float().numpy().copy()
            td = cap.get("text_decision")
            out["text_decision"].append(np.full((B,), -1, dtype=np.int64) if td is None else td[:, 0].long().numpy().copy())
            out["len_first"].append(pkv[1][0].numpy().copy())
            out["len_last"].append(pkv[1][-1].numpy().copy())
            out["kv_len_first"].append(pkv[0][0][0].shape[-2])
            out["kv_len_last"].append(pkv[0][-1][0].shape[-2])
            nxt = logits.argmax(dim=-1)
            out["ids"].append(nxt.numpy().copy())
            cur = nxt[:, None] if forced is None else forced[j][:, None]
    res = {k: (np.stack(v) if isinstance(v, list) else v) for k, v in out.items()}
    res["image_features"] = img_feat.float().numpy()
    return res


def run_reference_nocache(dll, cfg, sd, clip, dtype, input_ids, images, steps, forced):
    """The reference's no-KV-cache loop: model(total_input_ids, images=images, use_cache=False), one appended token per step."""
    model = build_reference_model(dll, cfg, sd, clip, dtype)
    cap = {}
    L = cfg.sparse_config["sparse_layer"]
    model.model.layers[L].register_forward_pre_hook(lambda m, a, kw: cap.__setitem__("position_ids", kw.get("position_ids")), with_kwargs=True)
    out = dict(step_logits=[], logits_len=[], position_ids=[], ids=[])
    total = input_ids
    imgs = images.to(dtype)
    with torch.inference_mode():
        for j in range(steps + 1):
            o = model(total, images=imgs, use_cache=False)
            assert o.past_key_values is None
            logits = o.logits[:, -1, :].float()
            out["step_logits"].append(logits.numpy().copy())
            out["logits_len"].append(o.logits.shape[1])
            out["position_ids"].append(cap["position_ids"].numpy().copy())
            out["ids"].append(logits.argmax(-1).numpy().copy())
            total = torch.cat([total, forced[j][:, None]], dim=1)
    res = {"step_logits": np.stack(out["step_logits"]), "logits_len": np.array(out["logits_len"]), "ids": np.stack(out["ids"])}
    for j, p in enumerate(out["position_ids"]):
        res[f"position_ids_{j}"] = p
    return res


def run_reference_rounds(dll, cfg, sd, clip, dtype, input_ids, images, rounds, seed=0):
    """prefill, then a schedule of single-token decode steps (int n) and multi-token chunks (("chunk", T)) on the growing cache."""
    model = build_reference_model(dll, cfg, sd, clip, dtype)
    cap = {}
    L = cfg.sparse_config["sparse_layer"]
    model.model.layers[L].register_forward_pre_hook(lambda m, a, kw: cap.__setitem__("text_decision", kw.get("text_decision")), with_kwargs=True)
    g = torch.Generator().manual_seed(4000 + seed)
    B = input_ids.shape[0]
    calls = [input_ids]
    for r in rounds:
        if isinstance(r, int):
            calls += [torch.randint(3, cfg.vocab_size, (B, 1), generator=g) for _ in range(r)]
        else:
            calls.append(torch.randint(3, cfg.vocab_size, (B, r[1]), generator=g))
    out = dict(step_logits=[], len_first=[], len_last=[], kv_len_first=[], kv_len_last=[])
    pkv = None
    with torch.inference_mode():
        for j, ids in enumerate(calls):
            cap.clear()
            o = model(ids, images=images.to(dtype) if j == 0 else None, past_key_values=pkv)
            pkv = o.past_key_values
            out["step_logits"].append(o.logits[:, -1, :].float().numpy().copy())
            td = cap.get("text_decision")
            out.setdefault("decisions", []).append(np.zeros((B, 0), dtype=np.int64) if td is None else td.long().numpy().copy())
            out["len_first"].append(pkv[1][0].numpy().copy())
            out["len_last"].append(pkv[1][-1].numpy().copy())
            out["kv_len_first"].append(pkv[0][0][0].shape[-2])
            out["kv_len_last"].append(pkv[0][-1][0].shape[-2])
    res = {k: np.stack(v) for k, v in out.items() if k != "decisions"}
    for j, (ids, d) in enumerate(zip(calls, out["decisions"])):
        res[f"call_ids_{j}"] = ids.numpy()
        res[f"decision_{j}"] = d
    res["n_calls"] = np.array(len(calls))
    return res


def pad_prompts(prompts):
    n = max(p.shape[0] for p in prompts)
    assert all(p.shape[0] == n for p in prompts), "golden cases use equal-length rows (reference B>1 + padding is not a supported eval mode)"
    return torch.stack(prompts)


def main():
    dll = import_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = sys.argv[1:]
    for name, c in CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        dtype = getattr(torch, c["dtype"])
        cfg = case_config(c)
        sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=c["gain"])
        clip = fx.build_clip(cfg, seed=1)
        prompts = case_prompts(c, cfg)
        input_ids = pad_prompts(prompts)
        images = fx.make_images(cfg, len(prompts), seed=0)
        forced = None
        if not c.get("greedy", False) and not c.get("rounds"):  # teacher forcing, like the reference's own loop (BLTM:310-337 feeds label ids)
            forced = fx.make_forced_tokens(cfg, c["steps"] + 1, len(prompts), seed=0)
        if c.get("rounds"):
            res = run_reference_rounds(dll, cfg, sd, clip, dtype, input_ids, images, c["rounds"])
        elif c.get("nocache"):
            res = run_reference_nocache(dll, cfg, sd, clip, dtype, input_ids, images, c["steps"], forced)
        else:
            res = run_reference(dll, cfg, sd, clip, dtype, input_ids, images, c["steps"], forced)
        res["input_ids"] = input_ids.numpy()
        if forced is not None:
            res["forced"] = forced.numpy()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **res)
        if c.get("rounds"):
            print(name, "kv_first", res["kv_len_first"].tolist(), "kv_last", res["kv_len_last"].tolist())
        elif c.get("nocache"):
            print(name, "ids", res["ids"][:, 0].tolist(), "logits_len", res["logits_len"].tolist())
        else:
            print(name, "ids", res["ids"][:, 0].tolist(), "kv_last", res["kv_len_last"].tolist(), "dec", res["text_decision"][:, 0].tolist())


if __name__ == "__main__":
    sys.exit(main())
