#!/usr/bin/env python
"""Counterpart of llava/dynamic_eval/model_lvis_multi_round_for_ppl.py:108-220 (the multi-round dialogue driver) on the MI355X-native
path: perplexity of the label answers of a multi-round dialogue, teacher-forced through `model(input_ids, images=, past_key_values=)` on the cache the
model hands back.

The reference loop, per round r and label token j (MRP:155-207):
    outputs = model(input_ids, images=images if (r == 0 and j == 0) else None, past_key_values=past_key_values)
    past_key_values = outputs.past_key_values;  logits_j = outputs.logits[:, -1:, :];  input_ids = label_id_j
so round 0's first call is the multimodal prefill, a later round's first call is a multi-token "USER: ... ASSISTANT:" chunk on the non-empty cache
(the new-instruct round of DML:2506-2521) and every other call a single decode step; the LAST label of a round is a cross-entropy target only -- it is
never fed, so it never enters the cache (reproduced on purpose).  Per round: ppl = exp(cross_entropy(logits, labels)) (MRP:209-213); the script reports
the mean over rounds and the cache-length bookkeeping read through the legacy indexing `past_key_values[0][-1][0].shape[-2]` (MRP:180-200).

The dataset (LVIS instruct conversations) and the tokenizer do not exist offline: prompts / label ids are seeded random ids of typical lengths.

    python tools/harness_multi_round_ppl.py [--model 7b|13b|tiny] [--rounds 3] [--prompt-len 40] [--question-len 12] [--answer-len 24]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


@torch.inference_mode()
def run(model, rounds, images, patch=14, on_call=None):
    """rounds: [(prompt_ids [1, n] (round 0: with one -200), label_ids: 1-D tensor of >= 1 label tokens), ...].  Returns the record the reference script
    accumulates: per-round ppl, their mean, the token / cache length counters of MRP:100-103, 165-200.  on_call(call_index, outputs) is a test hook."""
    pkv = None
    rec = {"round_ppl": [], "total_token_length": 0, "instruct_token_length": 0, "output_token_length": 0, "prefill_cache_length": 0, "output_cache_length": 0,
           "kv_len_last_layer": [], "calls": 0}
    for r, (prompt_ids, label_ids) in enumerate(rounds):
        input_ids = prompt_ids
        logits, labels = [], []
        for j in range(label_ids.numel()):
            step_images = images if (r == 0 and j == 0) else None
            if j == 0:
                if step_images is not None:
                    rec["total_token_length"] += step_images.shape[-2] * step_images.shape[-1] // patch // patch
                rec["total_token_length"] += input_ids.shape[-1]
                rec["instruct_token_length"] += input_ids.shape[-1]
            else:
                rec["total_token_length"] += input_ids.shape[-1]
                rec["output_token_length"] += input_ids.shape[-1]
            out = model(input_ids, images=step_images, past_key_values=pkv)
            pkv = out.past_key_values
            if on_call is not None:
                on_call(rec["calls"], out)
            rec["calls"] += 1
            kv_last = pkv[0][-1][0].shape[-2]
            rec["kv_len_last_layer"].append(int(kv_last))
            if r == 0 and j == 0:
                rec["prefill_cache_length"] = int(kv_last)
            elif j == 0:
                rec["prefill_cache_length"] += int(input_ids.shape[-1])
            if r == len(rounds) - 1 and j == label_ids.numel() - 1:
                rec["output_cache_length"] = int(kv_last) - rec["prefill_cache_length"]
            label = label_ids[j].reshape(1, 1).to(device=prompt_ids.device, dtype=prompt_ids.dtype)
            logits.append(out.logits[:, -1:, :].float())
            labels.append(label)
            input_ids = label
        lg = torch.cat(logits, dim=1).squeeze(0)
        lb = torch.cat(labels, dim=1).squeeze(0)
        rec["round_ppl"].append(float(torch.exp(F.cross_entropy(lg, lb))))
    rec["mean_round_ppl"] = sum(rec["round_ppl"]) / len(rec["round_ppl"])
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--prompt-len", type=int, default=40)
    ap.add_argument("--question-len", type=int, default=12)
    ap.add_argument("--answer-len", type=int, default=24)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--instruct-predictor", action="store_true", help="sparse_config.use_instruct_predictor (the new-instruct path DML:2261-2375 / 2506-2521)")
    ap.add_argument("--result-file", default=None)
    args = ap.parse_args()
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    kw = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40) if args.model == "13b" else {}
    if args.model == "tiny":
        kw = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=2, vocab_size=320)
    if args.layers:
        kw["num_hidden_layers"] = args.layers
    cfg = DynamicLlavaConfig(**kw)
    cfg.sparse_config["use_instruct_predictor"] = bool(args.instruct_predictor)
    model = build_random_model(cfg, dtype=torch.float16, device="cuda", seed=0, predictor_gain=50.0)
    g = torch.Generator().manual_seed(0)
    rnd = lambda n: torch.randint(3, cfg.vocab_size, (n,), generator=g)
    rounds = []
    for r in range(args.rounds):
        if r == 0:
            body = rnd(args.prompt_len)
            prompt = torch.cat([torch.tensor([1]), body[: args.prompt_len // 2], torch.tensor([-200]), body[args.prompt_len // 2 :]])
        else:
            prompt = rnd(args.question_len)  # "USER:" + question + "ASSISTANT:" without the BOS (MRP:148)
        rounds.append((prompt[None].cuda(), rnd(args.answer_len).cuda()))
    s = cfg.clip["image_size"]
    images = torch.randn((1, 3, s, s), generator=g).to("cuda", dtype=torch.float16)
    rec = run(model, rounds, images, patch=cfg.clip["patch_size"])
    print(json.dumps({k: v for k, v in rec.items() if k != "kv_len_last_layer"}))
    if args.result_file:
        with open(args.result_file, "w", encoding="utf-8") as f:
            json.dump(rec, f, indent=2)
    return rec


if __name__ == "__main__":
    main()
