#!/usr/bin/env python
"""Counterpart of llava/dynamic_eval/bench_test/dynamic_llava_long_text_mem.py:48-393 (BLTM) on the MI355X-native path: KV-cache length
and memory versus generated length.

The reference loop (BLTM:310-351): for every label token j, `outputs = model(input_ids, images=images if j == 0 else None,
past_key_values=past_key_values)` timed with an event pair, then `input_ids = label_id` (teacher forcing), `total_cache_length =
past_key_values[0][-1][0].shape[-2]` and a record {total_token_length, kv_cache_length, max_memory, without_model_memory} rewritten
to --result-file after every token.  This script drives the same loop through the same API surface; the dataset (LVIS instruct
conversations) does not exist offline, so prompts / label ids are seeded random ids of the dataset's typical lengths.

    python tools/harness_long_text_mem.py --gen-len 512 [--model 7b|13b|tiny] [--batch-size 1] [--result-file out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run(model, cfg, input_ids, label_ids, images, model_memory=0, result_file=None, verbose=True):
    """The BLTM:310-351 loop.  input_ids [B, n] (with one -200 per row), label_ids [B, T].  Returns the record dict."""
    record = {"total_token_length": [], "kv_cache_length": [], "max_memory": [], "without_model_memory": [], "step_time_ms": []}
    start_event = torch.cuda.Event(enable_timing=True)
    end_event = torch.cuda.Event(enable_timing=True)
    total_token_length = 0
    past_key_values = None
    for j in range(label_ids.shape[1]):
        label_id = label_ids[:, j : j + 1]
        if j > 0:
            images = None
        with torch.inference_mode():
            if images is not None:
                total_token_length += images.shape[-2] * images.shape[-1] // 14 // 14
                total_token_length += input_ids.shape[-1] - 1
            else:
                total_token_length += input_ids.shape[-1]
            start_event.record()
            outputs = model(input_ids, images=images, past_key_values=past_key_values)
            end_event.record()
            torch.cuda.synchronize()
            elapsed_time_ms = start_event.elapsed_time(end_event)
        input_ids = label_id
        past_key_values = outputs.past_key_values
        total_cache_length = past_key_values[0][-1][0].shape[-2]
        max_memory = torch.cuda.max_memory_allocated()
        record["total_token_length"].append(total_token_length)
        record["kv_cache_length"].append(int(total_cache_length))
        record["max_memory"].append(max_memory)
        record["without_model_memory"].append(max_memory - model_memory)
        record["step_time_ms"].append(elapsed_time_ms)
        if result_file:
            with open(result_file, "w", encoding="utf-8") as f:
                json.dump(record, f, ensure_ascii=False, indent=4)
        if verbose and total_token_length % 100 == 0:
            print("\n#--------------------------------------------------#")
            print("total_token_length: " + str(total_token_length))
            print("kv_cache_length: " + str(total_cache_length))
            print("max_memory: " + str(max_memory / (1024**3)) + "G")
            print("without_model_memory (kv cache): " + str((max_memory - model_memory) / (1024**3)) + "G")
    return record


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--gen-len", type=int, default=512)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--keep-rate", type=float, default=0.2)
    ap.add_argument("--layers", type=int, default=None)
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
    cfg.sparse_config["vision_keep_rate"] = args.keep_rate
    torch.cuda.reset_peak_memory_stats()
    model = build_random_model(cfg, dtype=torch.float16, device="cuda", seed=0, predictor_gain=50.0)
    model_memory = torch.cuda.max_memory_allocated()
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, cfg.vocab_size, (args.prompt_len,), generator=g)
    row = torch.cat([torch.tensor([1]), ids[: args.prompt_len // 2], torch.tensor([-200]), ids[args.prompt_len // 2 :]])
    input_ids = row[None].repeat(args.batch_size, 1).cuda()
    label_ids = torch.randint(3, cfg.vocab_size, (1, args.gen_len), generator=g).repeat(args.batch_size, 1).cuda()
    s = cfg.clip["image_size"]
    images = torch.randn((1, 3, s, s), generator=g).to("cuda", dtype=torch.float16).repeat(args.batch_size, 1, 1, 1)
    rec = run(model, cfg, input_ids, label_ids, images, model_memory, args.result_file)
    n_img = (s // cfg.clip["patch_size"]) ** 2
    dense = args.prompt_len + n_img + args.gen_len - 1
    print(f"final: total_token_length {rec['total_token_length'][-1]}, kv_cache_length (last layer) {rec['kv_cache_length'][-1]} "
          f"(dense would be {dense}), without_model_memory {rec['without_model_memory'][-1] / 2**20:.1f} MiB, "
          f"median step {sorted(rec['step_time_ms'][1:])[len(rec['step_time_ms'][1:]) // 2]:.3f} ms")
    return rec


if __name__ == "__main__":
    main()
