#!/usr/bin/env python
"""Counterpart of llava/dynamic_eval/bench_test/dynamic_llava_long_text_time_with_no_cache.py:300-360 (BLTN) on the MI355X-native path:
decode time WITHOUT a KV cache (the whole sequence is re-run every step, answer tokens compacted by the output-text predictor,
DML:2393-2504).

The reference loop (BLTN:316-357): `total_input_ids` starts as the prompt; for every label token j it times
`model(total_input_ids, images=images, past_key_values=None, use_cache=False)` with an event pair, appends the label token
(teacher forcing) and records {output_token_length, max_memory}.  Same API surface here; seeded random ids replace the LVIS dataset.

    python tools/harness_long_text_time_no_cache.py --gen-len 64 [--model 7b|tiny] [--batch-size 1] [--result-file out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run(model, input_ids, label_ids, images, result_file=None):
    record = {"output_token_length": [], "max_memory": [], "step_time_ms": []}
    start_event, end_event = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total_input_ids, total_time, output_token_length = input_ids, 0.0, 0
    for j in range(label_ids.shape[1]):
        label_id = label_ids[:, j : j + 1]
        with torch.inference_mode():
            start_event.record()
            outputs = model(total_input_ids, images=images, past_key_values=None, use_cache=False)
            end_event.record()
            torch.cuda.synchronize()
            elapsed_time_ms = start_event.elapsed_time(end_event)
            total_time += elapsed_time_ms
        assert outputs.logits.shape[0] == input_ids.shape[0]
        total_input_ids = torch.cat([total_input_ids, label_id], dim=1)
        output_token_length += label_id.shape[1]
        record["output_token_length"].append(output_token_length)
        record["max_memory"].append(torch.cuda.max_memory_allocated())
        record["step_time_ms"].append(elapsed_time_ms)
        if result_file:
            with open(result_file, "w", encoding="utf-8") as f:
                json.dump(record, f, ensure_ascii=False, indent=4)
    record["total_time_ms"] = total_time
    return record


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["7b", "tiny"])
    ap.add_argument("--gen-len", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--keep-rate", type=float, default=0.2)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--result-file", default=None)
    args = ap.parse_args()
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    kw = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=2, vocab_size=320) if args.model == "tiny" else {}
    if args.layers:
        kw["num_hidden_layers"] = args.layers
    cfg = DynamicLlavaConfig(**kw)
    cfg.sparse_config["vision_keep_rate"] = args.keep_rate
    model = build_random_model(cfg, dtype=torch.float16, device="cuda", seed=0, predictor_gain=50.0)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, cfg.vocab_size, (args.prompt_len,), generator=g)
    row = torch.cat([torch.tensor([1]), ids[: args.prompt_len // 2], torch.tensor([-200]), ids[args.prompt_len // 2 :]])
    input_ids = row[None].repeat(args.batch_size, 1).cuda()
    label_ids = torch.randint(3, cfg.vocab_size, (1, args.gen_len), generator=g).repeat(args.batch_size, 1).cuda()
    s = cfg.clip["image_size"]
    images = torch.randn((1, 3, s, s), generator=g).to("cuda", dtype=torch.float16).repeat(args.batch_size, 1, 1, 1)
    rec = run(model, input_ids, label_ids, images, args.result_file)
    t = sorted(rec["step_time_ms"][1:])
    print(f"final: output_token_length {rec['output_token_length'][-1]}, total {rec['total_time_ms']:.1f} ms, median step {t[len(t) // 2]:.3f} ms, "
          f"max_memory {rec['max_memory'][-1] / 2**30:.2f} GiB")
    return rec


if __name__ == "__main__":
    main()
