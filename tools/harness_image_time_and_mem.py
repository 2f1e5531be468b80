#!/usr/bin/env python
"""Counterpart of llava/dynamic_eval/bench_test/dynamic_llava_image_time_and_mem.py:54-156 (BIMG) on the MI355X-native path.

Same workload and record as the reference script: input_ids = [[1, -200, 1]] x batch_size (BIMG:124), one seeded 336x336 image
repeated batch_size times, 20 x generate(min_new_tokens=1, max_new_tokens=1, return_dict_in_generate=True, output_scores=True) with
an event pair around each call (BIMG:128-151), then peak memory and peak-minus-model memory (BIMG:153-156).  No checkpoints or
images exist offline: random-init LLaVA-1.5 weights of the named size and a randn image (say so in the record).

    python tools/harness_image_time_and_mem.py --batch-size 1 [--model 7b|13b] [--keep-rate 0.2] [--result-file out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--keep-rate", type=float, default=0.2)
    ap.add_argument("--layers", type=int, default=None, help="debug: fewer layers")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--result-file", default=None)
    ap.add_argument("--no-operand-copies", action="store_true", help="build without the operand-order weight copies (decoder dl_linear_packed, CLIP / projector "
                    "dl_linear_tiles): `model memory` then equals the reference's figure for the same checkpoint (the parameters), at the library GEMMs' speed")
    args = ap.parse_args(argv)
    knob_env = {"DL_PACKED_GEMM": "0", "DL_CLIP_TILES": "0"} if args.no_operand_copies else {}
    saved_env = {k: os.environ.get(k) for k in knob_env}
    os.environ.update(knob_env)  # read by the model's constructor; restored right after the build (a caller's process keeps its own settings)
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    kw = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40) if args.model == "13b" else {}
    if args.model == "tiny":
        kw = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=2, vocab_size=320)
    if args.layers:
        kw["num_hidden_layers"] = args.layers
    cfg = DynamicLlavaConfig(**kw)
    cfg.sparse_config["vision_keep_rate"] = args.keep_rate
    import gc

    gc.collect()  # (garbage of an earlier model in this process must not be freed in the middle of the measurement)
    torch.cuda.reset_peak_memory_stats()
    live_before_build = torch.cuda.memory_allocated()  # (another model of this process may still be alive: the figures below are THIS model's)
    model = build_random_model(cfg, dtype=torch.float16, device="cuda", seed=0, predictor_gain=50.0)  # the reference's eval loads fp16 (BLD:62)
    for k, v in saved_env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    model_memory = torch.cuda.max_memory_allocated()
    print("model memory: " + str(model_memory))
    # memory the reference does not hold for the same checkpoint: second copies of weights in matrix-core operand order.  Reported
    # separately and subtracted, so that the figure BIMG:59-67 prints stays comparable: parameters only.
    copies = model.operand_copy_bytes()
    param_bytes = model.parameter_bytes()
    live_after_build = torch.cuda.memory_allocated() - live_before_build
    print("operand-order weight copies: " + str(copies["total"]) + " (" + ", ".join(f"{k} {v}" for k, v in copies.items() if k != "total") + ")")
    print("model memory without operand-order weight copies: " + str(live_after_build - copies["total"]) + " (parameters + buffers: " + str(param_bytes) + ")")
    g = torch.Generator().manual_seed(0)
    s = cfg.clip["image_size"]
    images_tensor = torch.randn((1, 3, s, s), generator=g).to("cuda", dtype=torch.float16).repeat(args.batch_size, 1, 1, 1)
    input_ids = torch.tensor([[1, -200, 1]]).cuda().repeat(args.batch_size, 1)
    start_event = torch.cuda.Event(enable_timing=True)
    end_event = torch.cuda.Event(enable_timing=True)
    record = {"batch_size": args.batch_size, "model": args.model, "vision_keep_rate": args.keep_rate, "prefill_time_ms": [],
              "data": "synthetic: random-init weights, randn image (no checkpoints / images offline)"}
    for _ in range(args.reps):
        with torch.inference_mode():
            start_event.record()
            outputs = model.generate(input_ids, images=images_tensor, image_sizes=[(s, s)] * args.batch_size, do_sample=False, num_beams=1,
                                     use_cache=True, output_scores=True, return_dict_in_generate=True, min_new_tokens=1, max_new_tokens=1)
            end_event.record()
            torch.cuda.synchronize()
            elapsed_time_ms = start_event.elapsed_time(end_event)
            print("prefill time: " + str(elapsed_time_ms) + "ms")
            record["prefill_time_ms"].append(elapsed_time_ms)
    assert outputs["sequences"].shape == (args.batch_size, 1) and len(outputs["scores"]) == 1
    max_memory = torch.cuda.max_memory_allocated()
    print("max memory: " + str(max_memory))
    print("without model memory: " + str(max_memory - model_memory))
    n_img = (s // cfg.clip["patch_size"]) ** 2
    record.update(operand_copy_bytes=copies, parameter_bytes=param_bytes, model_memory_without_operand_copies=live_after_build - copies["total"],
                  operand_copies_built=not args.no_operand_copies,
                  memory_note="model_memory is the peak while building (BIMG:59-67's print); model_memory_without_operand_copies = live bytes after the build minus "
                              "the operand-order weight copies = what the reference holds for the same checkpoint (+ this engine's RoPE table and workspaces)")
    record.update(max_memory=max_memory, without_model_memory=max_memory - model_memory, model_memory=model_memory,
                  prompt_tokens=2 + n_img, tokens_after_sparse_layer=2 + int(n_img * args.keep_rate),
                  kv_cache_length_last_layer=int(outputs["past_key_values"][0][-1][0].shape[-2]))
    if args.result_file:
        with open(args.result_file, "w", encoding="utf-8") as f:
            json.dump(record, f, ensure_ascii=False, indent=4)
    return record


if __name__ == "__main__":
    main()
