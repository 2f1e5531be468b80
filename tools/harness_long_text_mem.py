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


class _Curve:
    """The record the reference harness writes (same JSON keys, BLTM:128-142 / 346-351) plus the per-step time."""

    KEYS = ("total_token_length", "kv_cache_length", "max_memory", "without_model_memory", "step_time_ms")

    def __init__(self, model_memory, result_file):
        self.rows = {k: [] for k in self.KEYS}
        self.model_memory, self.result_file = model_memory, result_file

    def add(self, n_tokens, kv_len, ms):
        peak = torch.cuda.max_memory_allocated()
        for k, v in zip(self.KEYS, (n_tokens, int(kv_len), peak, peak - self.model_memory, ms)):
            self.rows[k].append(v)
        if self.result_file:  # rewritten after every token, so that an interrupted run leaves its curve behind
            with open(self.result_file, "w", encoding="utf-8") as f:
                json.dump(self.rows, f, ensure_ascii=False, indent=4)


def _tokens_fed(step_ids, step_images, patch=14):
    """The reference's accounting of one call (BLTM:317-323): every image patch counts, the <image> placeholder does not."""
    n = step_ids.shape[-1]
    if step_images is not None:
        n += (step_images.shape[-2] // patch) * (step_images.shape[-1] // patch) - 1
    return n


@torch.inference_mode()
def run(model, cfg, input_ids, label_ids, images, model_memory=0, result_file=None, verbose=True):
    """Teacher-forced walk over `label_ids` [B, T] starting from `input_ids` [B, n] (one -200 per row): call 0 is the multimodal prefill,
    every later call feeds the previous label column with the cache the model handed back; after each call the LAST layer's KV length is
    read through the legacy indexing the reference harness uses (`pkv[0][-1][0].shape[-2]`).  Returns the record dict."""
    curve = _Curve(model_memory, result_file)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fed, pkv = 0, None
    feeds = [(input_ids, images)] + [(label_ids[:, j : j + 1], None) for j in range(label_ids.shape[1] - 1)]
    for step_ids, step_images in feeds:
        fed += _tokens_fed(step_ids, step_images)
        t0.record()
        pkv = model(step_ids, images=step_images, past_key_values=pkv).past_key_values
        t1.record()
        torch.cuda.synchronize()
        curve.add(fed, pkv[0][-1][0].shape[-2], t0.elapsed_time(t1))
        if verbose and fed % 100 == 0:
            r = curve.rows
            print(f"[{fed} tokens] last-layer KV {r['kv_cache_length'][-1]}, peak {r['max_memory'][-1] / 2**30:.3f} GiB "
                  f"({r['without_model_memory'][-1] / 2**30:.3f} GiB beyond the weights)")
    return curve.rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--gen-len", type=int, default=512)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--keep-rate", type=float, default=0.2)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--result-file", default=None)
    ap.add_argument("--no-operand-copies", action="store_true", help="build without the operand-order weight copies: `model memory` = the parameters, as in the reference")
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
    model = build_random_model(cfg, dtype=torch.float16, device="cuda", seed=0, predictor_gain=50.0)
    for k, v in saved_env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    model_memory = torch.cuda.max_memory_allocated()
    #  memory the reference does not hold for the same checkpoint, reported beside BLTM:64's figure
    copies, param_bytes, live = model.operand_copy_bytes(), model.parameter_bytes(), torch.cuda.memory_allocated() - live_before_build
    print("model memory: " + str(model_memory))
    print("operand-order weight copies: " + str(copies["total"]) + " (" + ", ".join(f"{k} {v}" for k, v in copies.items() if k != "total") + ")")
    print("model memory without operand-order weight copies: " + str(live - copies["total"]) + " (parameters + buffers: " + str(param_bytes) + ")")
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, cfg.vocab_size, (args.prompt_len,), generator=g)
    row = torch.cat([torch.tensor([1]), ids[: args.prompt_len // 2], torch.tensor([-200]), ids[args.prompt_len // 2 :]])
    input_ids = row[None].repeat(args.batch_size, 1).cuda()
    label_ids = torch.randint(3, cfg.vocab_size, (1, args.gen_len), generator=g).repeat(args.batch_size, 1).cuda()
    s = cfg.clip["image_size"]
    images = torch.randn((1, 3, s, s), generator=g).to("cuda", dtype=torch.float16).repeat(args.batch_size, 1, 1, 1)
    rec = run(model, cfg, input_ids, label_ids, images, model_memory, args.result_file)
    rec.update(model_memory=model_memory, operand_copy_bytes=copies, parameter_bytes=param_bytes, model_memory_without_operand_copies=live - copies["total"],
               operand_copies_built=not args.no_operand_copies)
    if args.result_file:
        with open(args.result_file, "w", encoding="utf-8") as f:
            json.dump(rec, f, ensure_ascii=False, indent=4)
    n_img = (s // cfg.clip["patch_size"]) ** 2
    dense = args.prompt_len + n_img + args.gen_len - 1
    print(f"final: total_token_length {rec['total_token_length'][-1]}, kv_cache_length (last layer) {rec['kv_cache_length'][-1]} "
          f"(dense would be {dense}), without_model_memory {rec['without_model_memory'][-1] / 2**20:.1f} MiB, "
          f"median step {sorted(rec['step_time_ms'][1:])[len(rec['step_time_ms'][1:]) // 2]:.3f} ms")
    return rec


if __name__ == "__main__":
    main()
