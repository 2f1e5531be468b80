"""CPU, build container only: the oracle (oracle/ref_cpu.py) against the reference RUN LIVE from /root/reference on seeds and
shapes that are NOT among the committed golden vectors.  Skipped where the reference tree is absent (the GPU box): there the
committed tests/golden/*.npz (produced by the same code path, oracle/make_golden.py) pin the oracle instead."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle._ref_import import reference_available
from oracle.ref_cpu import Oracle

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference is not present (reference code never travels)")


@pytest.fixture(scope="module")
def dll():
    from oracle._ref_import import import_reference

    return import_reference()


@pytest.mark.parametrize(
    "dtype,sparse,n_sys,n_q,steps,seed",
    [
        (torch.float32, {}, 4, 9, 5, 7),
        (torch.bfloat16, {}, 6, 5, 4, 8),
        (torch.float32, dict(vision_keep_rate=0.5, use_instruct_predictor=True), 3, 15, 3, 9),
    ],
)
def test_oracle_equals_live_reference(dll, dtype, sparse, n_sys, n_q, steps, seed):
    from oracle.make_golden import run_reference

    cfg = fx.tiny_config(**sparse)
    sd = fx.make_state_dict(cfg, seed=seed, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=seed + 1)
    ids = fx.make_prompt(cfg, n_sys, n_q, seed=seed)[None]
    images = fx.make_images(cfg, 1, seed=seed)
    forced = fx.make_forced_tokens(cfg, steps + 1, 1, seed=seed)
    ref = run_reference(dll, cfg, sd, clip, dtype, ids, images, steps, forced)
    o = Oracle(cfg, sd, dtype, clip=clip, tie_break="torch")  # argsort called exactly as the reference does (DML:1902-1908): ties included
    pkv, cur = None, ids
    with torch.no_grad():
        for j in range(steps + 1):
            logits, pkv = o.forward(cur, images=images.to(dtype) if j == 0 else None, past_key_values=pkv)
            np.testing.assert_array_equal(logits[:, -1].float().numpy(), ref["step_logits"][j], err_msg=f"step {j}")
            if j == 0:
                np.testing.assert_array_equal(o.records["position_ids"].numpy(), ref["position_ids"])
            np.testing.assert_array_equal(pkv[1][-1].numpy(), ref["len_last"][j])
            np.testing.assert_array_equal(pkv[1][0].numpy(), ref["len_first"][j])
            cur = forced[j][:, None]


@pytest.mark.parametrize("side,tmax", [("right", None), ("left", None), ("right", 42), ("left", 42)])
def test_oracle_prepare_inputs_padding_and_truncation_equal_live_reference(dll, side, tmax):
    """ARCH:493-579 -- tokenizer_model_max_length truncation and left / right padding of a ragged batch: embeddings, attention mask,
    position ids and the (shifted / clamped) segment dicts, bit for bit against the reference."""
    from oracle.make_golden import build_reference_model

    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=5, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=6)
    prompts = [fx.make_prompt(cfg, 4, 9, seed=1), fx.make_prompt(cfg, 2, 5, seed=2)]
    W = max(p.shape[0] for p in prompts)
    ids = torch.zeros(2, W, dtype=torch.long)
    am = torch.zeros(2, W, dtype=torch.bool)
    for b, p in enumerate(prompts):
        ids[b, : p.shape[0]] = p
        am[b, : p.shape[0]] = True
    images = fx.make_images(cfg, 2, seed=3)
    model = build_reference_model(dll, cfg, sd, clip, torch.float32)
    model.config.tokenizer_padding_side = side
    model.config.tokenizer_model_max_length = tmax
    pos = torch.arange(W)[None].repeat(2, 1)
    with torch.inference_mode():
        (_, r_pos, r_am, _, r_emb, _), (r_idx,) = model.prepare_inputs_labels_for_multimodal(ids, pos, am, None, None, images)
    cfg.tokenizer_padding_side = side
    cfg.tokenizer_model_max_length = tmax
    o = Oracle(cfg, sd, torch.float32, clip=clip)
    with torch.no_grad():
        (_, o_pos, o_am, _, o_emb, _), (o_idx,) = o.prepare_inputs_labels_for_multimodal(ids, pos, am, None, None, images)
    np.testing.assert_array_equal(o_emb.numpy(), r_emb.numpy())
    np.testing.assert_array_equal(o_am.numpy(), r_am.numpy())
    np.testing.assert_array_equal(o_pos.numpy(), r_pos.numpy())
    assert [{k: list(v) for k, v in d.items()} for d in o_idx] == [{k: [int(v[0]), int(v[1])] for k, v in d.items()} for d in r_idx]
