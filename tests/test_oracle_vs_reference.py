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
