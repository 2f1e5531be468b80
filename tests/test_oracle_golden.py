"""CPU: the oracle (oracle/ref_cpu.py) against the committed golden vectors, which were produced by
running the reference itself (oracle/make_golden.py).  Bit-exact: same torch ops in the same order."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle.make_golden import CASES, SD_SEED, case_config
from oracle.ref_cpu import Oracle, topk_keep_index


def _run_case(name, golden_dir, tie_break):
    c = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    dtype = getattr(torch, c["dtype"])
    cfg = case_config(c)
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=c["gain"])
    clip = fx.build_clip(cfg, seed=1)
    o = Oracle(cfg, sd, dtype, clip=clip, tie_break=tie_break)
    ids = torch.from_numpy(g["input_ids"])
    B = ids.shape[0]
    images = fx.make_images(cfg, B, seed=0).to(dtype)
    forced = torch.from_numpy(g["forced"]) if "forced" in g.files else None
    pkv, cur = None, ids
    with torch.no_grad():
        for j in range(g["step_logits"].shape[0]):
            logits, pkv = o.forward(cur, images=images if j == 0 else None, past_key_values=pkv)
            last = logits[:, -1].float().numpy()
            np.testing.assert_array_equal(last, g["step_logits"][j], err_msg=f"{name} step {j}")
            if j == 0:
                assert tuple(logits.shape) == tuple(g["prefill_logits_shape"])
                np.testing.assert_array_equal(o.records["position_ids"].numpy(), g["position_ids"])
                if "vision_logit" in g.files:
                    np.testing.assert_array_equal(o.records["vision_logit"].float().numpy(), g["vision_logit"])
            td = o.records.get("text_decision")
            tdn = np.full((B,), -1) if td is None else td[:, 0].long().numpy()
            np.testing.assert_array_equal(tdn, g["text_decision"][j])
            np.testing.assert_array_equal(pkv[1][0].numpy(), g["len_first"][j])
            np.testing.assert_array_equal(pkv[1][-1].numpy(), g["len_last"][j])
            assert pkv[0][0][0].shape[-2] == g["kv_len_first"][j]
            assert pkv[0][-1][0].shape[-2] == g["kv_len_last"][j]
            nxt = logits[:, -1].argmax(-1)
            np.testing.assert_array_equal(nxt.numpy(), g["ids"][j])
            cur = nxt[:, None] if forced is None else forced[j][:, None]


STD = sorted(n for n in CASES if not CASES[n].get("nocache") and not CASES[n].get("rounds"))
ROUNDS = sorted(n for n in CASES if CASES[n].get("rounds"))


@pytest.mark.parametrize("name", ROUNDS)
def test_oracle_multiround_matches_reference_golden(name, golden_dir):
    """multi-token chunks on a non-empty cache (DML:2506-2521 with the instruct predictor, plain chunked prefill
    without), interleaved with single-token decode steps."""
    c = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    dtype = getattr(torch, c["dtype"])
    cfg = case_config(c)
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=c["gain"])
    o = Oracle(cfg, sd, dtype, clip=fx.build_clip(cfg, seed=1), tie_break="torch")
    images = fx.make_images(cfg, 1, seed=0).to(dtype)
    pkv = None
    with torch.no_grad():
        for j in range(int(g["n_calls"])):
            ids = torch.from_numpy(g[f"call_ids_{j}"])
            logits, pkv = o.forward(ids, images=images if j == 0 else None, past_key_values=pkv)
            np.testing.assert_array_equal(logits[:, -1].float().numpy(), g["step_logits"][j], err_msg=f"call {j}")
            np.testing.assert_array_equal(pkv[1][0].numpy(), g["len_first"][j])
            np.testing.assert_array_equal(pkv[1][-1].numpy(), g["len_last"][j])
            assert pkv[0][0][0].shape[-2] == g["kv_len_first"][j] and pkv[0][-1][0].shape[-2] == g["kv_len_last"][j]
            td = o.records.get("text_decision")
            if g[f"decision_{j}"].size:
                np.testing.assert_array_equal(td.long().numpy(), g[f"decision_{j}"])
NOCACHE = sorted(n for n in CASES if CASES[n].get("nocache"))


@pytest.mark.parametrize("name", NOCACHE)
def test_oracle_nocache_matches_reference_golden(name, golden_dir):
    """use_cache=False decode (DML:2393-2504), incl. the reference's first-call quirk (last token duplicated)."""
    c = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    dtype = getattr(torch, c["dtype"])
    cfg = case_config(c)
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=c["gain"])
    o = Oracle(cfg, sd, dtype, clip=fx.build_clip(cfg, seed=1), tie_break="torch")
    total = torch.from_numpy(g["input_ids"])
    images = fx.make_images(cfg, total.shape[0], seed=0).to(dtype)
    forced = torch.from_numpy(g["forced"])
    with torch.no_grad():
        for j in range(g["step_logits"].shape[0]):
            logits, pkv = o.forward(total, images=images, use_cache=False)
            assert pkv is None and logits.shape[1] == g["logits_len"][j]
            np.testing.assert_array_equal(logits[:, -1].float().numpy(), g["step_logits"][j])
            np.testing.assert_array_equal(o.records["position_ids"].numpy(), g[f"position_ids_{j}"])
            total = torch.cat([total, forced[j][:, None]], dim=1)
    assert g["logits_len"][0] == g[f"position_ids_0"].shape[1] and g["position_ids_0"][0, -1] == g["position_ids_0"][0, -2]  # the quirk


@pytest.mark.parametrize("name", STD)
def test_oracle_matches_reference_golden(name, golden_dir):
    # tie_break="torch" calls argsort exactly as the reference does (DML:1902-1908)
    _run_case(name, golden_dir, "torch")


@pytest.mark.parametrize("name", [n for n in STD if "ties" not in n])
def test_pinned_tiebreak_equals_reference_when_no_ties(name, golden_dir):
    # with distinct boundary scores the pinned (stable) rule must select the same set as the reference
    _run_case(name, golden_dir, "stable")


def test_topk_tie_rule():
    s = torch.tensor([[0.5, 1.0, 0.5, 0.5, 2.0, 0.5]])
    assert topk_keep_index(s, 3, "stable").tolist() == [[0, 1, 4]]
    assert topk_keep_index(s, 4, "stable").tolist() == [[0, 1, 2, 4]]
    assert topk_keep_index(s, 6, "stable").tolist() == [[0, 1, 2, 3, 4, 5]]


def test_dense_keep_rate_one_is_identity(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_fp32_dense.npz"))
    n = g["position_ids"].shape[1]
    np.testing.assert_array_equal(g["position_ids"][0], np.arange(n))
    assert g["kv_len_last"][0] == g["kv_len_first"][0]


# ---- N5: training-time ops (goldens from oracle/make_golden_train.py = the reference's own functions + autograd) ----
def _train_golden():
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", "train_ops.npz"))


@pytest.mark.parametrize("name,dtype,causal,n_pad", [("f32_causal", torch.float32, True, 0), ("bf16_causal", torch.bfloat16, True, 0),
                                                     ("f32_mask", torch.float32, False, 7), ("bf16_mask", torch.bfloat16, False, 7)])
def test_train_ops_sdpa_with_policy_oracle_equals_reference(name, dtype, causal, n_pad):
    """oracle.sdpa_with_policy (DML:913-970 restated) forward and autograd gradients == the reference's, bit for bit."""
    from oracle import ref_cpu as O
    from oracle.make_golden_train import case_inputs, hf_mask

    G = _train_golden()
    q, k, v, do, pol = case_inputs(11, 2, 2, 40, 32, dtype)
    q, k, v, pol = (t.clone().requires_grad_(True) for t in (q, k, v, pol))
    mask = None if causal else hf_mask(2, 40, n_pad, dtype)
    o = O.sdpa_with_policy(q, k, v, attn_mask=mask, is_causal=causal, policy=pol)
    o.backward(do)
    for key, t in [("o", o), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad), ("dpolicy", pol.grad)]:
        np.testing.assert_array_equal(t.detach().float().numpy(), G[f"sdpa_{name}_{key}"], err_msg=key)


def test_train_ops_softmax_with_policy_oracle_equals_reference():
    from oracle import ref_cpu as O

    G = _train_golden()
    out = O.softmax_with_policy(torch.from_numpy(G["softmax_in_attn"]), torch.from_numpy(G["softmax_in_policy"]))
    np.testing.assert_array_equal(out.numpy(), G["softmax_out"])
    # a dropped key keeps weight on the diagonal only (up to the eps / N floor)
    pol = G["softmax_in_policy"][0, :, 0]
    j = int(np.where(pol == 0)[0][0])
    off_diag = np.delete(out.numpy()[0, 0, :, j], j)
    assert off_diag.max() < 1e-6 and out.numpy()[0, 0, j, j] > 1e-4


@pytest.mark.parametrize("name,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_train_ops_gumbel_hard_keep_oracle_equals_reference(name, dtype):
    from oracle import ref_cpu as O

    G = _train_golden()
    lp = torch.from_numpy(G[f"gumbel_{name}_logp"]).to(dtype).requires_grad_(True)
    keep = O.gumbel_hard_keep(lp, torch.from_numpy(G[f"gumbel_{name}_noise"]).to(dtype), 0.7, torch.from_numpy(G[f"gumbel_{name}_prev"]).to(dtype))
    keep.backward(torch.from_numpy(G[f"gumbel_{name}_w"]).to(dtype))
    np.testing.assert_array_equal(keep.detach().float().numpy(), G[f"gumbel_{name}_keep"])
    np.testing.assert_array_equal(lp.grad.float().numpy(), G[f"gumbel_{name}_dlogp"])
