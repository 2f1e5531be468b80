"""GPU parity tests, model level: the HIP path (dynamic_llava_amd, through the C ABI) against
  (a) the committed golden vectors produced by running the reference (tests/golden/, oracle/make_golden.py), and
  (b) the oracle (oracle/ref_cpu.py, pinned bit-exact to the reference) on fresh seeded inputs,
on identical weights and inputs.

Bar (BASELINE.json north_star): kept-token index sets bit-exact; logits within 1e-3 -- asserted literally in
fp32.  bf16 / fp16 logits cannot be compared to 1e-3 between two different summation orders (1 bf16 ulp of a
logit of magnitude 2 is 1.6e-2): there the requirement is "same noise class as the reference itself", i.e.
|hip - fp32 truth| <= 2 * |reference(same dtype) - fp32 truth| + 2 ulp, plus identical integer decisions."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx  # noqa: E402
from oracle.make_golden import CASES, SD_SEED, case_config  # noqa: E402
from oracle.ref_cpu import Oracle  # noqa: E402

ULP = {torch.float32: 2.0**-23, torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}
NEAR_TIED_KEPT_SETS = []  # golden cases that continued against the oracle on the HIP kept set (recorded, never skipped)


def _build(cfg_ns, sd, clip, dtype):
    from dynamic_llava_amd.builder import build_from_state_dict
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig.from_namespace(cfg_ns)
    return build_from_state_dict(cfg, sd, clip.state_dict() if clip is not None else None, dtype=dtype, device="cuda")


def _golden_setup(name):
    c = CASES[name]
    dtype = getattr(torch, c["dtype"])
    cfg = case_config(c)
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=c["gain"])
    clip = fx.build_clip(cfg, seed=1)
    return c, dtype, cfg, sd, clip


@pytest.mark.parametrize("name", sorted(n for n in CASES if "b3" not in n and not CASES[n].get("nocache") and not CASES[n].get("rounds")))
def test_forward_loop_vs_reference_golden(name, golden_dir):
    """The reference's own driver loop (dynamic_llava_long_text_mem.py:310-337): model(ids, images=..., past_key_values=pkv)."""
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    model.debug_records = {}
    ids = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, ids.shape[0], seed=0).to(dtype).cuda()
    forced = torch.from_numpy(g["forced"]) if "forced" in g.files else None
    tol = 1e-3 if dtype == torch.float32 else None
    # fp32 ground truth for the 16-bit noise-class bound
    if dtype != torch.float32:
        o32 = Oracle(cfg, {k: v.to(dtype) for k, v in sd.items()}, torch.float32, clip=copy.deepcopy(clip).to(dtype))
    pkv, cur = None, ids
    ties = "ties" in name
    alt = None  # set when a 16-bit kept set differs inside the rounding band: the oracle then continues on the HIP path's set
    p32 = None
    for j in range(g["step_logits"].shape[0]):
        out = model(cur, images=images if j == 0 else None, past_key_values=pkv)
        pkv = out.past_key_values
        last = out.logits[:, -1].float().cpu().numpy()
        ref = g["step_logits"][j]
        if j == 0:
            assert tuple(out.logits.shape) == tuple(g["prefill_logits_shape"])
            n_sys = int((g["input_ids"][0] == -200).nonzero()[0][0])
            k = int(fx.n_image_tokens(cfg) * cfg.sparse_config["vision_keep_rate"])
            pos = model.debug_records["position_ids"].cpu().numpy()[None]
            if not ties and dtype != torch.float32 and pos.shape == g["position_ids"].shape and not np.array_equal(pos, g["position_ids"]):
                # 16-bit: the reference's own scores carry ~1 ulp of noise, so two kept sets may differ -- but only among image
                # tokens whose REFERENCE score lies within the rounding band of the k-th reference score
                ref_score = torch.log_softmax(torch.from_numpy(g["vision_logit"]).to(dtype).float(), -1)[0, :, 0].to(dtype).float().numpy()
                kth = np.sort(ref_score)[::-1][k - 1]
                diff = set(pos[0].tolist()) ^ set(g["position_ids"][0].tolist())
                assert all(n_sys <= p_ < n_sys + fx.n_image_tokens(cfg) for p_ in diff), diff
                assert all(abs(ref_score[p_ - n_sys] - kth) <= 4 * ULP[dtype] * max(1.0, abs(kth)) for p_ in diff), (diff, kth)
                # both are valid top-k sets of scores that agree to the last bit of the dtype.  The golden's downstream values belong to
                # the other set, so the rest of this case (logits, decisions, KV lengths) is checked against the ORACLE (pinned to the
                # reference, same dtype) continuing on the HIP path's set -- nothing is skipped; the outcome is recorded.
                alt = Oracle(cfg, {k_: v.to(dtype) for k_, v in sd.items()}, dtype, clip=copy.deepcopy(clip).to(dtype))
                alt.force_keep_index = model.debug_records["keep_index"].cpu().long()
                o32.force_keep_index = alt.force_keep_index
                alt_pkv = None
                NEAR_TIED_KEPT_SETS.append(name)
                print(f"[{name}] kept set differs from the golden inside the rounding band ({sorted(diff)}): continuing against the oracle on the HIP set")
            elif not ties:
                np.testing.assert_array_equal(pos, g["position_ids"], err_msg="kept-token index set / position ids")
            else:  # tie-aware invariant: {s > s_k} subset kept subset {s >= s_k}, |kept| = k
                score = model.debug_records["vision_score"].float().cpu().numpy()[0]
                kept = pos[0][n_sys : n_sys + k] - n_sys
                kth = np.sort(score)[::-1][k - 1]
                assert len(set(kept.tolist())) == k and set(np.nonzero(score > kth)[0]) <= set(kept.tolist()) <= set(np.nonzero(score >= kth)[0])
            if "vision_logit" in g.files:
                vl = model.debug_records["vision_logit"].float().cpu().numpy()
                assert np.abs(vl - g["vision_logit"]).max() < (1e-3 if dtype == torch.float32 else 64 * ULP[dtype] * max(1.0, np.abs(g["vision_logit"]).max()))
        ref_dec, ref_len_first, ref_len_last = g["text_decision"][j], g["len_first"][j], g["len_last"][j]
        if alt is not None:
            with torch.no_grad():
                l_alt, alt_pkv = alt.forward(cur.cpu(), images=images.cpu() if j == 0 else None, past_key_values=alt_pkv)
            ref = l_alt[:, -1].float().numpy()
            td = alt.records.get("text_decision")
            ref_dec = np.full_like(g["text_decision"][j], -1) if td is None else td.reshape(-1).to(torch.int32).numpy()
            ref_len_first, ref_len_last = alt_pkv[1][0].numpy(), alt_pkv[1][-1].numpy()
            np.testing.assert_array_equal(model.debug_records["position_ids"].cpu().numpy().reshape(-1), alt.records["position_ids"].numpy().reshape(-1)) if j == 0 else None
        if tol is not None:
            assert np.abs(last - ref).max() < tol, f"{name} step {j}: max |logit diff| {np.abs(last - ref).max()}"
        elif not ties:
            # fp32 truth in lockstep: the same input tokens as the HIP path (and the same kept set when it was forced)
            with torch.no_grad():
                l32, p32 = o32.forward(cur.cpu(), images=images.cpu().float() if j == 0 else None, past_key_values=p32)
            truth_j = l32[:, -1].numpy()
            err_hip = np.abs(last - truth_j).max()
            err_ref = np.abs(ref - truth_j).max()
            assert err_hip <= 2.0 * err_ref + 2 * ULP[dtype] * np.abs(truth_j).max(), f"{name} step {j}: hip err {err_hip} vs reference err {err_ref}"
        dec = model.debug_records.get("text_decision")
        if j > 0 and (ref_dec >= 0).all() and not ties:
            np.testing.assert_array_equal(dec.cpu().numpy(), ref_dec, err_msg=f"text decision, step {j}")
        if not ties:
            np.testing.assert_array_equal(pkv[1][0].numpy(), ref_len_first)
            np.testing.assert_array_equal(pkv[1][-1].numpy(), ref_len_last)
            if alt is None:
                assert pkv[0][0][0].shape[-2] == g["kv_len_first"][j] and pkv[0][-1][0].shape[-2] == g["kv_len_last"][j]
            else:
                assert pkv[0][0][0].shape[-2] == alt_pkv[0][0][0].shape[-2] and pkv[0][-1][0].shape[-2] == alt_pkv[0][-1][0].shape[-2]
        if dtype == torch.float32:
            np.testing.assert_array_equal(out.logits[:, -1].argmax(-1).cpu().numpy(), g["ids"][j])
        cur = (out.logits[:, -1].argmax(-1) if forced is None else forced[j].cuda())[:, None]


def test_near_tied_kept_sets_are_reported():
    """Runs after the golden cases (file order): says how many 16-bit cases took the oracle-continuation branch on this GPU."""
    print(f"golden cases whose kept set differed from the golden inside the rounding band: {NEAR_TIED_KEPT_SETS or 'none'}")


def test_generate_greedy_matches_reference_golden(golden_dir):
    name = "tiny_fp32_b1_greedy"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    n = g["ids"].shape[0]
    for graph in (False, True):
        model.use_hip_graph = graph
        out = model.generate(ids, images=images, max_new_tokens=n, do_sample=False, num_beams=1, use_cache=True, eos_token_id=None)
        np.testing.assert_array_equal(out.cpu().numpy()[0], g["ids"][:, 0], err_msg=f"graph={graph}")
        c_ = model.last_cache
        # after n tokens, n-1 decode steps have run: lengths == golden step n-1
        np.testing.assert_array_equal(c_[1][-1].numpy(), g["len_last"][n - 1])
        np.testing.assert_array_equal(c_[1][0].numpy(), g["len_first"][n - 1])
    # second call re-uses the slab + captured graph and must be bit-identical (determinism)
    out2 = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=None)
    assert torch.equal(out, out2)
    # EOS handling: make the 4th generated token the EOS -> HF returns tokens up to and including it
    eos = int(g["ids"][3, 0])
    out3 = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=eos)
    first = [int(t) for t in g["ids"][:, 0]].index(eos)
    np.testing.assert_array_equal(out3.cpu().numpy()[0], g["ids"][: first + 1, 0])


def test_generate_schedule_follows_observed_lengths_and_matches_reference_golden(golden_dir):
    """Round 4: the decode attention is scheduled from the lengths the predictor actually leaves (observed on the device chunk by chunk), not
    from the reserved capacity -- so ONE generate() call replays several captured graphs (single-workgroup rows first, split-KV + in-kernel
    merge once the evicted group's bound passes the threshold).  Thresholds are lowered so that the switch happens inside the golden's
    16 tokens; tokens and KV lengths must still equal the reference's (DML:2377-2391, cache_utils.py:153-164)."""
    name = "tiny_fp32_b1_greedy"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    n = g["ids"].shape[0]
    n_sparse0 = int(g["len_last"][0][0])  # layers >= sparse_layer after the prefill
    model.single_split_keys_override = n_sparse0 + 6
    model.min_keys_per_split = 8
    for sync_every in (8, 3):
        model._dstate = None
        out = model.generate(ids, images=images, max_new_tokens=n, do_sample=False, num_beams=1, use_cache=True, eos_token_id=None, sync_every=sync_every)
        np.testing.assert_array_equal(out.cpu().numpy()[0], g["ids"][:, 0])
        np.testing.assert_array_equal(model.last_cache[1][-1].numpy(), g["len_last"][n - 1])
        np.testing.assert_array_equal(model.last_cache[1][0].numpy(), g["len_first"][n - 1])
        splits = sorted({k[2] for k in model._dstate.graphs})  # (k[3]: attention workgroups per head inside the fused launch)  # (layers < sparse_layer, layer sparse_layer, last layer) split factors of each captured step
        assert any(s[2] == 1 for s in splits) and any(s[2] > 1 for s in splits), f"one call must have replayed both schedules: {splits}"
        print(f"sync_every={sync_every}: captured decode graphs by split factors {splits}")
    # the same request again: same observations -> same schedule -> bit-identical
    out2 = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=None, sync_every=3)
    assert torch.equal(out, out2)
    # a forward()-driven loop (the reference's own driver, BLTM:310-337) follows the same schedule rule step by step: every decode step's
    # logits are BIT-identical to generate()'s (HF `scores`), across the switch of kernels
    res = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=None, return_dict_in_generate=True, output_scores=True)
    np.testing.assert_array_equal(res["sequences"].cpu().numpy()[0], g["ids"][:, 0])
    o = model(ids, images=images)
    pkv, seen = o.past_key_values, set()
    for j in range(1, n):
        o = model(res["sequences"][:, j - 1 : j], past_key_values=pkv)
        pkv = o.past_key_values
        seen.add(pkv.n_splits(model.config.num_hidden_layers - 1, model.config.num_attention_heads))
        assert torch.equal(o.logits[:, -1].float(), res["scores"][j].float()), f"decode step {j}: forward() loop and generate() differ"
    assert 1 in seen and max(seen) > 1, f"the forward() loop must have gone through both schedules too: {sorted(seen)}"


def test_batched_ragged_rows_equal_their_b1_runs():
    """SURVEY finding 2: the reference's B>1 decode attends zero-padded slots; parity for batches is defined per
    row against the B=1 reference.  Ragged prompts + per-row eviction, packed varlen, vs three oracle B=1 runs."""
    dtype = torch.float32
    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, dtype)
    prompts = [fx.make_prompt(cfg, 5, 7, seed=0), fx.make_prompt(cfg, 2, 15, seed=1), fx.make_prompt(cfg, 9, 3, seed=2)]
    B, steps = len(prompts), 10
    images = fx.make_images(cfg, B, seed=3)
    forced = fx.make_forced_tokens(cfg, steps, B, seed=5)
    n = max(p.shape[0] for p in prompts)
    ids = torch.zeros(B, n, dtype=torch.long)
    am = torch.zeros(B, n, dtype=torch.long)
    for b, p in enumerate(prompts):
        ids[b, : p.shape[0]] = p
        am[b, : p.shape[0]] = 1
    model.debug_records = {}
    out = model(ids.cuda(), attention_mask=am.cuda(), images=images.cuda())
    pkv = out.past_key_values
    cu = model.debug_records["cu_after"].cpu().tolist()
    hip_steps = [[out.logits[b, cu[b + 1] - cu[b] - 1].cpu() for b in range(B)]]
    hip_dec = []
    for j in range(steps):
        out = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = out.past_key_values
        hip_steps.append([out.logits[b, -1].cpu() for b in range(B)])
        hip_dec.append(model.debug_records["text_decision"].cpu().clone())
    lens_last = pkv[1][-1]
    for b in range(B):
        o = Oracle(cfg, sd, dtype, clip=clip)
        with torch.no_grad():
            l, p = o.forward(prompts[b][None], images=images[b : b + 1])
            assert float((l[0, -1] - hip_steps[0][b]).abs().max()) < 1e-3
            for j in range(steps):
                l, p = o.forward(forced[j][b : b + 1][:, None], past_key_values=p)
                assert float((l[0, -1] - hip_steps[j + 1][b]).abs().max()) < 1e-3, f"row {b} step {j}"
                assert bool(o.records["text_decision"][0, 0]) == bool(hip_dec[j][b])
            assert int(p[1][-1][0]) == int(lens_last[b])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_full_width_slice_vs_oracle(dtype):
    """LLaVA-1.5-7B layer width (H=4096, 32x128 heads, I=11008, 576 image tokens, real-size predictors), 3 layers
    (sparse_layer=2: two full-length layers + one compacted), random image features, prefill + 6 decode steps."""
    cfg = fx.llava7b_config(num_hidden_layers=3)
    cfg.vocab_size = 2048
    sd = fx.make_state_dict(cfg, seed=7, predictor_gain=50.0)
    model = _build(cfg, sd, None, dtype)
    model.debug_records = {}
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(1, 576, 4096, generator=g)
    ids = fx.make_prompt(cfg, 35, 20, seed=4)[None]
    steps = 6
    forced = fx.make_forced_tokens(cfg, steps, 1, seed=6)
    o = Oracle(cfg, sd, dtype)
    o32 = Oracle(cfg, {k: v.to(dtype) for k, v in sd.items()}, torch.float32) if dtype != torch.float32 else o
    with torch.no_grad():
        out = model(ids.cuda(), image_features=feats.to(dtype).cuda())
        l_ref, p_ref = o.forward(ids, image_features=feats.to(dtype))
        keep_ref = o.records["keep_index"]
        l_32, p_32 = (l_ref, None) if o32 is o else o32.forward(ids, image_features=feats.to(dtype).float())
        assert out.logits.shape == l_ref.shape == (1, 35 + 115 + 20, 2048)
        keep = model.debug_records["keep_index"].cpu()
        same_set = torch.equal(keep, keep_ref)
        if dtype == torch.float32:
            assert same_set, "kept-token index set must be identical in fp32"
            assert float((out.logits.cpu() - l_ref).abs().max()) < 1e-3
        else:
            score = model.debug_records["vision_score"].float().cpu()[0]
            kth = torch.sort(score, descending=True).values[114]
            diff = set(keep[0].tolist()) ^ set(keep_ref[0].tolist())
            # index sets may differ only inside the k-th-score rounding band of the 16-bit scores
            assert all(abs(float(score[i] - kth)) <= 4 * ULP[dtype] * max(1.0, abs(float(kth))) for i in diff), diff
        pkv = out.past_key_values
        if dtype != torch.float32 and not same_set:
            # a kept set that differs inside the rounding band of the k-th score: both oracles continue on the HIP path's set (test hook), so that
            # every later step is still compared -- no silent exit
            o.force_keep_index = keep
            l_ref, p_ref = o.forward(ids, image_features=feats.to(dtype))
            o32.force_keep_index = keep
            l_32, p_32 = o32.forward(ids, image_features=feats.to(dtype).float())
        pr, p32 = p_ref, p_32
        n_forced, n_compared = 0, 0

        def oracle_step(orc, j, pkv_, hip_dec, hip_tl):
            """One oracle step; a keep / evict pair inside the boundary band (fixtures.boundary_band) that fell the other way is repeated with the
            HIP path's decision forced, so the caches stay in the same state and the comparison goes on."""
            l_, p_ = orc.forward(forced[j][:, None], past_key_values=pkv_)
            tl_ = orc.records["text_logit"][0, 0]
            if bool(orc.records["text_decision"][0, 0]) != hip_dec:
                assert fx.decision_may_differ(tl_, orc.dtype, hip_tl, dtype), f"step {j}: eviction decision differs away from the boundary (oracle {tl_.tolist()}, hip {hip_tl.tolist()})"
                orc.force_text_decision = torch.tensor([[int(hip_dec)]])
                l_, p_ = orc.forward(forced[j][:, None], past_key_values=pkv_)
                orc.force_text_decision = None
                return l_, p_, 1
            return l_, p_, 0

        for j in range(steps):
            out = model(forced[j][:, None].cuda(), past_key_values=pkv)
            pkv = out.past_key_values
            hip_dec = bool(model.debug_records["text_decision"][0])
            hip_tl = model.debug_records["text_logit"].float().cpu()[0]
            l_ref, pr, f1 = oracle_step(o, j, pr, hip_dec, hip_tl)
            n_forced += f1
            if dtype == torch.float32:
                assert float((out.logits.cpu() - l_ref).abs().max()) < 1e-3, f"step {j}"
            else:
                l_32, p32, f2 = oracle_step(o32, j, p32, hip_dec, hip_tl)
                n_forced += f2
                e_hip = float((out.logits.cpu() - l_32).abs().max())
                e_ref = float((l_ref - l_32).abs().max())
                assert e_hip <= 2.0 * e_ref + 2 * ULP[dtype] * float(l_32.abs().max()), (j, e_hip, e_ref)
            n_compared += 1
        assert n_compared == steps, "every decode step is compared"
        assert n_forced <= 2 * fx.MAX_FORCED_DECISIONS, f"{n_forced} decisions had to be forced over {steps} steps: more than a boundary effect"
        assert int(pr[1][-1][0]) == int(pkv[1][-1][0]), "KV length of the evicting layers after the compared steps"


def test_keep_rate_one_equals_dense_path():
    """C1 (BASELINE configs[0]): vision_keep_rate=1.0, text predictors off == dense LLaVA semantics."""
    cfg_s = fx.tiny_config(vision_keep_rate=1.0, use_text_predictor=False, use_output_text_predictor=False)
    cfg_d = fx.tiny_config(use_vision_predictor=False, use_text_predictor=False, use_output_text_predictor=False)
    sd = fx.make_state_dict(cfg_s, seed=SD_SEED)
    clip = fx.build_clip(cfg_s, seed=1)
    ms = _build(cfg_s, sd, clip, torch.float32)
    md = _build(cfg_d, {k: v for k, v in sd.items() if "score_predictor" not in k}, clip, torch.float32)
    ids = fx.make_prompt(cfg_s, 5, 7)[None].cuda()
    images = fx.make_images(cfg_s, 1).cuda()
    a = ms.generate(ids, images=images, max_new_tokens=8, eos_token_id=None)
    b = md.generate(ids, images=images, max_new_tokens=8, eos_token_id=None)
    assert torch.equal(a, b)
    assert torch.equal(ms.last_prefill_logits, md.last_prefill_logits)


def test_reference_api_surface():
    """What the dynamic_eval harness touches."""
    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float16)
    # (iii) runtime toggles, dynamic_llava_image_time_and_mem.py:63-65
    model.model.config.sparse_config["use_vision_predictor"] = True
    model.model.config.sparse_config["use_instruct_predictor"] = False
    model.model.config.sparse_config["use_output_text_predictor"] = False
    # bench prompt [[1,-200,1]] x B, same image repeated (ibid. :112-124), max_new_tokens=1
    ids = torch.tensor([[1, -200, 1]]).cuda().repeat(2, 1)
    img = fx.make_images(cfg, 1).half().cuda().repeat(2, 1, 1, 1)
    out = model.generate(ids, images=img, image_sizes=[(84, 84)] * 2, do_sample=False, num_beams=1, use_cache=True, min_new_tokens=1, max_new_tokens=1)
    assert out.shape == (2, 1) and out.dtype == torch.int64 and int(out[0, 0]) == int(out[1, 0])
    # (iv) hookable predictor module, visualize.py:74
    seen = {}
    h = model.model.image_score_predictor.register_forward_hook(lambda m, i, o: seen.update(x=i[0].shape, o=o.shape))
    model.generate(ids[:1], images=img[:1], max_new_tokens=1)
    h.remove()
    assert tuple(seen["x"]) == (1, 36, 256) and tuple(seen["o"]) == (1, 36, 2)
    # prepare_inputs_labels_for_multimodal tuple format (dynamic_llava_arch.py:594-601) and early-out (:180-188)
    (a, pos, mask, pkv, emb, lab), (idx,) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, img)
    assert a is None and emb.shape == (2, 38, 256) and idx[0] == {"system": [0, 1], "image": [1, 37], "instruct": [37, 38], "answer": [38, 38], "last_instruct": [37, 38]}
    (a, *_), (idx,) = model.prepare_inputs_labels_for_multimodal(ids[:, :1], None, None, None, None, img)
    assert a is not None and idx is None
    with pytest.raises(NotImplementedError):
        model.generate(ids, images=img, inputs_embeds=emb)  # dynamic_llava_llama.py:128-129
    with pytest.raises(ValueError):
        model(None)  # dynamic_modeling_llama.py:1686-1695
    # legacy cache indexing used by the memory bench (dynamic_llava_long_text_mem.py:337-338)
    o = model(ids[:1], images=img[:1])
    pkv = o.past_key_values
    assert pkv[0][-1][0].shape[-2] == 1 + 7 + 1 and pkv[0][0][0].shape[-2] == 38 and len(pkv[1]) == cfg.num_hidden_layers
    assert o.logits.dtype == torch.float32 and o.logits.shape == (1, 9, cfg.vocab_size)


def test_instruct_predictor_generate_equals_reference_golden(golden_dir):
    """SURVEY 8f row N2 (prefill part, DML:2261-2375): with `use_instruct_predictor` (the training default) generate() must
    take the eager, data-dependent-shape path and reproduce the reference's teacher-free greedy tokens of the forward loop."""
    name = "tiny_fp32_instruct"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    out = model.generate(ids, images=images, max_new_tokens=1, eos_token_id=None)
    assert int(out[0, 0]) == int(g["ids"][0, 0])
    assert int(model.last_cache[1][-1][0]) == int(g["len_last"][0][0]) and int(model.last_cache[1][0][0]) == int(g["len_first"][0][0])
    o = Oracle(cfg, sd, dtype, clip=clip)
    ref, _ = o.greedy(ids.cpu(), images=images.cpu(), max_new_tokens=10, eos_token_id=None)
    out = model.generate(ids, images=images, max_new_tokens=10, eos_token_id=None)
    assert out.cpu().tolist() == ref.tolist()
    with pytest.raises(AssertionError):  # DML:2269: the reference asserts B == 1 on this branch
        model.generate(ids.repeat(2, 1), images=images.repeat(2, 1, 1, 1), max_new_tokens=2)


def test_instruct_prefill_graph_is_not_shared_between_prompts_with_different_last_user_offsets():
    """the instruct predictor compacts the LAST instruct span (from the last "USER:" match, ARCH:418-454, DML:2261-2375); the
    span is part of the captured plan while the graph key ignored token values.  Two equally long prompts whose last "USER:" sits at different
    offsets must each equal the oracle, in either order, through first sighting, capture and replay."""
    from dynamic_llava_amd.model import USER_IDS

    c, dtype, cfg, sd, clip = _golden_setup("tiny_fp32_userprompt")
    model = _build(cfg, sd, clip, dtype)
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    base = fx.make_prompt(cfg, 5, 26, seed=4)
    img_pos = int((base == -200).nonzero()[0])
    prompts = []
    for off in (3, 15):
        p_ = base.clone()
        p_[img_pos + 1 + off], p_[img_pos + 2 + off] = USER_IDS[0], USER_IDS[1]
        prompts.append(p_[None])
    o = Oracle(cfg, sd, dtype, clip=clip)
    refs = []
    for p_ in prompts:
        ref, _ = o.greedy(p_, images=images.cpu(), max_new_tokens=6, eos_token_id=None)
        refs.append((ref.tolist(), None))
    kept = []
    for rnd in range(3):  # first sighting, capture, replay -- alternating the two prompts
        for p_, (ref, _) in zip(prompts, refs):
            out = model.generate(p_.cuda(), images=images, max_new_tokens=6, eos_token_id=None)
            assert out.cpu().tolist() == ref, f"round {rnd}: a prompt ran on the other prompt's instruct span"
            kept.append(int(model.last_cache[1][-1][0]))
    assert len({k for k in model._prefill_graphs}) == 2, "one cached prefill per last-instruct span"
    print("kept KV lengths (layers >= sparse_layer) per call:", kept)


@pytest.mark.parametrize("name", ["tiny_fp32_instruct", "tiny_bf16_instruct", "tiny_fp32_userprompt"])
def test_instruct_prefill_is_captured_and_matches_the_eager_path(name, golden_dir):
    """the instruct predictor's prefill compaction (DML:2261-2375) keeps its data-dependent row count on the device
    (dl_compact_rows_by_mask), so generate() captures the whole prefill in a hipGraph: graph replay == eager path (host-side nonzero /
    index_select, one device->host copy) on tokens and KV lengths, and -- fp32 -- == the oracle's greedy continuation."""
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    res = {}
    model.record_timing = True
    paths = []
    for graph in (False, True, True, True):  # host-side eager path; then first sighting (run once eagerly, on the device-side plan), capture, replay
        model.use_hip_graph = graph
        out = model.generate(ids, images=images, max_new_tokens=8, eos_token_id=None)
        lens = model.last_cache[1]
        paths.append(model.last_timing["path"])
        res.setdefault(graph, []).append((out.cpu(), int(lens[0][0]), int(lens[-1][0]), model.last_prefill_logits.float().cpu().clone()))
    assert paths == ["eager", "eager", "graph-capture", "graph-replay"], paths
    assert len(model._prefill_graphs) == 1 and next(iter(model._prefill_graphs.values()))["graph"] is not None, "the instruct prefill must have been captured"
    e, g1, g2, g3 = res[False][0], res[True][0], res[True][1], res[True][2]
    for gx in (g2, g3):
        assert torch.equal(g1[0], gx[0]) and g1[1:3] == gx[1:3] and torch.equal(g1[3], gx[3]), "first sighting / capture / replay differ"
    assert g1[1:3] == e[1:3], (g1[1:3], e[1:3])
    if dtype == torch.float32:
        assert torch.equal(g1[0], e[0])
        assert float((g1[3] - e[3]).abs().max()) < 1e-3
        ref, _ = Oracle(cfg, sd, dtype, clip=clip).greedy(ids.cpu(), images=images.cpu(), max_new_tokens=8, eos_token_id=None)
        assert g1[0].tolist() == ref.tolist()


def test_checkpoint_roundtrip_through_load_pretrained_model(tmp_path):
    """dynamic_llava_builder.load_pretrained_model surface: config.json + safetensors with the reference's key names."""
    from dynamic_llava_amd.builder import load_pretrained_model, save_pretrained

    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float16)
    save_pretrained(model, str(tmp_path))
    with pytest.raises(Exception):  # no tokenizer files in the directory: an error, as in the reference (BLD:45-49) -- not a silent None
        load_pretrained_model(str(tmp_path), None, "dynamic-llava-tiny")
    with pytest.warns(RuntimeWarning, match="no tokenizer"):
        tok, m2, proc, ctx = load_pretrained_model(str(tmp_path), None, "dynamic-llava-tiny", require_tokenizer=False)  # fp16 default like BLD:62
    assert tok is None and m2.dtype == torch.float16 and ctx == 2048
    ids = fx.make_prompt(cfg, 5, 7)[None].cuda()
    images = fx.make_images(cfg, 1).half().cuda()
    a = model.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    b = m2.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    assert torch.equal(a, b)
    with pytest.raises(NotImplementedError):
        load_pretrained_model(str(tmp_path), None, "x", load_4bit=True)
    with pytest.raises(NotImplementedError):
        load_pretrained_model(str(tmp_path), "base", "x")


def test_vision_tower_hub_id_resolves_from_the_local_hf_cache(tmp_path):
    """BLD:237-242 / clip_encoder.py:22-38 load `config.mm_vision_tower` BY NAME; released checkpoints name a hub id ("openai/clip-vit-large-patch14-336").
    No network here: the id must resolve from the local Hugging Face cache, and a missing entry must say so instead of pretending the id is a path."""
    import json
    import os
    import shutil

    from dynamic_llava_amd.builder import load_pretrained_model, save_pretrained

    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float16)
    ckpt, cache = tmp_path / "ckpt", tmp_path / "hf_cache"
    save_pretrained(model, str(ckpt))
    # the checkpoint names a hub id and carries NO tower tensors (like a released LLaVA checkpoint)
    from safetensors.torch import load_file, save_file

    tensors = {k: v for k, v in load_file(str(ckpt / "model.safetensors")).items() if "vision_tower" not in k}
    save_file(tensors, str(ckpt / "model.safetensors"))
    cj = json.load(open(ckpt / "config.json"))
    cj["mm_vision_tower"] = "openai/clip-tiny-for-tests"
    json.dump(cj, open(ckpt / "config.json", "w"))
    with pytest.raises(FileNotFoundError, match="local Hugging Face cache"):
        load_pretrained_model(str(ckpt), None, "x", require_tokenizer=False, cache_dir=str(cache))
    # put the tower into the cache the way the hub client lays it out: models--org--name/{refs/main, snapshots/<rev>/...}
    from transformers import CLIPImageProcessor

    rev = "0" * 40
    snap = cache / "models--openai--clip-tiny-for-tests" / "snapshots" / rev
    clip.save_pretrained(str(snap))
    CLIPImageProcessor(size={"shortest_edge": cfg.clip["image_size"]}, crop_size={"height": cfg.clip["image_size"], "width": cfg.clip["image_size"]}).save_pretrained(str(snap))
    os.makedirs(cache / "models--openai--clip-tiny-for-tests" / "refs", exist_ok=True)
    (cache / "models--openai--clip-tiny-for-tests" / "refs" / "main").write_text(rev)
    with pytest.warns(RuntimeWarning):
        tok, m2, proc, ctx = load_pretrained_model(str(ckpt), None, "x", require_tokenizer=False, cache_dir=str(cache))
    assert proc is not None, "the image processor comes from the tower's entry, as in clip_encoder.py:33-35"
    ids = fx.make_prompt(cfg, 5, 7)[None].cuda()
    images = fx.make_images(cfg, 1).half().cuda()
    assert torch.equal(model.generate(ids, images=images, max_new_tokens=6, eos_token_id=None), m2.generate(ids, images=images, max_new_tokens=6, eos_token_id=None))
    shutil.rmtree(cache)


def test_text_only_and_edge_prompts_vs_oracle():
    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float32)
    o = Oracle(cfg, sd, torch.float32, clip=clip)
    # (a) no image at all: plain language model (ARCH:180-188 early-out), generate + forward
    ids = fx.make_prompt(cfg, 4, 9)[None]
    ids = ids[ids != -200][None]
    ref, _ = o.greedy(ids, max_new_tokens=6, eos_token_id=None)
    out = model.generate(ids.cuda(), max_new_tokens=6, eos_token_id=None)
    assert out.cpu().tolist() == ref.tolist()
    l_ref, _ = o.forward(ids)
    l = model(ids.cuda()).logits
    assert l.shape == l_ref.shape and float((l.cpu() - l_ref).abs().max()) < 1e-3
    # (b) image is the last prompt token (empty instruct span) and (c) image first (empty system span)
    images = fx.make_images(cfg, 1)
    for prompt in (torch.tensor([[1, 7, 9, -200]]), torch.tensor([[-200, 5, 6, 7]])):
        ref, _ = o.greedy(prompt, images=images, max_new_tokens=5, eos_token_id=None)
        out = model.generate(prompt.cuda(), images=images.cuda(), max_new_tokens=5, eos_token_id=None)
        assert out.cpu().tolist() == ref.tolist(), prompt.tolist()
    # (d) EOS on the very first token and max_new_tokens=1 (the prefill-latency bench, BIMG:134-147)
    first = int(ref[0, 0])
    out = model.generate(prompt.cuda(), images=images.cuda(), max_new_tokens=5, eos_token_id=first)
    assert out.cpu().tolist() == [[first]]


@pytest.mark.parametrize("name", sorted(n for n in CASES if CASES[n].get("nocache")))
def test_nocache_decode_vs_reference_golden(name, golden_dir):
    """`model(total_input_ids, images=..., use_cache=False)` (DML:2393-2504), whole sequence re-run per step,
    answer tokens compacted by top-k of the raw keep logit, incl. the reference's first-call duplicated-last-token quirk."""
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    model.debug_records = {}
    total = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, total.shape[0], seed=0).to(dtype).cuda()
    forced = torch.from_numpy(g["forced"]).cuda()
    for j in range(g["step_logits"].shape[0]):
        out = model(total, images=images, use_cache=False)
        assert out.past_key_values is None and out.logits.shape[1] == g["logits_len"][j], (j, out.logits.shape)
        pos = model.debug_records["position_ids"].cpu().numpy().reshape(total.shape[0], -1)
        np.testing.assert_array_equal(pos, g[f"position_ids_{j}"], err_msg=f"step {j}")
        assert np.abs(out.logits[:, -1].cpu().numpy() - g["step_logits"][j]).max() < 1e-3, f"step {j}"
        total = torch.cat([total, forced[j][:, None]], dim=1)


@pytest.mark.parametrize("name", sorted(n for n in CASES if CASES[n].get("rounds")))
def test_multiround_chunks_on_cache_vs_reference_golden(name, golden_dir):
    """multi-token chunks on a non-empty cache (new-instruct round DML:2506-2521 / chunked prefill),
    interleaved with decode steps; logits, per-token store decisions and KV lengths against the reference."""
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    model.debug_records = {}
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    pkv = None
    for j in range(int(g["n_calls"])):
        ids = torch.from_numpy(g[f"call_ids_{j}"]).cuda()
        out = model(ids, images=images if j == 0 else None, past_key_values=pkv)
        pkv = out.past_key_values
        assert np.abs(out.logits[:, -1].cpu().numpy() - g["step_logits"][j]).max() < 1e-3, f"call {j}"
        np.testing.assert_array_equal(pkv[1][0].numpy(), g["len_first"][j], err_msg=f"call {j}")
        np.testing.assert_array_equal(pkv[1][-1].numpy(), g["len_last"][j], err_msg=f"call {j}")
        assert pkv[0][0][0].shape[-2] == g["kv_len_first"][j] and pkv[0][-1][0].shape[-2] == g["kv_len_last"][j]
        if ids.shape[1] > 1 and j > 0 and g[f"decision_{j}"].size:
            np.testing.assert_array_equal(model.debug_records["text_decision"].cpu().numpy(), g[f"decision_{j}"])


@pytest.mark.parametrize("name", ["tiny_fp32_multiround", "tiny_fp32_chunked"])
def test_generate_continues_on_a_returned_cache_vs_oracle(name):
    """generate(new_turn_ids, past_key_values=cache): the second turn of a dialogue on the cache the first generate() returned (the new chunk
    through the chunk-on-cache path DML:2506-2521, then greedy steps) against the oracle driving the reference's own forward() loop."""
    c, dtype, cfg, sd, clip = _golden_setup(name)
    model = _build(cfg, sd, clip, dtype)
    ids = fx.make_prompt(cfg, 5, 9, seed=0)[None]
    images = fx.make_images(cfg, 1, seed=0).to(dtype)
    turn2 = torch.randint(3, cfg.vocab_size, (1, 7), generator=torch.Generator().manual_seed(12))
    r1 = model.generate(ids.cuda(), images=images.cuda(), max_new_tokens=4, eos_token_id=None, return_dict_in_generate=True)
    out2 = model.generate(turn2.cuda(), past_key_values=r1["past_key_values"], max_new_tokens=5, eos_token_id=None)
    o = Oracle(cfg, sd, dtype, clip=clip)
    t1, pkv = o.greedy(ids, images=images, max_new_tokens=4, eos_token_id=None)
    assert r1["sequences"].cpu().tolist() == t1.tolist()
    with torch.no_grad():
        logits, pkv = o.forward(turn2, past_key_values=pkv)
        ref = []
        for step in range(5):
            nxt = logits[:, -1].argmax(-1)
            ref.append(int(nxt[0]))
            if step < 4:
                logits, pkv = o.forward(nxt[:, None], past_key_values=pkv)
    assert out2.cpu().tolist() == [ref]
    lens = model.last_cache[1]
    assert int(lens[0][0]) == int(pkv[1][0][0]) and int(lens[-1][0]) == int(pkv[1][-1][0])
    with pytest.raises(NotImplementedError):
        model.generate(turn2.cuda(), images=images.cuda(), past_key_values=model.last_cache, max_new_tokens=2)
    # the HF convention (the FULL dialogue's ids + the cache, DML:2835-2848) must be refused, not silently appended a second time
    full = torch.cat([ids, r1["sequences"].cpu(), turn2], dim=1)
    with pytest.raises(ValueError, match="ONLY the new turn"):
        model.generate(full.cuda(), past_key_values=model.last_cache, max_new_tokens=2)
    # ... and a max_length-style limit counts what the cache already holds
    seen = model.last_cache.seen_tokens
    out3 = model.generate(turn2.cuda(), past_key_values=model.last_cache, max_length=seen + turn2.shape[1] + 3, eos_token_id=None)
    assert out3.shape[1] == 3


def test_generate_with_a_set_of_eos_ids(golden_dir):
    """HF accepts a list of eos_token_ids: the greedy device path compares up to three ids (dl_decode_advance), bans all of them while
    fewer than min_new_tokens tokens exist, and stops / trims at the first hit of any."""
    name = "tiny_fp32_b1_greedy"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    gold = [int(t) for t in g["ids"][:, 0]]
    n = len(gold)
    never = max(gold) + 1 if max(gold) + 1 < cfg.vocab_size else 0
    assert never not in gold
    for graph in (False, True):
        model.use_hip_graph = graph
        out = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=[never, gold[3], gold[5]])
        first = min(gold.index(gold[3]), gold.index(gold[5]))
        assert out.cpu().tolist()[0] == gold[: first + 1], graph
        # min_new_tokens bans the whole set: the run must go past position `first`
        out = model.generate(ids, images=images, max_new_tokens=n, min_new_tokens=first + 2, eos_token_id=[gold[3], gold[5]])
        assert out.shape[1] > first + 1 and gold[3] not in out.cpu().tolist()[0][: first + 2] and gold[5] not in out.cpu().tolist()[0][: first + 2]
    # more than three EOS ids (round 4): the plain forward() loop takes over -- same tokens, stop at the first id of the set
    others = [t for t in range(cfg.vocab_size) if t not in gold][:3]
    out = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=others + [gold[4], never])
    assert out.cpu().tolist()[0] == gold[: gold.index(gold[4]) + 1]
    out = model.generate(ids, images=images, max_new_tokens=n, eos_token_id=others + [never])
    assert out.cpu().tolist()[0] == gold


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_clip_tower_packed_path_matches_eager_module(dtype):
    """the CLIP ViT-L/14-336 tower on the packed HIP path (dl_layernorm / dl_add_layernorm / dl_attn_prefill /
    dl_quick_gelu around the library GEMMs) against the HF module run eagerly, full size (24 layers, 577 tokens, 2 images).
    fp32: equal to 1e-4.  16-bit: in the same noise class as the eager module itself, both judged against an fp32 run."""
    from dynamic_llava_amd.config import DynamicLlavaConfig
    from dynamic_llava_amd.model import CLIPVisionTower

    cfg = DynamicLlavaConfig.from_namespace(fx.llava7b_config(num_hidden_layers=1))
    torch.manual_seed(5)
    tower = CLIPVisionTower(cfg)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for n, p in tower.named_parameters():  # non-trivial LayerNorm affine / biases so that every term is exercised
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    truth_tower = copy.deepcopy(tower).to("cuda", torch.float32)
    imgs = torch.randn(2, 3, 336, 336, generator=g)
    truth = truth_tower.forward_eager(imgs.cuda()).float()
    t = tower.to("cuda", dtype).pack()
    x = imgs.cuda().to(dtype)
    eager = t.forward_eager(x).float()
    default_max = t.tiles_max_batch
    t.tiles_max_batch = 0  # the library-GEMM path (what three or more images, and fp32, run)
    hip = t(x).float()
    t.tiles_max_batch = default_max
    assert hip.shape == (2, 576, 1024) and torch.isfinite(hip).all()
    if dtype == torch.float32:
        assert float((hip - truth).abs().max()) < 1e-4 * max(1.0, float(truth.abs().max()))
        assert float((t(x).float() - truth).abs().max()) < 1e-4 * max(1.0, float(truth.abs().max()))  # (fp32 never takes the tiled path)
        return
    e_ref, e_hip = (eager - truth), (hip - truth)
    assert float(e_hip.abs().max()) <= 2.0 * float(e_ref.abs().max()) + 1e-3, (float(e_hip.abs().max()), float(e_ref.abs().max()))
    assert float(e_hip.pow(2).mean().sqrt()) <= 1.5 * float(e_ref.pow(2).mean().sqrt()) + 1e-4
    # round 6: up to `tiles_max_batch` (2) images run every projection on dl_linear_tiles (operand-order weight copies, QuickGELU in fc1's epilogue, out_proj /
    # fc2 as fp32 k-range partial sums added by the residual-add + LayerNorm launch; 80-row tiles at one image, 160-row tiles at two).  Same bounds, against
    # the same fp32 truth.
    assert t._tiles[0] is not None and t.tiles_max_batch == 2
    hip1 = t(x[:1]).float()
    e_hip1 = hip1 - truth[:1]
    assert float(e_hip1.abs().max()) <= 2.0 * float(e_ref[:1].abs().max()) + 1e-3, (float(e_hip1.abs().max()), float(e_ref[:1].abs().max()))
    assert float(e_hip1.pow(2).mean().sqrt()) <= 1.5 * float(e_ref[:1].pow(2).mean().sqrt()) + 1e-4
    hip2 = t(x).float()  # (the attention launch picks another kernel at this grid size, so same bounds rather than the one-image run's bits -- the GEMM
    e_hip2 = hip2 - truth  # itself is row-position invariant: tests/test_linear_tiles_gpu.py)
    assert float(e_hip2.abs().max()) <= 2.0 * float(e_ref.abs().max()) + 1e-3 and float(e_hip2.pow(2).mean().sqrt()) <= 1.5 * float(e_ref.pow(2).mean().sqrt()) + 1e-4
    x3 = torch.cat([x, x[:1]])  # three images: library path; rows of the first two images must not depend on the batch they rode in beyond kernel choice
    hip3 = t(x3).float()
    assert float((hip3[:2] - truth).abs().max()) <= 2.0 * float(e_ref.abs().max()) + 1e-3
    # unused last layer is really skipped, CLS token dropped
    t.select_feature = "cls_patch"
    assert t(x).shape == (2, 577, 1024)


def test_sparse_layer_beyond_depth_is_dense():
    """A model shallower than `sparse_layer` never reaches the sparsification point (DML:1826 sits inside the layer loop): nothing is
    dropped, logits cover all N positions and equal the dense oracle."""
    dtype = torch.float32
    cfg = fx.tiny_config()
    cfg.num_hidden_layers = 2  # sparse_layer = 2 is never reached
    sd = fx.make_state_dict(cfg, seed=3, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1, dtype=dtype)
    model = _build(cfg, sd, clip, dtype)
    ids = fx.make_prompt(cfg, 5, 7, seed=1)[None]
    imgs = fx.make_images(cfg, 1, seed=2)
    out = model(ids.cuda(), images=imgs.cuda().to(dtype))
    n = ids.shape[1] - 1 + fx.n_image_tokens(cfg)
    assert out.logits.shape[1] == n and out.past_key_values[0][-1][0].shape[-2] == n
    o = Oracle(cfg, sd, dtype, clip=clip)
    with torch.no_grad():
        l_ref, _ = o.forward(ids, images=imgs)
    assert l_ref.shape == out.logits.shape
    assert float((out.logits.cpu() - l_ref).abs().max()) < 1e-3


@pytest.mark.parametrize("sparse_layer", [1, 3])  # 0 is not a valid reference setting: DML:1026-1043 reads the cache of layer sparse_layer-1
def test_other_sparse_layers_vs_oracle(sparse_layer):
    """`sparse_config["sparse_layer"]` is a runtime toggle of the reference (BIMG:63-65 flips such keys between runs): the vision block,
    the text-predictor decision and the two KV-length groups follow it.  Prefill + 6 teacher-forced decode steps against the oracle."""
    dtype = torch.float32
    cfg = fx.tiny_config()
    cfg.sparse_config = dict(cfg.sparse_config, sparse_layer=sparse_layer)
    sd = fx.make_state_dict(cfg, seed=4, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1, dtype=dtype)
    model = _build(cfg, sd, clip, dtype)
    assert model.config.sparse_config["sparse_layer"] == sparse_layer
    ids = fx.make_prompt(cfg, 5, 9, seed=2)[None]
    imgs = fx.make_images(cfg, 1, seed=3)
    forced = fx.make_forced_tokens(cfg, 6, 1, seed=5)
    o = Oracle(cfg, sd, dtype, clip=clip)
    model.debug_records = {}
    out = model(ids.cuda(), images=imgs.cuda().to(dtype))
    with torch.no_grad():
        l_ref, p_ref = o.forward(ids, images=imgs)
    assert out.logits.shape == l_ref.shape and float((out.logits.cpu() - l_ref).abs().max()) < 1e-3
    assert torch.equal(model.debug_records["keep_index"].cpu(), o.records["keep_index"])
    pkv = out.past_key_values
    for j in range(6):
        out = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = out.past_key_values
        with torch.no_grad():
            l_ref, p_ref = o.forward(forced[j][:, None], past_key_values=p_ref)
        assert float((out.logits.cpu() - l_ref).abs().max()) < 1e-3, f"step {j}"
        assert int(model.debug_records["text_decision"][0]) == int(o.records["text_decision"][0, 0]), f"step {j}"
        for layer in range(cfg.num_hidden_layers):
            assert pkv[0][layer][0].shape[-2] == p_ref[0][layer][0].shape[-2], (j, layer)
    model.debug_records = None


def test_generate_sampling_path():
    """generate(do_sample=True, temperature, top_p, top_k) -- what the harness passes when temperature > 0 (model_vqa_loader.py:162-175).
    Draws cannot match HF's RNG; the warpers' semantics can be pinned: top_k=1 or a tiny top_p keep only the argmax, i.e. greedy."""
    dtype = torch.float32
    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1, dtype=dtype)
    model = _build(cfg, sd, clip, dtype)
    ids = torch.stack([fx.make_prompt(cfg, 5, 9, seed=2), fx.make_prompt(cfg, 5, 9, seed=3)])
    imgs = fx.make_images(cfg, 2, seed=3).cuda()
    greedy = model.generate(ids.cuda(), images=imgs, max_new_tokens=8, eos_token_id=None)
    a = model.generate(ids.cuda(), images=imgs, max_new_tokens=8, eos_token_id=None, do_sample=True, temperature=0.7, top_k=1)
    b = model.generate(ids.cuda(), images=imgs, max_new_tokens=8, eos_token_id=None, do_sample=True, temperature=1.3, top_p=1e-6)
    assert torch.equal(a, greedy) and torch.equal(b, greedy)
    g1 = torch.Generator(device="cuda").manual_seed(7)
    g2 = torch.Generator(device="cuda").manual_seed(7)
    c1 = model.generate(ids.cuda(), images=imgs, max_new_tokens=8, eos_token_id=None, do_sample=True, temperature=5.0, top_p=0.95, generator=g1)
    c2 = model.generate(ids.cuda(), images=imgs, max_new_tokens=8, eos_token_id=None, do_sample=True, temperature=5.0, top_p=0.95, generator=g2)
    assert c1.shape == (2, 8) and torch.equal(c1, c2), "same generator state -> same draws"
    assert not torch.equal(c1, greedy), "a hot temperature on near-uniform random-init logits must leave the greedy path"
    with pytest.raises(NotImplementedError):
        model.generate(ids.cuda(), images=imgs, max_new_tokens=2, num_beams=2)


def test_generate_then_forward_on_returned_cache_vs_oracle():
    """the cache generate() hands back must be consistent on the HOST side too (un-evicted length mirrors), so that a
    follow-up forward(past_key_values=cache) -- the multi-round driver of BLTM:326-337 -- attends to the whole cache, a returned
    cache is not overwritten by the next generate(), and HF kwargs that are not built fail loudly."""
    c, dtype, cfg, sd, clip = _golden_setup("tiny_fp32_b1_gain50")
    model = _build(cfg, sd, clip, dtype)
    ids = fx.make_prompt(cfg, 5, 7)[None]
    images = fx.make_images(cfg, 1, seed=0)
    n = 7
    res = model.generate(ids.cuda(), images=images.cuda(), max_new_tokens=n, eos_token_id=None, return_dict_in_generate=True, output_scores=True)
    seq, cache = res["sequences"], res["past_key_values"]
    n_prompt = ids.shape[1] - 1 + fx.n_image_tokens(cfg)
    assert cache.full_len_host == [n_prompt + n - 1] and cache.get_seq_length(0) == n_prompt + n - 1
    assert cache[0][0][0].shape[-2] == n_prompt + n - 1 == int(cache[1][0][0])
    assert len(res["scores"]) == n and all(s.shape == (1, cfg.vocab_size) for s in res["scores"])
    assert [int(s.argmax(-1)) for s in res["scores"]] == seq[0].tolist()
    o = Oracle(cfg, sd, dtype, clip=clip)
    with torch.no_grad():
        ref, pkv = o.greedy(ids, images=images, max_new_tokens=n, eos_token_id=None)
        assert seq.cpu().tolist() == ref.tolist()
        l_ref, pkv = o.forward(ref[:, -1:], past_key_values=pkv)
    # another generate() in between must NOT touch the returned cache (it was detached from the pool)
    k_before = cache.k[3][:, :, : n_prompt, :].clone()
    model.generate(fx.make_prompt(cfg, 4, 9, seed=3)[None].cuda(), images=images.cuda(), max_new_tokens=3, eos_token_id=None)
    assert torch.equal(cache.k[3][:, :, : n_prompt, :], k_before)
    out = model(seq[:, -1:], past_key_values=cache)
    assert float((out.logits[0, -1].cpu() - l_ref[0, -1]).abs().max()) < 1e-3
    np.testing.assert_array_equal(out.past_key_values[1][-1].numpy(), pkv[1][-1].numpy())
    np.testing.assert_array_equal(out.past_key_values[1][0].numpy(), pkv[1][0].numpy())
    # min_new_tokens: EOS cannot be emitted before that many tokens exist (HF MinNewTokensLengthLogitsProcessor), then it stops
    eos = int(ref[0, 0])
    a = model.generate(ids.cuda(), images=images.cuda(), max_new_tokens=4, eos_token_id=eos)
    assert a.cpu().tolist() == [[eos]]
    b = model.generate(ids.cuda(), images=images.cuda(), max_new_tokens=4, min_new_tokens=2, eos_token_id=eos)
    assert b.shape[1] >= 2 and eos not in b[0, :2].tolist()
    first_logits = model.last_prefill_logits[0].clone()
    first_logits[eos] = float("-inf")
    assert int(b[0, 0]) == int(first_logits.argmax())
    for bad in (dict(past_key_values=cache), dict(num_beams=2), dict(inputs_embeds=torch.zeros(1))):
        with pytest.raises(NotImplementedError):
            model.generate(ids.cuda(), images=images.cuda(), max_new_tokens=2, **bad)
    with pytest.raises(NotImplementedError):
        model.generate(ids.cuda(), images=images.cuda(), max_new_tokens=2, do_sample=True, num_beams=3)


def test_harness_long_text_kv_length_curve_matches_reference_golden(golden_dir):
    """SURVEY A9 / BLTM:326-351: tools/harness_long_text_mem.py drives the reference's own loop and record format; the KV-length curve
    (`past_key_values[0][-1][0].shape[-2]` after every call) must equal the one the REFERENCE produced (golden), and the dense layers'
    length must grow by one per token."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("harness_long_text_mem", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "harness_long_text_mem.py"))
    h = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h)
    name = "tiny_fp32_b1_gain50"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    forced = torch.from_numpy(g["forced"]).cuda()  # [steps + 1, B]: the label ids the reference was teacher-forced with
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    rec = h.run(model, cfg, ids, forced.t().contiguous(), images, verbose=False)
    n = forced.shape[0]
    assert rec["kv_cache_length"] == [int(x) for x in g["kv_len_last"][:n]]
    n_prompt = ids.shape[1] - 1 + fx.n_image_tokens(cfg)
    # total_token_length follows the script's own accounting (images.shape[-2] * images.shape[-1] // 14 // 14 patches + text)
    assert rec["total_token_length"][0] == n_prompt and rec["total_token_length"][-1] == n_prompt + n - 1
    assert len(rec["max_memory"]) == n and all(m > 0 for m in rec["max_memory"])


def test_harness_model_memory_reports_operand_copies_separately():
    """VERDICT r5 weak #8 / BIMG:59-67, 153-156: the harness counterpart prints `model memory` like the reference script; the operand-order weight copies this
    engine keeps beside the parameters (dl_linear_packed, dl_linear_tiles) are reported separately, and the figure WITHOUT them equals the parameter bytes to
    1 % -- with the copies built, and with --no-operand-copies (where nothing but the tower's fused q|k|v is left to subtract)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("harness_bimg", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "harness_image_time_and_mem.py"))
    h = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h)
    recs = {}
    env_before = {k: os.environ.get(k) for k in ("DL_PACKED_GEMM", "DL_CLIP_TILES")}
    for flag in ((), ("--no-operand-copies",)):
        torch.cuda.empty_cache()
        recs[flag] = h.main(["--model", "7b", "--layers", "3", "--reps", "2", *flag])
    assert {k: os.environ.get(k) for k in env_before} == env_before, "the harness must leave the process environment as it found it"
    with_c, without = recs[()], recs[("--no-operand-copies",)]
    oc = with_c["operand_copy_bytes"]
    assert with_c["operand_copies_built"] and oc["decoder_operand_order"] == 3 * (12288 + 22016 + 11008) * 4096 * 2  # q|k|v + gate|up + down, three layers, fp16
    assert oc["clip_operand_order"] == 24 * 12 * 1024 * 1024 * 2 and oc["projector_operand_order"] == (1024 * 4096 + 4096 * 4096) * 2
    no = without["operand_copy_bytes"]
    assert not without["operand_copies_built"] and no["decoder_operand_order"] == 0 and no["clip_operand_order"] == 0 and no["projector_operand_order"] == 0
    for r in (with_c, without):
        assert abs(r["model_memory_without_operand_copies"] - r["parameter_bytes"]) <= 0.01 * r["parameter_bytes"], (r["model_memory_without_operand_copies"], r["parameter_bytes"])
    assert with_c["parameter_bytes"] == without["parameter_bytes"]
    assert with_c["kv_cache_length_last_layer"] == without["kv_cache_length_last_layer"] == 2 + 115


def test_weights_replaced_after_finalize_reach_every_path():
    """the operand-order copies are detached from the parameters; a load_state_dict() after finalize() must not leave the packed prefill
    (dl_linear_packed / dl_linear_tiles) on the old weights while the GEMV path uses the new ones.  Two models with different seeds; loading B's state dict
    into A must make A generate exactly what B generates (prefill logits bit-equal), through the packed paths."""
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig(num_hidden_layers=3)
    a = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=1, predictor_gain=50.0)
    b = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=2, predictor_gain=50.0)
    g = torch.Generator().manual_seed(3)
    images = torch.randn((1, 3, 336, 336), generator=g).to(torch.bfloat16).cuda()
    ids = fx.make_prompt(cfg, 35, 20, seed=0)[None].cuda()
    out_a0 = a.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    la0 = a.last_prefill_logits.clone()
    out_b = b.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    lb = b.last_prefill_logits.clone()
    assert not torch.equal(la0, lb)
    assert a.model.layers[2].wp_qkv is not None and a.get_vision_tower()._tiles[0] is not None, "the packed paths must be the ones under test"
    a.load_state_dict(b.state_dict())
    out_a1 = a.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    assert torch.equal(a.last_prefill_logits, lb) and torch.equal(out_a1, out_b)
    # an in-place edit of ONE projection is noticed too
    with torch.no_grad():
        a.model.layers[2].mlp.down_proj.weight.mul_(0.5)
        b.model.layers[2].mlp.down_proj.weight.mul_(0.5)
    out_a2 = a.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    la2 = a.last_prefill_logits.clone()
    out_b2 = b.generate(ids, images=images, max_new_tokens=6, eos_token_id=None)
    assert torch.equal(la2, b.last_prefill_logits) and torch.equal(out_a2, out_b2) and not torch.equal(la2, lb)


def test_harness_multi_round_ppl_drives_the_reference_loop(golden_dir):
    """SURVEY 8f row N2 / model_lvis_multi_round_for_ppl.py:108-220: tools/harness_multi_round_ppl.py runs the reference's multi-round perplexity loop
    (prefill, label tokens teacher-forced, every later round a multi-token chunk on the returned cache, a round's last label never fed) -- on the golden
    multi-round case its sequence of model calls IS the golden's: every call's last-token logits within 1e-3 of the REFERENCE's, the last layer's KV
    length after every call equal, and the per-round perplexities equal to the ones computed from the reference's logits."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("harness_mrp", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "harness_multi_round_ppl.py"))
    h = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h)
    name = "tiny_fp32_multiround"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    # the golden's call list, cut into rounds: a multi-token call opens a round, the single-token calls after it are that round's fed labels; the round's
    # final label (a target only) is drawn here
    calls = [torch.from_numpy(g[f"call_ids_{j}"]) for j in range(int(g["n_calls"]))]
    gl = torch.Generator().manual_seed(77)
    rounds, call_round = [], []
    for j, ids in enumerate(calls):
        if ids.shape[1] > 1:
            rounds.append([ids, []])
        else:
            rounds[-1][1].append(int(ids[0, 0]))
        call_round.append(len(rounds) - 1)
    for r in rounds:
        r[1].append(int(torch.randint(3, cfg.vocab_size, (1,), generator=gl)))
    seen = []
    rec = h.run(model, [(p_.cuda(), torch.tensor(l_).cuda()) for p_, l_ in rounds], images, patch=cfg.clip["patch_size"], on_call=lambda j, out: seen.append(out.logits[:, -1].float().cpu().numpy()))
    assert rec["calls"] == len(calls) == len(seen)
    for j in range(len(calls)):
        assert np.abs(seen[j] - g["step_logits"][j]).max() < 1e-3, f"call {j}"
    assert rec["kv_len_last_layer"] == [int(x) for x in g["kv_len_last"]]
    for r, (p_, l_) in enumerate(rounds):
        lg = torch.from_numpy(np.concatenate([g["step_logits"][j] for j in range(len(calls)) if call_round[j] == r]))
        ppl_ref = float(torch.exp(F.cross_entropy(lg, torch.tensor(l_))))
        assert abs(rec["round_ppl"][r] - ppl_ref) <= 2e-3 * ppl_ref, (r, rec["round_ppl"][r], ppl_ref)
    assert rec["instruct_token_length"] == sum(p_.shape[1] for p_, _ in rounds) and rec["output_token_length"] == sum(len(l_) - 1 for _, l_ in rounds)
    assert rec["prefill_cache_length"] == int(g["kv_len_last"][0]) + sum(p_.shape[1] for p_, _ in rounds[1:])
    assert rec["output_cache_length"] == int(g["kv_len_last"][-1]) - rec["prefill_cache_length"]


def test_harness_long_text_time_no_cache_drives_the_reference_loop(golden_dir):
    """SURVEY A9 / BLTN:316-357: tools/harness_long_text_time_no_cache.py runs the reference's no-KV-cache timing loop
    (`model(total_input_ids, images=, past_key_values=None, use_cache=False)` per label token, teacher forcing) with its record format,
    on the golden no-cache case: the loop must reach the same final logits as the golden's last step."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("harness_bltn", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "harness_long_text_time_no_cache.py"))
    h = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h)
    name = "tiny_fp32_nocache"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = _build(cfg, sd, clip, dtype)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    forced = torch.from_numpy(g["forced"]).cuda()
    n = g["step_logits"].shape[0]
    images = fx.make_images(cfg, ids.shape[0], seed=0).to(dtype).cuda()
    rec = h.run(model, ids, forced[:n].t().contiguous(), images)
    assert rec["output_token_length"] == list(range(1, n + 1)) and len(rec["step_time_ms"]) == n and all(t > 0 for t in rec["step_time_ms"])
    assert all(m > 0 for m in rec["max_memory"]) and abs(rec["total_time_ms"] - sum(rec["step_time_ms"])) < 1e-6
    # the stateful answer_indice left behind by the loop is the reference's: one more direct call reproduces the golden's next step
    total = torch.cat([ids, forced[: n - 1].t()], dim=1)
    model2 = _build(cfg, sd, clip, dtype)
    for j in range(n):
        out = model2(torch.cat([ids, forced[:j].t()], dim=1), images=images, use_cache=False)
    assert total.shape[1] == ids.shape[1] + n - 1
    assert np.abs(out.logits[:, -1].cpu().numpy() - g["step_logits"][n - 1]).max() < 1e-3


@pytest.mark.parametrize("side,tmax", [("right", None), ("left", None), ("right", 42), ("left", 42)])
def test_prepare_inputs_padding_side_and_truncation_vs_oracle(side, tmax):
    """ARCH:493-579 (the oracle is pinned to the live reference for exactly these cases in tests/test_oracle_vs_reference.py): ragged
    batch -> padded embeddings / mask / position ids / shifted-clamped segment dicts, then forward(inputs_embeds=...) on them."""
    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=5, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=6)
    model = _build(cfg, sd, clip, torch.float32)
    prompts = [fx.make_prompt(cfg, 4, 9, seed=1), fx.make_prompt(cfg, 2, 5, seed=2)]
    W = max(p.shape[0] for p in prompts)
    ids = torch.zeros(2, W, dtype=torch.long)
    am = torch.zeros(2, W, dtype=torch.bool)
    for b, p in enumerate(prompts):
        ids[b, : p.shape[0]] = p
        am[b, : p.shape[0]] = True
    images = fx.make_images(cfg, 2, seed=3)
    pos = torch.arange(W)[None].repeat(2, 1)
    cfg.tokenizer_padding_side = side
    cfg.tokenizer_model_max_length = tmax
    model.config.tokenizer_padding_side = side
    model.config.tokenizer_model_max_length = tmax
    o = Oracle(cfg, sd, torch.float32, clip=clip)
    with torch.no_grad():
        (_, o_pos, o_am, _, o_emb, _), (o_idx,) = o.prepare_inputs_labels_for_multimodal(ids, pos, am, None, None, images)
    (_, h_pos, h_am, _, h_emb, _), (h_idx,) = model.prepare_inputs_labels_for_multimodal(ids.cuda(), pos.cuda(), am.cuda(), None, None, images.cuda())
    assert torch.equal(h_am.cpu().bool(), o_am.bool()) and torch.equal(h_pos.cpu(), o_pos)
    assert [{k: [int(v[0]), int(v[1])] for k, v in d.items()} for d in h_idx] == [{k: [int(v[0]), int(v[1])] for k, v in d.items()} for d in o_idx]
    assert h_emb.shape == o_emb.shape and float((h_emb.cpu() - o_emb).abs().max()) < 1e-3
    # the padded tensors drive forward() exactly as DLL:68-115 passes them on
    import copy as _copy

    out = model(inputs_embeds=h_emb, attention_mask=h_am, input_embeds_indices=_copy.deepcopy(h_idx))
    lens = out.past_key_values[1][-1].tolist()
    # a batched row equals its own B=1 run (the only well-defined batched semantics: the reference zero-pads B>1 KV, SURVEY finding 2)
    for b in range(2):
        o1 = Oracle(cfg, sd, torch.float32, clip=clip)
        with torch.no_grad():
            l_ref, pkv = o1.forward(prompts[b][None], images=images[b : b + 1])
        assert lens[b] == int(pkv[1][-1][0]), (b, side, tmax)
        got = out.logits[b, lens[b] - 1].cpu()
        assert float((got - l_ref[0, -1]).abs().max()) < 1e-3, (b, side, tmax)


def test_device_prompt_layout_kernel_and_generate_fast_path():
    """dl_prompt_layout == the host restatement of ARCH:330-340 / 418-489 (image position, last "USER:" match, packed
    index lists); generate() with the device layout == generate() with the host layout; rows without an image fall back."""
    from dynamic_llava_amd import hip_ops as ops
    from dynamic_llava_amd.config import IMAGE_TOKEN_INDEX
    from dynamic_llava_amd.model import USER_IDS

    cfg = fx.tiny_config()
    cfg.vocab_size = 30000
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float32)
    n_feat = fx.n_image_tokens(cfg)
    g = torch.Generator().manual_seed(3)
    B, W = 5, 40
    ids = torch.randint(3, cfg.vocab_size, (B, W), generator=g)
    img_pos = [0, 7, 39, 20, 11]
    for b in range(B):
        ids[b, img_pos[b]] = IMAGE_TOKEN_INDEX
    for b, offs in enumerate([[], [3, 20], [], [1], [5, 6, 25]]):  # "USER:" pairs inside the instruct span
        for o_ in offs:
            c = img_pos[b] + 1 + o_
            if c + 1 < W:
                ids[b, c], ids[b, c + 1] = USER_IDS[0], USER_IDS[1]
    out = ops.prompt_layout(ids.cuda().contiguous(), n_feat, IMAGE_TOKEN_INDEX, USER_IDS)
    lay = model._layout(ids.cuda(), None, None, n_feat)
    seg = out["seg"].cpu()
    assert int(out["err"].item()) == 0
    for b in range(B):
        ix = lay["indices"][b]
        assert int(seg[b, 0]) == ix["image"][0] and int(seg[b, 2]) == 1
        assert int(seg[b, 1]) == ix["last_instruct"][0] - ix["instruct"][0], b
    assert out["text_src"].cpu().tolist() == lay["text_src"] and out["text_dst"].cpu().tolist() == lay["text_dst"]
    assert out["img_dst"].cpu().tolist() == lay["img_dst"] and out["img_start"].cpu().tolist() == [ix["image"][0] for ix in lay["indices"]]
    # ... and directly against the ORACLE's segment table (pinned to the reference, ARCH:330-489): image start, last "USER:" offset
    o = Oracle(cfg, sd, torch.float32, clip=clip)
    feats = torch.zeros(B, n_feat, cfg.hidden_size)
    (_, _, _, _, _, _), (o_idx,) = o.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, None, image_features=feats)
    for b in range(B):
        assert int(seg[b, 0]) == o_idx[b]["image"][0], b
        assert int(seg[b, 1]) == o_idx[b]["last_instruct"][0] - o_idx[b]["instruct"][0], (b, seg[b].tolist(), o_idx[b])
        assert int(out["img_start"][b]) == o_idx[b]["image"][0]
    bad = ids.clone()
    bad[2, img_pos[2]] = 5  # no image token in row 2
    assert int(ops.prompt_layout(bad.cuda().contiguous(), n_feat, IMAGE_TOKEN_INDEX, USER_IDS)["err"].item()) == 3
    # generate(): device layout (graph replay on changing image positions) == host layout
    images = fx.make_images(cfg, 2, seed=0)
    p2 = ids[:2, :12].clone()
    p2[0, 4], p2[1, 9] = IMAGE_TOKEN_INDEX, IMAGE_TOKEN_INDEX
    p2[p2 == IMAGE_TOKEN_INDEX] = 7
    for trial, (c0, c1) in enumerate([(4, 9), (0, 11), (6, 2)]):
        q = p2.clone()
        q[0, c0], q[1, c1] = IMAGE_TOKEN_INDEX, IMAGE_TOKEN_INDEX
        model.device_prompt_layout = True
        a = model.generate(q.cuda(), images=images.cuda(), max_new_tokens=5, eos_token_id=None)
        model.device_prompt_layout = False
        b_ = model.generate(q.cuda(), images=images.cuda(), max_new_tokens=5, eos_token_id=None)
        assert torch.equal(a, b_), trial
    # a row without an image: the device layout flags it and the call transparently re-runs on the host layout
    q = p2.clone()
    q[0, 3] = IMAGE_TOKEN_INDEX
    model.device_prompt_layout = True
    a = model.generate(q.cuda(), images=images.cuda(), max_new_tokens=4, eos_token_id=None)
    model.device_prompt_layout = False
    b_ = model.generate(q.cuda(), images=images.cuda(), max_new_tokens=4, eos_token_id=None)
    assert torch.equal(a, b_)


def test_userprompt_golden_layout_kernel_vs_oracle_and_generate(golden_dir):
    """The golden case WITH "USER:" tokens in its prompt (tiny_fp32_userprompt, made by running the reference): (i) dl_prompt_layout's
    "USER:" scan against the oracle's segment table on exactly these ids; (ii) generate() on this case -- the instruct predictor is on, so
    the last-instruct span that the scan found decides which tokens are dropped -- reproduces the golden's first token, its KV lengths
    and the oracle's greedy continuation, with and without a captured graph."""
    from dynamic_llava_amd import hip_ops as ops
    from dynamic_llava_amd.config import IMAGE_TOKEN_INDEX
    from dynamic_llava_amd.model import USER_IDS

    name = "tiny_fp32_userprompt"
    c, dtype, cfg, sd, clip = _golden_setup(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ids = torch.from_numpy(g["input_ids"])
    assert any(int(ids[0, j]) == USER_IDS[0] and int(ids[0, j + 1]) == USER_IDS[1] for j in range(ids.shape[1] - 1)), "the golden prompt must contain USER:"
    n_feat = fx.n_image_tokens(cfg)
    o = Oracle(cfg, sd, dtype, clip=clip)
    (_, _, _, _, _, _), (o_idx,) = o.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, None, image_features=torch.zeros(1, n_feat, cfg.hidden_size))
    out = ops.prompt_layout(ids.cuda().contiguous(), n_feat, IMAGE_TOKEN_INDEX, USER_IDS)
    seg = out["seg"].cpu()
    assert int(out["err"].item()) == 0
    assert int(seg[0, 0]) == o_idx[0]["image"][0] and int(seg[0, 2]) == 1
    assert int(seg[0, 1]) == o_idx[0]["last_instruct"][0] - o_idx[0]["instruct"][0] > 0, (seg[0].tolist(), o_idx[0])
    model = _build(cfg, sd, clip, dtype)
    images = fx.make_images(cfg, 1, seed=0).to(dtype).cuda()
    ref, _ = Oracle(cfg, sd, dtype, clip=clip).greedy(ids, images=images.cpu(), max_new_tokens=8, eos_token_id=None)
    for graph in (False, True):
        model.use_hip_graph = graph
        one = model.generate(ids.cuda(), images=images, max_new_tokens=1, eos_token_id=None)
        assert int(model.last_cache[1][-1][0]) == int(g["len_last"][0][0]) and int(model.last_cache[1][0][0]) == int(g["len_first"][0][0]), graph
        assert abs(model.last_prefill_logits.cpu().numpy()[0] - g["step_logits"][0][0]).max() < 1e-3
        assert int(one[0, 0]) == int(g["step_logits"][0][0].argmax())
        got = model.generate(ids.cuda(), images=images, max_new_tokens=8, eos_token_id=None)
        assert got.cpu().tolist() == ref.tolist(), graph


def test_device_layout_speculation_is_memory_safe_for_multi_image_rows():
    """a row with TWO image tokens on the (speculative) device-layout path used to gather embedding row -200 inside the captured
    graph before the error flag was looked at.  The gathered ids are clamped now; the call must end in the host layout's clean
    NotImplementedError (ARCH:330-332 calls .item() on the image position: exactly one image per row), and a text-only row mixed with an
    image row must still fall back and match the host layout."""
    from dynamic_llava_amd.config import IMAGE_TOKEN_INDEX

    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float32)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, cfg.vocab_size, (2, 12), generator=g)
    ids[0, 4] = IMAGE_TOKEN_INDEX
    ids[1, 2], ids[1, 9] = IMAGE_TOKEN_INDEX, IMAGE_TOKEN_INDEX  # two images in row 1
    images = fx.make_images(cfg, 2, seed=0).cuda()
    model.device_prompt_layout = True
    with pytest.raises(NotImplementedError):
        model.generate(ids.cuda(), images=images, max_new_tokens=3, eos_token_id=None)
    torch.cuda.synchronize()  # no sticky device fault was left behind
    ok = ids.clone()
    ok[1, 2], ok[1, 9] = 5, 6  # row 1 text-only
    a = model.generate(ok.cuda(), images=images, max_new_tokens=3, eos_token_id=None)
    model.device_prompt_layout = False
    b_ = model.generate(ok.cuda(), images=images, max_new_tokens=3, eos_token_id=None)
    assert torch.equal(a, b_)


def test_decode_graph_is_reused_across_prompt_lengths():
    """the captured decode step is keyed by what its launches depend on (slab, split-KV factors), not by the requested
    capacity: prompts of different lengths that share a pooled slab and the same split factors replay ONE graph."""
    from dynamic_llava_amd.config import IMAGE_TOKEN_INDEX

    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float32)
    images = fx.make_images(cfg, 1, seed=0).cuda()
    g = torch.Generator().manual_seed(6)
    outs = {}
    for W in (20, 14, 17, 20):
        ids = torch.randint(3, cfg.vocab_size, (1, W), generator=torch.Generator().manual_seed(W))
        ids[0, 3] = IMAGE_TOKEN_INDEX
        out = model.generate(ids.cuda(), images=images, max_new_tokens=6, eos_token_id=None)
        ref, _ = Oracle(cfg, sd, torch.float32, clip=clip).greedy(ids, images=images.cpu(), max_new_tokens=6, eos_token_id=None)
        assert out.cpu().tolist() == ref.tolist(), W
        outs.setdefault(W, out)
        assert torch.equal(outs[W], out)
    assert len(model._dstate.graphs) == 1, list(model._dstate.graphs)


def test_fused_qkv_attention_launch_vs_the_two_launches_default_config():
    """dl_gemv_qkv_attn (four waves) against dl_gemv + the stand-alone single-split attention as the DEFAULT configuration runs it
    (eight waves at batch 1: another summation order, so bits may differ -- the kernel-level bit-identity test switches that off).  What must
    hold in the default configuration: same greedy tokens, same eviction decisions / KV lengths, prefill logits untouched, on the 7B-width model."""
    from dynamic_llava_amd.builder import build_from_state_dict
    from dynamic_llava_amd.config import DynamicLlavaConfig

    dtype = torch.bfloat16
    cfg = fx.llava7b_config(num_hidden_layers=3)
    cfg.vocab_size = 4096
    sd = fx.make_state_dict(cfg, seed=11, predictor_gain=50.0)
    ids = fx.make_prompt(cfg, 35, 20, seed=0)[None].cuda()
    feats = torch.randn(1, 576, 4096, generator=torch.Generator().manual_seed(3)).to(dtype).cuda()
    runs = {}
    for fuse in (True, False):
        model = build_from_state_dict(DynamicLlavaConfig.from_namespace(cfg), sd, None, dtype=dtype, device="cuda")
        model.fuse_qkv_attn = fuse
        out = model.generate(ids, image_features=feats, max_new_tokens=24, eos_token_id=None)
        model.check_device_errors()
        runs[fuse] = (out.cpu(), [t.clone() for t in model.last_cache[1]], model.last_prefill_logits.float().cpu().clone())
        del model
    assert torch.equal(runs[True][2], runs[False][2]), "the prefill does not depend on the decode launch choice"
    assert torch.equal(runs[True][0], runs[False][0]), "greedy tokens, fused vs two launches"
    assert all(torch.equal(a, b) for a, b in zip(runs[True][1], runs[False][1])), "KV lengths (eviction decisions), fused vs two launches"
    kept = int(runs[True][1][-1][0]) - (35 + 115 + 20)
    print(f"fused == unfused on tokens and KV lengths over 24 tokens ({kept} of 23 decode tokens kept)")


def test_width_buckets_share_prefill_graphs_and_every_width_matches_the_oracle():
    """Round 4 (VERDICT r3 missing #5 / weak #9): the reference's eval loop presents a new prompt width nearly every call (VQAL:123-196).  The
    device-layout prefill is captured per WIDTH BUCKET: launches sized for the bucket, true width as a device scalar, sequences packed at their
    true lengths by dl_prompt_layout.  Every width of a sweep must equal the oracle (tokens, prefill logits 1e-3 in fp32, KV lengths) on its
    first sighting (eager), on the capture and on replays, while the sweep shares a handful of cached prefills."""
    from dynamic_llava_amd.config import IMAGE_TOKEN_INDEX

    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=SD_SEED, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=1)
    model = _build(cfg, sd, clip, torch.float32)
    model.record_timing = True
    assert model.prefill_width_bucket == 16
    o = Oracle(cfg, sd, torch.float32, clip=clip)
    images = [fx.make_images(cfg, 1, seed=s_).cuda() for s_ in range(3)]
    widths = list(range(9, 42))
    order = widths + widths[::-1] + widths[::3]  # every bucket is seen first (eager), captured, and replayed with several true widths
    paths, n_checked = [], 0
    for n_call, W in enumerate(order):
        ids = torch.randint(3, cfg.vocab_size, (1, W), generator=torch.Generator().manual_seed(100 + W))
        ids[0, 2 + W % 5] = IMAGE_TOKEN_INDEX
        img = images[n_call % 3]
        out = model.generate(ids.cuda(), images=img, max_new_tokens=5, eos_token_id=None)
        paths.append(model.last_timing["path"])
        ref, _ = o.greedy(ids, images=img.cpu(), max_new_tokens=5, eos_token_id=None)
        assert out.cpu().tolist() == ref.tolist(), (W, paths[-1])
        l_ref, pkv = o.forward(ids, images=img.cpu())
        assert float((model.last_prefill_logits.cpu()[0] - l_ref[0, -1]).abs().max()) < 1e-3, (W, paths[-1])
        n_checked += 1
    model.check_device_errors()
    buckets = {model._width_bucket(W, fx.n_image_tokens(cfg)) for W in widths}
    assert len(model._prefill_graphs) == len(buckets) <= 4, (len(model._prefill_graphs), sorted(buckets))
    assert paths.count("eager") == len(buckets) and paths.count("graph-capture") == len(buckets) and paths.count("graph-replay") == len(order) - 2 * len(buckets), paths
    # a bucketed prefill equals the exact-width one up to the library GEMMs' choice of kernel for the (different) row count: same tokens, same lengths
    W = 23
    ids = torch.randint(3, cfg.vocab_size, (1, W), generator=torch.Generator().manual_seed(100 + W))
    ids[0, 2 + W % 5] = IMAGE_TOKEN_INDEX
    a = model.generate(ids.cuda(), images=images[0], max_new_tokens=5, eos_token_id=None).cpu()
    la, lens_a = model.last_prefill_logits.float().cpu().clone(), [x.clone() for x in model.last_cache[1]]
    model.prefill_width_bucket = 0
    b = model.generate(ids.cuda(), images=images[0], max_new_tokens=5, eos_token_id=None).cpu()
    assert torch.equal(a, b) and float((la - model.last_prefill_logits.float().cpu()).abs().max()) < 1e-3
    assert all(torch.equal(x, y) for x, y in zip(lens_a, model.last_cache[1]))
    print(f"{n_checked} requests of {len(widths)} widths vs the oracle through {len(buckets)} cached prefills; buckets {sorted(buckets)}")


def test_decode_attention_merge_granules_of_an_earlier_request_cannot_be_consumed():
    """The in-kernel split merge of dl_attn_decode_rope accepts a granule whose tag equals f(position, layer).  A request that reaches a
    position an EARLIER request left granules at must not consume them: generate() and the eager decode forward() clear the workspace.
    The workspace is poisoned with correctly-tagged garbage for the first decode position of layers 0 and 1 (full-length rows, two
    splits) before each run; results must equal the unpoisoned run."""
    cfg = fx.tiny_config()
    sd = fx.make_state_dict(cfg, seed=5, predictor_gain=50.0)
    clip = fx.build_clip(cfg, seed=6)
    model = _build(cfg, sd, clip, torch.float32)
    assert model.attn_inkernel_combine
    ids = fx.make_prompt(cfg, 6, 40, seed=3)[None].cuda()
    images = fx.make_images(cfg, 1, seed=3).cuda()
    n_full = ids.shape[1] - 1 + fx.n_image_tokens(cfg)
    assert n_full > 64, "layers 0-1 must run with two KV splits for the merge to happen at all"
    clean = model.generate(ids, images=images, max_new_tokens=6, eos_token_id=None).cpu()
    clean_logits = model.last_prefill_logits.float().cpu().clone()

    def poison(layer):
        st = model._dstate
        tag = (((n_full & 0x7FFFFF) << 8) | layer) + 1  # the tag the first decode step of this prompt expects at `layer`
        g = (tag << 32) | int(np.float32(1e4).view(np.uint32))
        st.attn_ws.view(torch.int64).fill_(g if g < 2**63 else g - 2**64)

    for layer in (0, 1):
        poison(layer)
        out = model.generate(ids, images=images, max_new_tokens=6, eos_token_id=None).cpu()
        assert torch.equal(out, clean), f"generate() consumed stale merge granules at layer {layer}"
        assert torch.equal(model.last_prefill_logits.float().cpu(), clean_logits)
    # eager forward() decode on a fresh cache, poisoned right before the decode call
    o = model(ids, images=images)
    ref = model(clean[:, :1].cuda(), past_key_values=o.past_key_values).logits.cpu()
    for layer in (0, 1):
        o = model(ids, images=images)
        poison(layer)
        got = model(clean[:, :1].cuda(), past_key_values=o.past_key_values).logits.cpu()
        assert torch.equal(got, ref), f"forward() consumed stale merge granules at layer {layer}"
