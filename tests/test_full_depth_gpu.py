"""GPU: BASELINE.json configs at FULL DEPTH against the oracle run on the GPU box's host cores (bf16 as the reference computes, and fp32 on
the same bf16 weights as ground truth).  Every other model-level oracle comparison uses <= 4 layers; these check that nothing drifts over the
whole stack (error growth, RoPE positions after compaction, KV bookkeeping, decisions) on the exact models bench.py / tools/bench_configs.py time:

  configs[1]  LLaVA-1.5-7B, 32 layers, B=1, the bench prompt (35 + 576 + 20 = 631 tokens -> 170 after layer 2), prefill + 8 decode steps
  configs[2]  the same 7B model, B=32 ragged prompts (question lengths ~U[8,64]) through the packed-varlen batched path (decode batch > 24:
              library GEMMs + the ragged decode attention); two rows' prefill + 4 decode steps against the oracle's B=1 runs of those rows
              (B=1 is the only batched semantics the reference defines: SURVEY finding 2)
  configs[4]  LLaVA-1.5-13B, 40 layers, B=1, prompt 35 + 576 + 29 = 640 -> 179, prefill + 8 decode steps with output-text KV eviction

No comparison can end silently: a kept set that differs inside the bf16 rounding band of the k-th score, or a keep/evict logit pair on the
decision boundary, is FORCED to the HIP path's outcome on the oracle side (test hooks of oracle/ref_cpu.py) and every later step is still
compared; how many steps were compared / forced is printed and asserted."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx  # noqa: E402
from oracle.ref_cpu import Oracle  # noqa: E402

N_IMG, K_KEPT = 576, 115
ULP = 2.0**-7


def _compare_rows_vs_oracle(model, cfg, prompts, feats, rows, forced, label, dtype=torch.bfloat16, literal_tol=None):
    """HIP: ONE (possibly batched, ragged) run of prefill + len(forced) teacher-forced decode steps through the reference's own driver loop
    (BLTM:310-337).  Oracle: a B=1 run per row in `rows`, in the model dtype and in fp32.  Returns a dict of what was compared.
    literal_tol (fp32 models): the logits are held to max |hip - oracle| < literal_tol LITERALLY (north_star's 1e-3) instead of the noise-class bound."""
    B, n_steps = len(prompts), forced.shape[0]
    W = max(p.shape[0] for p in prompts)
    ids = torch.zeros(B, W, dtype=torch.long)
    am = torch.zeros(B, W, dtype=torch.long)
    for b, p in enumerate(prompts):
        ids[b, : p.shape[0]] = p
        am[b, : p.shape[0]] = 1
    ragged = B > 1
    model.debug_records = {}
    out = model(ids.cuda(), attention_mask=am.cuda() if ragged else None, image_features=feats)
    pkv = out.past_key_values
    cu = model.debug_records["cu_after"].cpu().tolist()
    n2 = [cu[b + 1] - cu[b] for b in range(B)]
    assert n2 == [p.shape[0] - 1 + K_KEPT for p in prompts]
    hip_logits = {b: [out.logits[b, n2[b] - 1].float().cpu()] for b in rows}
    pos_hip = model.debug_records["position_ids"].cpu()
    keep_hip = model.debug_records["keep_index"].cpu()
    dec_hip, gap_hip, tl_hip = [], [], []
    for j in range(n_steps):
        out = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = out.past_key_values
        for b in rows:
            hip_logits[b].append(out.logits[b, -1].float().cpu())
        dec_hip.append(model.debug_records["text_decision"].cpu().clone())
        tl = model.debug_records["text_logit"].cpu()
        gap_hip.append((tl[:, 0] - tl[:, 1]).abs())
        tl_hip.append(tl.float().clone())
    lens = pkv[1]
    for b in range(B):
        assert int(lens[0][b]) == prompts[b].shape[0] - 1 + N_IMG + n_steps
        assert int(lens[-1][b]) == n2[b] + sum(int(d[b]) for d in dec_hip)
    model.debug_records = None
    model.check_device_errors()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if "vision_tower" not in k}
    summary = {}
    for b in rows:
        ids_b, feats_b = prompts[b][None], feats[b : b + 1].cpu()
        o = Oracle(cfg, sd, dtype)
        with torch.no_grad():
            l_ref, p_ref = o.forward(ids_b, image_features=feats_b)
            keep_ref = o.records["keep_index"]
            # kept-token index set: bit-exact unless the reference's own k-th score is inside the bf16 rounding band (then only tokens inside
            # that band may differ, and both sides continue on the HIP path's set)
            kept_forced = False
            if not torch.equal(keep_hip[b : b + 1], keep_ref):
                score_ref = o.records["vision_score"].float()
                kth = torch.sort(score_ref[0], descending=True).values[K_KEPT - 1]
                diff = set(keep_hip[b].tolist()) ^ set(keep_ref[0].tolist())
                assert all(abs(float(score_ref[0, t]) - float(kth)) <= 4 * fx._ULP[dtype] * max(1.0, abs(float(kth))) for t in diff), (label, b, diff, float(kth))
                kept_forced = True
                o.force_keep_index = keep_hip[b : b + 1]
                l_ref, p_ref = o.forward(ids_b, image_features=feats_b)
            assert torch.equal(pos_hip[cu[b] : cu[b + 1]].long().view(-1), o.records["position_ids"].long().view(-1)), f"{label} row {b}: position ids after compaction"

            def step(orc, j, pkv_):
                """One decode step of an oracle.  A keep/evict logit pair inside the boundary band (fixtures.boundary_band: a few ulps of the logits'
                magnitude, on either side) may legitimately fall the other way: the step is then repeated with the HIP path's decision forced, so
                that every LATER step is still compared.  Outside the band a difference is an error, and the number of forced steps is bounded."""
                tok = forced[j][b : b + 1][:, None]
                fresh = lambda: (pkv_[0], [t.clone() for t in pkv_[1]])  # the cache adds to its length tensors in place (CU:153-164): a step that may be repeated runs on a copy
                l_, p_ = orc.forward(tok, past_key_values=fresh())
                tl_ = orc.records["text_logit"]
                gap_ = float((tl_[0, 0, 0] - tl_[0, 0, 1]).abs())
                was_forced = False
                if int(orc.records["text_decision"][0, 0]) != int(dec_hip[j][b]):
                    assert fx.decision_may_differ(tl_[0, 0], orc.dtype, tl_hip[j][b], dtype), (
                        f"{label} row {b} step {j}: eviction decision differs away from the boundary (oracle logits {tl_[0, 0].tolist()}, hip logits {tl_hip[j][b].tolist()}, "
                        f"bands {fx.boundary_band(tl_[0, 0], orc.dtype):.3g} / {fx.boundary_band(tl_hip[j][b], dtype):.3g}, gaps {gap_} / {float(gap_hip[j][b])})")
                    orc.force_text_decision = torch.tensor([[int(dec_hip[j][b])]])
                    l_, p_ = orc.forward(tok, past_key_values=fresh())
                    orc.force_text_decision = None
                    was_forced = True
                return l_, p_, was_forced

            ref_logits = [l_ref[0, -1].float()]
            forced_steps = {"bf16": [], "fp32": []}
            for j in range(n_steps):
                l_ref, p_ref, f_ = step(o, j, p_ref)
                ref_logits.append(l_ref[0, -1].float())
                if f_:
                    forced_steps["bf16"].append(j)
            lens_ref = p_ref[1]
            assert int(lens_ref[-1][0]) == int(lens[-1][b]) and int(lens_ref[0][0]) == int(lens[0][b]), f"{label} row {b}: KV lengths after the decode steps"
            del o, p_ref
            if dtype == torch.float32:
                truth = ref_logits  # the oracle in fp32 IS the ground truth
            else:
                o32 = Oracle(cfg, sd, torch.float32)  # casts the bf16 tensors up: identical parameter values
                o32.force_keep_index = keep_hip[b : b + 1]  # the fp32 run must follow the same kept set to be a ground truth for these logits
                l_32, p_32 = o32.forward(ids_b, image_features=feats_b.float())
                truth = [l_32[0, -1]]
                for j in range(n_steps):
                    l_32, p_32, f_ = step(o32, j, p_32)
                    truth.append(l_32[0, -1])
                    if f_:
                        forced_steps["fp32"].append(j)
                del o32, p_32
        worst = 0.0
        for j in range(n_steps + 1):  # EVERY step is compared (no early exit): boundary decisions were forced, not skipped
            e_hip = float((hip_logits[b][j] - truth[j]).abs().max())
            e_ref = float((ref_logits[j] - truth[j]).abs().max())
            if literal_tol is not None:  # fp32: north_star's number, literally, against the oracle in the same dtype
                e_lit = float((hip_logits[b][j] - ref_logits[j]).abs().max())
                assert e_lit < literal_tol, f"{label} row {b} step {j}: max |hip - oracle| = {e_lit} (literal bound {literal_tol})"
                worst = max(worst, e_lit / literal_tol)
                continue
            bound = 2.0 * e_ref + 2 * fx._ULP[dtype] * float(truth[j].abs().max())  # the reference's own noise class in THIS dtype (+ 2 ulp of the logit magnitude)
            assert e_hip <= bound, f"{label} row {b} step {j}: hip err {e_hip} vs reference err {e_ref}"
            worst = max(worst, e_hip / bound)
        n_forced = len(set(forced_steps["bf16"]) | set(forced_steps["fp32"]))
        assert n_forced <= fx.MAX_FORCED_DECISIONS, f"{label} row {b}: {n_forced} of {n_steps} decisions had to be forced ({forced_steps}): more than a boundary effect"
        summary[b] = dict(steps_compared=n_steps + 1, kept_set="forced to the HIP set (near-tied boundary)" if kept_forced else "bit-exact", decisions_forced=forced_steps,
                          evicted=n_steps - sum(int(d[b]) for d in dec_hip), worst_err_over_bound=round(worst, 3))
        print(f"{label} row {b}: {summary[b]}")
    assert all(s["steps_compared"] == n_steps + 1 for s in summary.values())
    return summary


def _calibrate(model, cfg, n_sys, n_q, seed, dtype=torch.bfloat16):
    """A random-init output-text predictor keeps (or evicts) every token; bench.py shifts its final bias until about half of a greedy
    continuation is evicted -- the regime `output_text_keep_rate=0.5` names.  The same calibration here, so that the full-depth comparisons
    contain kept AND evicted tokens (asserted by the tests)."""
    from bench import calibrate_text_predictor

    g = torch.Generator().manual_seed(100 + seed)
    images = torch.randn((1, 3, 336, 336), generator=g).to(dtype).cuda()
    frac = calibrate_text_predictor(model, fx.make_prompt(cfg, n_sys, n_q, seed=seed)[None].cuda(), images, 24)
    print(f"text predictor calibrated: keep fraction {frac}")
    return frac


@pytest.fixture(scope="module")
def model7b():
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig()  # LLaVA-1.5-7B defaults: 32 layers, sparse_layer 2, keep 0.2
    assert cfg.num_hidden_layers == 32 and cfg.hidden_size == 4096
    model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)  # == bench.py's model
    _calibrate(model, cfg, 35, 20, 0)
    yield cfg, model
    del model
    torch.cuda.empty_cache()


def test_configs1_32_layers_prefill_and_decode_vs_oracle(model7b):
    cfg, model = model7b
    g = torch.Generator().manual_seed(0)
    prompt = fx.make_prompt(cfg, 35, 20, seed=0)
    images = torch.randn((1, 3, 336, 336), generator=g).to(torch.bfloat16)
    feats = model.encode_images(images.cuda())  # the same projector output feeds both sides (CLIP parity: test_kernels_gpu)
    forced = fx.make_forced_tokens(cfg, 8, 1, seed=5)
    s = _compare_rows_vs_oracle(model, cfg, [prompt], feats, [0], forced, "configs[1] 7B x 32 layers")
    assert s[0]["steps_compared"] == 9
    assert 0 < s[0]["evicted"] < 8, "the eviction must go both ways inside the compared steps"


def test_configs1_32_layers_fp16_vs_oracle():
    """configs[1] at full depth in fp16 -- the dtype every loader of the reference's eval / bench scripts uses (BLD:62 `torch_dtype=torch.float16`,
    VQAL:165-167 `.half()`): 7B x 32 layers, the bench prompt, prefill + 8 teacher-forced decode steps with eviction against the oracle computing in fp16 on the
    host (and in fp32 on the same fp16 weights as ground truth).  Same bounds as the bf16 test with fp16 ulps: kept set / position ids / decisions / KV lengths
    bit-exact (boundary cases forced and counted), logits in the reference's own fp16 noise class."""
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig()
    dtype = torch.float16
    model = build_random_model(cfg, dtype=dtype, device="cuda", seed=0, predictor_gain=50.0)
    _calibrate(model, cfg, 35, 20, 0, dtype)
    g = torch.Generator().manual_seed(0)
    prompt = fx.make_prompt(cfg, 35, 20, seed=0)
    images = torch.randn((1, 3, 336, 336), generator=g).to(dtype)
    feats = model.encode_images(images.cuda())
    assert torch.isfinite(feats).all()
    forced = fx.make_forced_tokens(cfg, 8, 1, seed=5)
    s = _compare_rows_vs_oracle(model, cfg, [prompt], feats, [0], forced, "configs[1] 7B x 32 layers fp16", dtype=dtype)
    assert s[0]["steps_compared"] == 9
    assert 0 < s[0]["evicted"] < 8, "the eviction must go both ways inside the compared steps"
    del model
    torch.cuda.empty_cache()


def test_configs2_batch32_ragged_32_layers_two_rows_vs_oracle(model7b):
    """configs[2] at full depth: 32 ragged requests in one packed batch; rows 0 and 17 against the oracle's B=1 runs, prefill + 4 decode steps."""
    cfg, model = model7b
    g = torch.Generator().manual_seed(1)
    B = 32
    n_q = torch.randint(8, 65, (B,), generator=g).tolist()  # question lengths ~U[8,64]
    prompts = [fx.make_prompt(cfg, 35, n_q[b], seed=b) for b in range(B)]
    images = torch.randn((B, 3, 336, 336), generator=g).to(torch.bfloat16)
    feats = model.encode_images(images.cuda())
    forced = fx.make_forced_tokens(cfg, 4, B, seed=6)
    s = _compare_rows_vs_oracle(model, cfg, prompts, feats, [0, 17], forced, "configs[2] 7B x 32 layers, B=32 ragged")
    st = model._dstate
    assert st.B == B and not st.use_gemv and st.use_smallm and st.use_lp_mlp, "B=32 decodes on dl_linear_packed (q|k|v, MLP) + dl_gemm_smallm (o_proj) + ragged attention"
    assert sorted(s) == [0, 17] and all(v["steps_compared"] == 5 for v in s.values())


def test_configs0_keep_rate_one_equals_dense_path_32_layers(model7b):
    """BASELINE configs[0] at full size: 7B x 32 layers, input_ids [[1, -200, 1]], vision_keep_rate = 1.0, text predictors off, greedy 16 tokens.  The
    sparsified path (vision predictor + top-k with k = 576 + compaction: all identities) must equal the dense path (the same weights with the vision
    predictor switched off, DML:1826-1831) bit for bit -- tokens, prefill logits -- and drop nothing from any layer's KV."""
    cfg, model = model7b
    sc = model.config.sparse_config
    assert sc is model.model.config.sparse_config
    saved = dict(sc)
    try:
        g = torch.Generator().manual_seed(0)
        images = torch.randn((1, 3, 336, 336), generator=g).to(torch.bfloat16).cuda()
        feats = model.encode_images(images)
        ids = torch.tensor([[1, -200, 1]]).cuda()
        sc.update(vision_keep_rate=1.0, use_vision_predictor=True, use_text_predictor=False, use_output_text_predictor=False)
        a = model.generate(ids, image_features=feats, max_new_tokens=16, eos_token_id=None)
        la = model.last_prefill_logits.clone()
        o = model(ids, image_features=feats)
        lens = [int(t[0]) for t in o.past_key_values[1]]
        assert lens == [578] * cfg.num_hidden_layers, "keep rate 1.0 drops nothing"
        l_fwd = o.logits[0, -1].float().clone()
        sc.update(use_vision_predictor=False)
        b = model.generate(ids, image_features=feats, max_new_tokens=16, eos_token_id=None)
        lb = model.last_prefill_logits.clone()
        o2 = model(ids, image_features=feats)
        assert a.shape == (1, 16) and torch.equal(a, b), (a.tolist(), b.tolist())
        assert torch.equal(la, lb) and torch.equal(l_fwd, o2.logits[0, -1].float())
        assert [int(t[0]) for t in o2.past_key_values[1]] == lens
    finally:
        sc.clear()
        sc.update(saved)
        model._dstate = None


def test_configs4_13b_40_layers_prefill_and_decode_vs_oracle():
    """configs[4] at full depth: LLaVA-1.5-13B, all 40 layers, the model tools/bench_configs.py times."""
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40)
    model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
    _calibrate(model, cfg, 35, 29, 9)
    g = torch.Generator().manual_seed(2)
    prompt = fx.make_prompt(cfg, 35, 29, seed=9)
    images = torch.randn((1, 3, 336, 336), generator=g).to(torch.bfloat16)
    feats = model.encode_images(images.cuda())
    forced = fx.make_forced_tokens(cfg, 8, 1, seed=7)
    s = _compare_rows_vs_oracle(model, cfg, [prompt], feats, [0], forced, "configs[4] 13B x 40 layers")
    assert s[0]["steps_compared"] == 9
    assert 0 < s[0]["evicted"] < 8, "the eviction must go both ways inside the compared steps"
    del model
    torch.cuda.empty_cache()


def test_configs1_32_layers_fp32_literal_1e3():
    """north_star's "logits within 1e-3", LITERALLY, at full depth: LLaVA-1.5-7B, all 32 layers, fp32 weights through the HIP path
    against the fp32 oracle -- max |hip - oracle| < 1e-3 over the whole vocabulary for the prefill's last token and 8 decode steps; kept-token set,
    position ids, eviction decisions and per-layer KV lengths bit-exact (a decision inside the fp32 boundary band may be forced: at most
    MAX_FORCED_DECISIONS).  The 16-bit tests above hold the bf16 path to the reference's own bf16 noise class; this one pins the arithmetic itself."""
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    cfg = DynamicLlavaConfig()
    assert cfg.num_hidden_layers == 32 and cfg.hidden_size == 4096
    dtype = torch.float32
    model = build_random_model(cfg, dtype=dtype, device="cuda", seed=0, predictor_gain=50.0)  # bench.py's weights before the cast to bf16
    _calibrate(model, cfg, 35, 20, 0, dtype)
    g = torch.Generator().manual_seed(0)
    prompt = fx.make_prompt(cfg, 35, 20, seed=0)
    images = torch.randn((1, 3, 336, 336), generator=g).to(dtype)
    feats = model.encode_images(images.cuda())
    forced = fx.make_forced_tokens(cfg, 8, 1, seed=5)
    s = _compare_rows_vs_oracle(model, cfg, [prompt], feats, [0], forced, "configs[1] 7B x 32 layers fp32", dtype=dtype, literal_tol=1e-3)
    assert s[0]["steps_compared"] == 9 and s[0]["kept_set"] == "bit-exact"
    assert 0 < s[0]["evicted"] < 8, "the eviction must go both ways inside the compared steps"
    del model
    torch.cuda.empty_cache()
