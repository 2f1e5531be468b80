"""GPU: the BASELINE.json configs beyond the bench line, at full layer WIDTH (few layers so the oracle finishes in seconds),
plus size-independent properties at full sizes.

  C3  batch=32 images, ragged prompt lengths + per-row eviction (packed varlen; decode batch > 16 => library GEMM path)
  mid batch  4..16 rows on dl_gemm_smallm + partial-sum consumers
  C5  LLaVA-1.5-13B width, long decode with incremental output-text KV eviction (slab growth, split-KV, hipGraph vs eager)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx  # noqa: E402
from oracle.ref_cpu import Oracle  # noqa: E402


def _build(cfg_ns, sd, dtype):
    from dynamic_llava_amd.builder import build_from_state_dict
    from dynamic_llava_amd.config import DynamicLlavaConfig

    return build_from_state_dict(DynamicLlavaConfig.from_namespace(cfg_ns), sd, None, dtype=dtype, device="cuda")


def _b1_step_following(model, tok, pkv, dec_batched, gap_batched, where):
    """One B=1 decode step whose keep/evict BOOKKEEPING follows the batched run's decision (model.force_text_decision), so that a logit pair on
    the decision boundary -- which the two GEMM paths may legitimately round to different sides -- never ends the comparison: every later
    step is still compared.  The step's own decision is still checked: outside the boundary band (fixtures.boundary_band: a few ulps of the logits'
    magnitude) it must agree; the callers bound how many steps may need the band.
    Returns (output, True if the B=1 run's own decision differed and was overridden)."""
    model.force_text_decision = torch.tensor([int(dec_batched)])
    o1 = model(tok, past_key_values=pkv)
    model.force_text_decision = None
    own = int(model.debug_records["text_decision"][0])
    tl = model.debug_records["text_logit"].cpu()
    gap1 = float((tl[0, 0] - tl[0, 1]).abs())
    if own != int(dec_batched):
        band = fx.boundary_band(tl[0], torch.bfloat16)  # (both runs are this model in bf16: their logits have the same magnitude)
        assert min(gap1, float(gap_batched)) <= band, f"{where}: eviction decision differs away from the boundary (B=1 gap {gap1}, batched gap {float(gap_batched)}, band {band:.3g})"
        return o1, True
    return o1, False


def test_c3_batch32_ragged_rows_equal_b1_and_oracle():
    """Every row of a ragged B=32 batch must equal its own B=1 run (SURVEY finding 2: that is the only well-defined batched
    semantics), for prefill logits, per-step eviction decisions, per-row KV lengths and greedy tokens."""
    dtype = torch.bfloat16
    cfg = fx.llava7b_config(num_hidden_layers=3)
    cfg.vocab_size = 4096
    sd = fx.make_state_dict(cfg, seed=11, predictor_gain=50.0)
    model = _build(cfg, sd, dtype)
    g = torch.Generator().manual_seed(1)
    B, steps = 32, 12
    n_q = torch.randint(8, 65, (B,), generator=g).tolist()  # question lengths ~U[8,64]
    prompts = [fx.make_prompt(cfg, 35, n_q[b], seed=b) for b in range(B)]
    feats = torch.randn(B, 576, 4096, generator=g).to(dtype)
    W = max(p.shape[0] for p in prompts)
    ids = torch.zeros(B, W, dtype=torch.long)
    am = torch.zeros(B, W, dtype=torch.long)
    for b, p in enumerate(prompts):
        ids[b, : p.shape[0]] = p
        am[b, : p.shape[0]] = 1
    out = model.generate(ids.cuda(), attention_mask=am.cuda(), image_features=feats.cuda(), max_new_tokens=steps, eos_token_id=None)
    lens_b = [t.clone() for t in model.last_cache[1]]
    logits_b = model.last_prefill_logits.clone()
    assert out.shape == (B, steps)
    assert lens_b[0].tolist() == [35 + 576 + n_q[b] + steps - 1 for b in range(B)]
    kept = [int(lens_b[-1][b]) - (35 + 115 + n_q[b]) for b in range(B)]
    assert all(0 <= k <= steps - 1 for k in kept) and 0 < sum(kept) < B * (steps - 1), "eviction must be exercised both ways"
    ulp = 2.0**-7
    # teacher-forced forward() loop, batched vs B=1 (different GEMM paths: hipBLASLt at B=32, dl_gemv at B=1 => same noise
    # class, not bit-equal; decisions compared away from the decision boundary)
    forced = fx.make_forced_tokens(cfg, 6, B, seed=3)
    model.debug_records = {}
    ob = model(ids.cuda(), attention_mask=am.cuda(), image_features=feats.cuda())
    pkv = ob.past_key_values
    cu = model.debug_records["cu_after"].cpu().tolist()
    rows = [0, 7, 19, 31]
    hist = {b: [ob.logits[b, cu[b + 1] - cu[b] - 1].cpu()] for b in rows}
    dec_b, gap_b = [], []
    for j in range(6):
        ob = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = ob.past_key_values
        for b in rows:
            hist[b].append(ob.logits[b, -1].cpu())
        dec_b.append(model.debug_records["text_decision"].cpu().clone())
        tl = model.debug_records["text_logit"].cpu()
        gap_b.append((tl[:, 0] - tl[:, 1]).abs())
    for b in rows:
        o1 = model(prompts[b][None].cuda(), image_features=feats[b : b + 1].cuda())
        p1 = o1.past_key_values
        assert float((o1.logits[0, -1].cpu() - hist[b][0]).abs().max()) <= 8 * ulp * float(hist[b][0].abs().max()), f"row {b} prefill"
        # generate() runs lm_head on the last rows only, forward() on all rows: different hipBLASLt kernels, <= 1-2 ulp apart
        assert float((logits_b[b].cpu() - hist[b][0]).abs().max()) <= 2 * ulp * float(hist[b][0].abs().max()), "generate() vs forward() prefill"
        n_forced = 0
        for j in range(6):  # every step is compared: a boundary decision is followed (forced), never a reason to stop
            o1, f_ = _b1_step_following(model, forced[j][b : b + 1][:, None].cuda(), p1, dec_b[j][b], gap_b[j][b], f"row {b} step {j}")
            p1 = o1.past_key_values
            n_forced += f_
            assert float((o1.logits[0, -1].cpu() - hist[b][j + 1]).abs().max()) <= 8 * ulp * float(hist[b][j + 1].abs().max()), f"row {b} step {j}"
        assert int(p1[1][-1][0]) == int(pkv[1][-1][b]) and int(p1[1][0][0]) == int(pkv[1][0][b]), f"row {b}: KV lengths after 6 steps"
        print(f"C3 row {b}: 7 of 7 logit vectors compared with its B=1 run, {n_forced} boundary decisions followed")
        assert n_forced <= fx.MAX_FORCED_DECISIONS, f"row {b}: {n_forced} of 6 decisions had to be followed: more than a boundary effect"
    model.debug_records = None
    # oracle on two rows (B=1 reference semantics), prefill logits
    for b in (0, 17):
        o = Oracle(cfg, sd, dtype)
        o32 = Oracle(cfg, {k: v.to(dtype) for k, v in sd.items()}, torch.float32)
        with torch.no_grad():
            l_ref, _ = o.forward(prompts[b][None], image_features=feats[b : b + 1])
            l_32, _ = o32.forward(prompts[b][None], image_features=feats[b : b + 1].float())
        e_hip = float((logits_b[b].cpu() - l_32[0, -1]).abs().max())
        e_ref = float((l_ref[0, -1] - l_32[0, -1]).abs().max())
        assert e_hip <= 2.0 * e_ref + 2 * ulp * float(l_32.abs().max()), (b, e_hip, e_ref)


def test_c5_13b_width_long_decode_with_eviction():
    """13B width (H=5120, 40x128 heads, I=13824), 3 layers, prompt 35+576+29 = 640 -> 179, decode to a total length of 2048
    (1408 steps) with output-text eviction (CU:153-164, DML:2377-2391): hipGraph replay == eager launches bit-for-bit, KV-length bookkeeping exact,
    slab growth through the forward() API, the first 8 steps against the oracle on the host, and -- VERDICT r5 item 3b -- the oracle's op sequence (run by
    PyTorch-ROCm on the device: the reference's eager GPU path) teacher-forced through ALL 1407 decode steps in bf16 and in fp32:
    the eviction decision and both KV lengths at EVERY step (a decision inside the boundary band is forced on the oracle side and counted: at most
    MAX_FORCED_DECISIONS per 256 steps), the logits in the reference's own noise class at the first 8 steps, every 64th step and the last one -- so the
    split-KV schedule the long row walks through (one attention workgroup per head, then four, then the six-split stand-alone launch with the in-kernel
    combine) is compared with the reference at its far end, not only with itself."""
    dtype = torch.bfloat16
    cfg = fx.llava13b_config(num_hidden_layers=3)
    cfg.vocab_size = 4096
    cfg.mm_hidden_size = 1024
    sd = fx.make_state_dict(cfg, seed=13, predictor_gain=50.0)
    model = _build(cfg, sd, dtype)
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(1, 576, 5120, generator=g).to(dtype)
    ids = fx.make_prompt(cfg, 35, 29, seed=9)[None]
    n_new = 2048 - 640
    model.use_hip_graph = True
    a = model.generate(ids.cuda(), image_features=feats.cuda(), max_new_tokens=n_new, eos_token_id=None)
    lens_a = [t.clone() for t in model.last_cache[1]]
    model.use_hip_graph = False
    b = model.generate(ids.cuda(), image_features=feats.cuda(), max_new_tokens=n_new, eos_token_id=None)
    lens_b = model.last_cache[1]
    assert torch.equal(a, b), "graph replay and eager launches must produce identical tokens"
    assert int(lens_a[0][0]) == 640 + n_new - 1 == int(lens_b[0][0]) and int(lens_a[-1][0]) == int(lens_b[-1][0])
    kept = int(lens_a[-1][0]) - 179
    assert 0 < kept < n_new - 1, f"kept {kept} of {n_new - 1} generated tokens"
    # forward() loop (the reference's own driver) reproduces generate() over the WHOLE generation; the caller-owned slab (reserve 256) must grow
    model.debug_records = {}
    out = model(ids.cuda(), image_features=feats.cuda())
    pkv = out.past_key_values
    cap0 = pkv.t_cap
    tok = out.logits[:, -1].argmax(-1)
    dec, gap_hip, tl_hip = [], [], []
    n_check = n_new - 1
    logit_steps = sorted(set(range(8)) | set(range(63, n_check, 64)) | {n_check - 1})  # decode steps whose logits are compared (index j: after feeding token j)
    hip_logits = {-1: out.logits[0, -1].float().cpu()}
    for j in range(n_check):
        assert int(tok[0]) == int(a[0, j]), f"forward-loop token {j}"
        out = model(tok[:, None], past_key_values=pkv)
        pkv = out.past_key_values
        dec.append(int(model.debug_records["text_decision"][0]))
        tl_h = model.debug_records["text_logit"].cpu()
        gap_hip.append(float((tl_h[0, 0] - tl_h[0, 1]).abs()))
        tl_hip.append(tl_h[0].float().clone())
        if j in logit_steps:
            hip_logits[j] = out.logits[0, -1].float().cpu()
        tok = out.logits[:, -1].argmax(-1)
    assert int(tok[0]) == int(a[0, n_check])
    assert pkv.t_cap > cap0, "slab must have grown"
    assert int(pkv[1][-1][0]) == 179 + sum(dec) == int(lens_a[-1][0]) and int(pkv[1][0][0]) == 640 + n_check
    assert pkv[0][-1][0].shape[-2] == 179 + sum(dec) and pkv[0][0][0].shape[-2] == 640 + n_check
    model.debug_records = None
    model.check_device_errors()
    # (1) the first steps against the oracle run on the HOST (the pinned restatement), bf16 and fp32: logits in the reference's own noise class
    ulp = 2.0**-7
    sd32 = {k: v.to(dtype) for k, v in sd.items()}

    def make_step(orc):
        def step(j, pkv_):
            """One oracle step.  The reference's cache adds to its per-layer length tensors IN PLACE (CU:153-164, restated by the oracle), so a step that may be
            repeated runs on a copy of the lengths.  A keep/evict logit pair on the boundary may fall the other way between summation orders: the step is then
            repeated with the HIP path's decision forced (oracle test hook), and the comparison goes on."""
            fresh = lambda: (pkv_[0], [t.clone() for t in pkv_[1]])
            tok_ = a[:, j : j + 1].to(orc.device)
            l_, p_ = orc.forward(tok_, past_key_values=fresh())
            tl_ = orc.records["text_logit"]
            if int(orc.records["text_decision"][0, 0]) != dec[j]:
                assert fx.decision_may_differ(tl_[0, 0].cpu(), orc.dtype, tl_hip[j], dtype), (
                    f"eviction decision differs away from the boundary, step {j}: oracle logits {tl_[0, 0].tolist()}, hip logits {tl_hip[j].tolist()}")
                orc.force_text_decision = torch.tensor([[dec[j]]], device=orc.device)
                l_, p_ = orc.forward(tok_, past_key_values=fresh())
                orc.force_text_decision = None
                return l_, p_, True
            return l_, p_, False
        return step

    def noise_class(hip, ref, truth):
        e_hip, e_ref = float((hip - truth).abs().max()), float((ref - truth).abs().max())
        bound = 2.0 * e_ref + 2 * ulp * float(truth.abs().max())
        assert e_hip <= bound, (e_hip, e_ref)
        return e_hip / bound

    n_host = 8
    with torch.no_grad():
        o, o32 = Oracle(cfg, sd, dtype), Oracle(cfg, sd32, torch.float32)
        st_o, st_32 = make_step(o), make_step(o32)
        l_ref, p_ref = o.forward(ids, image_features=feats)
        l_32, p_32 = o32.forward(ids, image_features=feats.float())
        noise_class(hip_logits[-1], l_ref[0, -1].float(), l_32[0, -1])
        host_forced = 0
        for j in range(n_host):
            l_ref, p_ref, f1 = st_o(j, p_ref)
            l_32, p_32, f2 = st_32(j, p_32)
            host_forced += f1 + f2
            noise_class(hip_logits[j], l_ref[0, -1].float(), l_32[0, -1])
        assert int(p_ref[1][-1][0]) == 179 + sum(dec[:n_host]) and host_forced <= 2 * fx.MAX_FORCED_DECISIONS
        host_ref_logits = l_ref[0, -1].float().clone()
        del o, o32, p_ref, p_32
        # (2) ALL steps: the same oracle code with its tensors on the device -- the reference's own eager op sequence (torch.cat cache, one
        # host sync per layer on the decision, SDPA) executed by PyTorch-ROCm, which is what "the reference GPU path" is; on the host cores 2 x 1407 steps of a
        # 13B-wide model cost nine minutes of a suite that runs in ten.  Decisions and BOTH KV lengths at every step, logits at the checkpoints; step n_host - 1
        # ties this execution to the host-run one above (same noise-class bound, and against each other).
        og, og32 = Oracle(cfg, sd, dtype, device="cuda"), Oracle(cfg, sd32, torch.float32, device="cuda")
        st_g, st_g32 = make_step(og), make_step(og32)
        l_ref, p_ref = og.forward(ids.cuda(), image_features=feats.cuda())
        l_32, p_32 = og32.forward(ids.cuda(), image_features=feats.cuda().float())
        worst = noise_class(hip_logits[-1], l_ref[0, -1].float().cpu(), l_32[0, -1].cpu())
        forced = {"bf16": [], "fp32": []}
        n_sparse = 179
        for j in range(n_check):
            l_ref, p_ref, f1 = st_g(j, p_ref)
            l_32, p_32, f2 = st_g32(j, p_32)
            if f1:
                forced["bf16"].append(j)
            if f2:
                forced["fp32"].append(j)
            n_sparse += dec[j]
            # both KV lengths, every step, both dtypes (CU:153-164: layers < sparse_layer always append, layers >= sparse_layer append iff the decision says keep)
            assert int(p_ref[1][-1][0]) == n_sparse == int(p_32[1][-1][0]), f"KV length of the sparse layers after step {j}"
            assert int(p_ref[1][0][0]) == 640 + j + 1 == int(p_32[1][0][0]), f"KV length of the dense layers after step {j}"
            assert p_ref[0][-1][0].shape[-2] == n_sparse and p_ref[0][0][0].shape[-2] == 640 + j + 1
            if j in hip_logits:
                worst = max(worst, noise_class(hip_logits[j], l_ref[0, -1].float().cpu(), l_32[0, -1].cpu()))
            if j == n_host - 1:  # the device-run and the host-run execution of the restatement agree to the bf16 noise class at the step both reach
                noise_class(host_ref_logits, l_ref[0, -1].float().cpu(), l_32[0, -1].cpu())
        assert n_sparse == int(lens_a[-1][0])
        for name, steps_ in forced.items():
            for w0 in range(0, n_check, 256):
                n_w = sum(1 for j in steps_ if w0 <= j < w0 + 256)
                # (the fp32 run differs from the bf16 HIP path by the bf16 rounding noise itself, so more of its decisions sit inside the band: one more allowed; seen: <= 1 per window)
                lim = fx.MAX_FORCED_DECISIONS + (1 if name == "fp32" else 0)
                assert n_w <= lim, f"{name} oracle: {n_w} decisions forced in steps [{w0}, {w0 + 256}): more than a boundary effect ({steps_})"
        print(f"C5: oracle teacher-forced through {n_check} steps: decisions + both KV lengths at every step, {len(hip_logits)} logit vectors compared "
              f"(steps {sorted(hip_logits)[:10]}..., worst err / bound {worst:.3f}), boundary decisions forced: {forced}, kept {sum(dec)} of {n_check}")


@pytest.mark.parametrize("B,width", [(4, "7b"), (7, "7b"), (16, "7b"), (20, "7b"), (24, "7b"), (28, "7b"), (32, "7b"), (12, "13b"), (16, "13b"), (32, "13b")])
def test_mid_batch_decode_smallm_rows_equal_b1(B, width):
    """Decode batches 4..32 run q|k|v (up to 15 rows) and o_proj on dl_gemm_smallm (+ partial-sum consumers), the MLP -- and q|k|v from 16 rows on -- on
    dl_linear_packed (16..20 rows: as fp32 partial sums of its two k ranges, added by dl_attn_decode_rope_parts -- round 6).  Every row of a ragged batch must match its own B=1 run
    (dl_gemv path): greedy tokens and per-step eviction decisions away from decision boundaries, logits in the same noise class."""
    dtype = torch.bfloat16
    cfg = fx.llava7b_config(num_hidden_layers=3) if width == "7b" else fx.llava13b_config(num_hidden_layers=3)  # 13B: 8 units per workgroup, down_proj in 3 units x 2 k ranges
    cfg.vocab_size = 4096
    sd = fx.make_state_dict(cfg, seed=11, predictor_gain=50.0)
    model = _build(cfg, sd, dtype)
    assert model.gemv_max_decode_batch < B <= model.smallm_max_decode_batch
    g = torch.Generator().manual_seed(5)
    n_q = torch.randint(8, 40, (B,), generator=g).tolist()
    prompts = [fx.make_prompt(cfg, 35, n_q[b], seed=20 + b) for b in range(B)]
    feats = torch.randn(B, 576, cfg.hidden_size, generator=g).to(dtype)
    W = max(p.shape[0] for p in prompts)
    ids = torch.zeros(B, W, dtype=torch.long)
    am = torch.zeros(B, W, dtype=torch.long)
    for b, p in enumerate(prompts):
        ids[b, : p.shape[0]] = p
        am[b, : p.shape[0]] = 1
    ulp = 2.0**-7
    forced = fx.make_forced_tokens(cfg, 8, B, seed=4)
    model.debug_records = {}
    ob = model(ids.cuda(), attention_mask=am.cuda(), image_features=feats.cuda())
    pkv = ob.past_key_values
    hist, dec_b, gap_b = [], [], []
    for j in range(8):
        ob = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = ob.past_key_values
        hist.append(ob.logits[:, -1].cpu())
        dec_b.append(model.debug_records["text_decision"].cpu().clone())
        tl = model.debug_records["text_logit"].cpu()
        gap_b.append((tl[:, 0] - tl[:, 1]).abs())
    assert model._dstate.use_smallm and model._dstate.use_lp_mlp and not model._dstate.use_gemv, "this batch size must run on dl_gemm_smallm (o_proj) + dl_linear_packed (MLP)"
    lens_b = [t.clone() for t in pkv[1]]
    for b in sorted({0, B // 2, B - 1}):
        o1 = model(prompts[b][None].cuda(), image_features=feats[b : b + 1].cuda())
        p1 = o1.past_key_values
        kept, n_forced = 0, 0
        for j in range(8):  # every step is compared: a boundary decision is followed (forced), never a reason to stop
            o1, f_ = _b1_step_following(model, forced[j][b : b + 1][:, None].cuda(), p1, dec_b[j][b], gap_b[j][b], f"B={B} row {b} step {j}")
            p1 = o1.past_key_values
            n_forced += f_
            kept += int(dec_b[j][b])
            ref = o1.logits[0, -1].cpu()
            assert float((ref - hist[j][b]).abs().max()) <= 8 * ulp * float(ref.abs().max()), f"row {b} step {j}"
        assert int(lens_b[-1][b]) == 35 + 115 + n_q[b] + kept and int(lens_b[0][b]) == 35 + 576 + n_q[b] + 8
        assert int(p1[1][-1][0]) == int(lens_b[-1][b])
        print(f"B={B} row {b}: 8 of 8 steps compared with its B=1 run, {n_forced} boundary decisions followed")
        assert n_forced <= fx.MAX_FORCED_DECISIONS, f"B={B} row {b}: {n_forced} of 8 decisions had to be followed: more than a boundary effect"
    model.debug_records = None
    # generate(): hipGraph replay == eager launches on this path too
    model.use_hip_graph = True
    a = model.generate(ids.cuda(), attention_mask=am.cuda(), image_features=feats.cuda(), max_new_tokens=6, eos_token_id=None)
    model.use_hip_graph = False
    c = model.generate(ids.cuda(), attention_mask=am.cuda(), image_features=feats.cuda(), max_new_tokens=6, eos_token_id=None)
    assert torch.equal(a, c)
