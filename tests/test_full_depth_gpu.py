"""GPU: BASELINE.json configs[1] at FULL DEPTH -- LLaVA-1.5-7B, all 32 layers, bf16, the bench prompt (35 + 576 + 20 = 631
tokens -> 170 after layer 2, vision_keep_rate 0.2, output-text KV eviction on) -- against the oracle run on the GPU box's host
cores (bf16 as the reference computes, and fp32 on the same bf16 weights as ground truth).  Every other model-level oracle
comparison uses <= 4 layers; this one checks that nothing drifts over 32 (error growth, RoPE positions after compaction, KV
bookkeeping, decisions) on the exact weights bench.py times."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx  # noqa: E402
from oracle.ref_cpu import Oracle  # noqa: E402

N_SYS, N_Q, N_IMG = 35, 20, 576


def test_configs1_32_layers_prefill_and_decode_vs_oracle():
    from dynamic_llava_amd.builder import build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    dtype = torch.bfloat16
    cfg = DynamicLlavaConfig()  # LLaVA-1.5-7B defaults: 32 layers, sparse_layer 2, keep 0.2
    assert cfg.num_hidden_layers == 32 and cfg.hidden_size == 4096
    model = build_random_model(cfg, dtype=dtype, device="cuda", seed=0, predictor_gain=50.0)  # == bench.py's model
    g = torch.Generator().manual_seed(0)
    ids = fx.make_prompt(cfg, N_SYS, N_Q, seed=0)[None]
    images = torch.randn((1, 3, 336, 336), generator=g).to(dtype)
    feats = model.encode_images(images.cuda())  # the same projector output feeds both sides (CLIP parity: test_kernels_gpu)
    n_steps = 8
    forced = fx.make_forced_tokens(cfg, n_steps, 1, seed=5)
    # ---- HIP path, the reference's own driver loop (BLTM:310-337) ----
    model.debug_records = {}
    out = model(ids.cuda(), image_features=feats)
    pkv = out.past_key_values
    hip_logits = [out.logits[0, -1].float().cpu()]
    pos_hip = model.debug_records["position_ids"].cpu()
    keep_hip = model.debug_records["keep_index"].cpu()
    score_hip = model.debug_records["vision_score"].float().cpu()
    assert out.logits.shape == (1, N_SYS + 115 + N_Q, cfg.vocab_size)
    dec_hip, gap_hip = [], []
    for j in range(n_steps):
        out = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = out.past_key_values
        hip_logits.append(out.logits[0, -1].float().cpu())
        dec_hip.append(int(model.debug_records["text_decision"][0]))
        tl = model.debug_records["text_logit"].cpu()
        gap_hip.append(float((tl[0, 0] - tl[0, 1]).abs()))
    lens = pkv[1]
    assert int(lens[0][0]) == N_SYS + N_IMG + N_Q + n_steps and int(lens[-1][0]) == N_SYS + 115 + N_Q + sum(dec_hip)
    model.debug_records = None
    # ---- oracle on the host cores: bf16 (what the reference computes) and fp32 on the same bf16 weights (truth) ----
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if "vision_tower" not in k}
    o = Oracle(cfg, sd, dtype)
    with torch.no_grad():
        l_ref, p_ref = o.forward(ids, image_features=feats.cpu())
        pos_ref, keep_ref = o.records["position_ids"], o.records["keep_index"]
        # kept-token index set: bit-exact unless the reference's own k-th score is inside the bf16 rounding band (then only
        # tokens inside that band may differ)
        forced_keep = None
        if not torch.equal(keep_hip, keep_ref):
            score_ref = o.records["vision_score"].float()
            kth = torch.sort(score_ref[0], descending=True).values[114]
            diff = set(keep_hip[0].tolist()) ^ set(keep_ref[0].tolist())
            assert all(abs(float(score_ref[0, t]) - float(kth)) <= 4 * 2.0**-7 * max(1.0, abs(float(kth))) for t in diff), (diff, float(kth))
            # both sets are valid top-k sets of scores that agree to the last bf16 bit: continue with the HIP path's set on both sides
            forced_keep = keep_hip
            o.force_keep_index = forced_keep
            l_ref, p_ref = o.forward(ids, image_features=feats.cpu())
            pos_ref = o.records["position_ids"]
        assert torch.equal(pos_hip.long().view(-1), pos_ref.long().view(-1)), "position ids after compaction"
        def step(orc, j, pkv_):
            """One decode step of an oracle.  A keep/evict logit pair that sits on the boundary (|gap| <= 0.5 on either side) may
            legitimately fall the other way: the step is then repeated with the HIP path's decision forced (oracle test hook), so that
            every LATER step is still compared instead of the comparison ending here.  Away from the boundary a difference is an error."""
            l_, p_ = orc.forward(forced[j][:, None], past_key_values=pkv_)
            tl_ = orc.records["text_logit"]
            gap_ = float((tl_[0, 0, 0] - tl_[0, 0, 1]).abs())
            was_forced = False
            if int(orc.records["text_decision"][0, 0]) != dec_hip[j]:
                assert min(gap_, gap_hip[j]) <= 0.5, f"eviction decision differs away from the boundary, step {j}: oracle gap {gap_}, hip gap {gap_hip[j]}"
                orc.force_text_decision = torch.tensor([[dec_hip[j]]])
                l_, p_ = orc.forward(forced[j][:, None], past_key_values=pkv_)
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
        assert int(lens_ref[-1][0]) == int(lens[-1][0]) and int(lens_ref[0][0]) == int(lens[0][0]), "KV lengths after the decode steps"
        del o, p_ref
        o32 = Oracle(cfg, sd, torch.float32)  # casts the bf16 tensors up: identical parameter values
        o32.force_keep_index = keep_hip  # the fp32 run must follow the same kept set to be a ground truth for these logits
        l_32, p_32 = o32.forward(ids, image_features=feats.cpu().float())
        truth = [l_32[0, -1]]
        for j in range(n_steps):
            l_32, p_32, f_ = step(o32, j, p_32)
            truth.append(l_32[0, -1])
            if f_:
                forced_steps["fp32"].append(j)
    ulp = 2.0**-7
    for j in range(n_steps + 1):  # EVERY step is compared (no early exit): boundary decisions were forced, not skipped
        e_hip = float((hip_logits[j] - truth[j]).abs().max())
        e_ref = float((ref_logits[j] - truth[j]).abs().max())
        assert e_hip <= 2.0 * e_ref + 2 * ulp * float(truth[j].abs().max()), f"step {j}: hip err {e_hip} vs reference err {e_ref}"
    print(f"full depth: kept set {'forced to the HIP set (near-tied boundary)' if forced_keep is not None else 'bit-exact'}; "
          f"eviction decisions forced at steps {forced_steps} of {n_steps}; evicted {n_steps - sum(dec_hip)} of {n_steps}")
