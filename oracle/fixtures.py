"""TEST INFRASTRUCTURE ONLY -- deterministic configs / weights / inputs shared by the oracle,
the golden-vector generator and the tests.  Nothing under dynamic_llava_amd/ imports this.

Weights are *not* committed: every fixture state-dict is regenerated from a seed with the CPU
torch generator (bit-stable for a given torch build; the GPU box runs this same image), in a
fixed key order.  Golden files under tests/golden/ store only inputs and reference outputs.
"""
from __future__ import annotations

import copy
import math
from types import SimpleNamespace

import torch

IMAGE_TOKEN_INDEX = -200  # llava/constants.py:8
IGNORE_INDEX = -100  # llava/constants.py:7

# llava/train/train_sparse.py:145-165 (SparseArguments defaults)
DEFAULT_SPARSE = dict(
    use_vision_predictor=True,
    vision_keep_rate=0.2,
    use_text_predictor=True,
    use_output_text_predictor=True,
    output_text_keep_rate=0.5,
    output_text_len_for_training=50,
    use_instruct_predictor=False,
    instruct_keep_rate=0.7,
    instruct_len_for_training=25,
    sparse_layer=2,
    d_model=512,
    nhead=8,
    dim_feedforward=2048,
    num_layers=2,
    mask_loss_weight=100.0,
)


def make_config(
    hidden_size=4096,
    intermediate_size=11008,
    num_hidden_layers=32,
    num_attention_heads=32,
    num_key_value_heads=None,
    vocab_size=32000,
    max_position_embeddings=4096,
    rms_norm_eps=1e-5,
    rope_theta=10000.0,
    mm_hidden_size=1024,
    clip=None,
    **sparse_overrides,
):
    sc = copy.deepcopy(DEFAULT_SPARSE)
    sc.update(sparse_overrides)
    return SimpleNamespace(
        hidden_size=hidden_size,
        intermediate_size=intermediate_size,
        num_hidden_layers=num_hidden_layers,
        num_attention_heads=num_attention_heads,
        num_key_value_heads=num_key_value_heads or num_attention_heads,
        vocab_size=vocab_size,
        max_position_embeddings=max_position_embeddings,
        rms_norm_eps=rms_norm_eps,
        rope_theta=rope_theta,
        mm_hidden_size=mm_hidden_size,
        mm_projector_type="mlp2x_gelu",
        mm_vision_select_layer=-2,
        mm_vision_select_feature="patch",
        clip=clip
        or dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14),
        sparse_config=sc,
    )


def tiny_config(**sparse_overrides):
    """H=256 (2 heads x 128), 4 layers, predictor d_model=128 (2 heads x 64), 36-token images."""
    kw = dict(d_model=128, nhead=2, dim_feedforward=256, num_layers=2)
    kw.update(sparse_overrides)
    return make_config(
        hidden_size=256,
        intermediate_size=512,
        num_hidden_layers=4,
        num_attention_heads=2,
        vocab_size=320,
        max_position_embeddings=2048,
        mm_hidden_size=64,
        clip=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=84, patch_size=14),
        **kw,
    )


def llava7b_config(num_hidden_layers=32, **sparse_overrides):
    return make_config(num_hidden_layers=num_hidden_layers, **sparse_overrides)


def llava13b_config(num_hidden_layers=40, **sparse_overrides):
    return make_config(
        hidden_size=5120, intermediate_size=13824, num_hidden_layers=num_hidden_layers, num_attention_heads=40, **sparse_overrides
    )


def n_image_tokens(cfg) -> int:
    return (cfg.clip["image_size"] // cfg.clip["patch_size"]) ** 2


# ---------------------------------------------------------------------------------------------
# state dict (same key names as the reference: dynamic_modeling_llama.py:1591-1631,
# dynamic_llava_arch.py:44-46, custom_transformer_layer.py:146-150,107-113)
# ---------------------------------------------------------------------------------------------
def decoder_param_shapes(cfg, with_predictors=True, with_projector=True):
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    d = H // cfg.num_attention_heads
    KV = cfg.num_key_value_heads * d
    sc = cfg.sparse_config
    D, F = sc["d_model"], sc["dim_feedforward"]
    out = [("model.embed_tokens.weight", (V, H), "w")]
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        out += [
            (p + "self_attn.q_proj.weight", (H, H), "w"),
            (p + "self_attn.k_proj.weight", (KV, H), "w"),
            (p + "self_attn.v_proj.weight", (KV, H), "w"),
            (p + "self_attn.o_proj.weight", (H, H), "w"),
            (p + "mlp.gate_proj.weight", (I, H), "w"),
            (p + "mlp.up_proj.weight", (I, H), "w"),
            (p + "mlp.down_proj.weight", (H, I), "w"),
            (p + "input_layernorm.weight", (H,), "g"),
            (p + "post_attention_layernorm.weight", (H,), "g"),
        ]
    out.append(("model.norm.weight", (H,), "g"))
    if with_predictors and sc["use_vision_predictor"]:
        p = "model.image_score_predictor."
        out += [
            (p + "down_mlp.0.weight", (H,), "g"),
            (p + "down_mlp.0.bias", (H,), "b"),
            (p + "down_mlp.1.weight", (D, H), "w"),
            (p + "down_mlp.1.bias", (D,), "b"),
        ]
        for j in range(sc["num_layers"]):
            q = p + f"transformer.{j}."
            out += [
                (q + "norm1.weight", (D,), "g"),
                (q + "norm1.bias", (D,), "b"),
                (q + "attn.qkv.weight", (3 * D, D), "wp"),
                (q + "attn.proj.weight", (D, D), "wp"),
                (q + "attn.proj.bias", (D,), "b"),
                (q + "norm2.weight", (D,), "g"),
                (q + "norm2.bias", (D,), "b"),
                (q + "mlp.fc1.weight", (F, D), "wp"),
                (q + "mlp.fc1.bias", (F,), "b"),
                (q + "mlp.fc2.weight", (D, F), "wp"),
                (q + "mlp.fc2.bias", (D,), "b"),
            ]
        out += [
            (p + "output_mlp.0.weight", (D // 2, D), "wp"),
            (p + "output_mlp.0.bias", (D // 2,), "b"),
            (p + "output_mlp.2.weight", (D // 4, D // 2), "wp"),
            (p + "output_mlp.2.bias", (D // 4,), "b"),
            (p + "output_mlp.4.weight", (2, D // 4), "wlast"),
            (p + "output_mlp.4.bias", (2,), "b"),
        ]
    if with_predictors and sc["use_text_predictor"]:
        names = []
        if sc["use_output_text_predictor"]:
            names.append("output_text_score_predictor")
        if sc["use_instruct_predictor"]:
            names.append("instruct_score_predictor")
        for nm in names:
            p = f"model.{nm}.output_mlp."
            out += [
                (p + "0.weight", (H,), "g"),
                (p + "0.bias", (H,), "b"),
                (p + "1.weight", (D, H), "w"),
                (p + "1.bias", (D,), "b"),
                (p + "3.weight", (D // 2, D), "wp"),
                (p + "3.bias", (D // 2,), "b"),
                (p + "5.weight", (D // 4, D // 2), "wp"),
                (p + "5.bias", (D // 4,), "b"),
                (p + "7.weight", (2, D // 4), "wlast"),
                (p + "7.bias", (2,), "b"),
            ]
    if with_projector:
        out += [
            ("model.mm_projector.0.weight", (H, cfg.mm_hidden_size), "wp"),
            ("model.mm_projector.0.bias", (H,), "b"),
            ("model.mm_projector.2.weight", (H, H), "w"),
            ("model.mm_projector.2.bias", (H,), "b"),
        ]
    out.append(("lm_head.weight", (V, H), "w"))
    return out


def make_state_dict(cfg, seed=0, dtype=torch.float32, predictor_gain=1.0, init="fanin", **kw):
    """Seeded weights.  Unlike HF `_init_weights` (dynamic_modeling_llama.py:1492-1501: N(0, 0.02),
    zero bias, unit norm gains) biases and gains are perturbed so that bias / gain bugs are visible.
    `init="hf"` uses N(0, 0.02) for the decoder matrices (the 7B-size benches), "fanin" (tests) 1/sqrt(fan_in).
    `predictor_gain` scales the last predictor layer ("trained-like" variant: SURVEY section 7 -- no score ties).
    Values are generated in fp32 and then cast, so the bf16 and fp32 models share parameters up to rounding."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, kind in decoder_param_shapes(cfg, **kw):
        if kind == "w" and init == "hf":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "w":  # fan-in scaling keeps q.k / sqrt(d) ~ N(0,1): attention is peaky, RoPE matters
            t = torch.randn(shape, generator=g) * (1.0 if name.endswith("embed_tokens.weight") else 1.0 / math.sqrt(shape[-1]))
        elif kind == "wp":  # small predictor / projector matrices: fan-in scaled so activations stay O(1)
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(shape[-1]))
        elif kind == "wlast":
            t = torch.randn(shape, generator=g) * (predictor_gain / math.sqrt(shape[-1]))
        elif kind == "g":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "b":
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            raise ValueError(kind)
        sd[name] = t.to(dtype)
    return sd


def make_clip_state_dict(clip_model, seed=1):
    """Re-initialise a transformers CLIPVisionModel in place from a seed (key order = state_dict order)."""
    g = torch.Generator().manual_seed(seed)
    sd = clip_model.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.is_floating_point():
            new[k] = v.clone()
        elif v.ndim == 1 and ("norm" in k or "layrnorm" in k) and k.endswith("weight"):
            new[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif v.ndim == 1:
            new[k] = 0.02 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel() if v.ndim > 1 else v.numel()
            new[k] = torch.randn(v.shape, generator=g) * (0.5 / math.sqrt(fan_in))
    clip_model.load_state_dict(new)
    return clip_model


def build_clip(cfg, seed=1, dtype=torch.float32):
    from transformers import CLIPVisionConfig, CLIPVisionModel

    c = cfg.clip
    ccfg = CLIPVisionConfig(
        hidden_size=c["hidden_size"],
        intermediate_size=c["intermediate_size"],
        num_hidden_layers=c["num_hidden_layers"],
        num_attention_heads=c["num_attention_heads"],
        image_size=c["image_size"],
        patch_size=c["patch_size"],
        projection_dim=c["hidden_size"],
    )
    m = CLIPVisionModel(ccfg)
    make_clip_state_dict(m, seed)
    return m.to(dtype).eval()


def make_images(cfg, batch, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(1000 + seed)
    s = cfg.clip["image_size"]
    return torch.randn((batch, 3, s, s), generator=g).to(dtype)


def make_prompt(cfg, n_sys, n_q, seed=0):
    """[BOS, sys..., <image>, question...] with ids in [3, vocab)."""
    g = torch.Generator().manual_seed(2000 + seed)
    ids = torch.randint(3, cfg.vocab_size, (n_sys + n_q,), generator=g)
    sys_part = torch.cat([torch.tensor([1]), ids[: n_sys - 1]]) if n_sys > 0 else ids[:0]
    return torch.cat([sys_part, torch.tensor([IMAGE_TOKEN_INDEX]), ids[n_sys:]]).long()


def make_forced_tokens(cfg, steps, batch, seed=0):
    """Teacher-forced decode inputs [steps, B] (the reference's long-text bench feeds label ids the same way)."""
    g = torch.Generator().manual_seed(3000 + seed)
    return torch.randint(3, cfg.vocab_size, (steps, batch), generator=g)


# ---- keep / evict decisions on the boundary (tests only) -------------------------------------------------------------------------------------
# DML:2385-2391 decides `logit[keep] > logit[evict]` on raw logits.  Two correct evaluations of the predictor in a 16-bit dtype differ by a few units in
# the last place OF THE LOGITS' MAGNITUDE (the gained final layer sums O(100) products of that size), so a pair whose gap is inside that band may fall
# either way; outside it a different decision is an error.  The band is stated in ulps of the larger logit, not as an absolute number, and the tests
# assert how many steps of a run needed it: a regression that flips many near-boundary decisions fails.
BOUNDARY_ULPS = {torch.bfloat16: 8.0, torch.float16: 8.0, torch.float32: 4096.0}  # fp32: 4096 ulp = 5e-4 relative, the literal 1e-3 logit budget at |logit| ~ 2
_ULP = {torch.bfloat16: 2.0**-7, torch.float16: 2.0**-10, torch.float32: 2.0**-23}
MAX_FORCED_DECISIONS = 2  # per compared row and run


def boundary_band(text_logit_pair, dtype) -> float:
    """Width of the band around zero inside which a keep / evict gap may legitimately change sign: BOUNDARY_ULPS ulps of max(|keep|, |evict|, 1)."""
    t = torch.as_tensor(text_logit_pair).float().abs().reshape(-1)
    return BOUNDARY_ULPS[dtype] * _ULP[dtype] * max(1.0, float(t.max()))


def decision_may_differ(pair_a, dtype_a, pair_b, dtype_b) -> bool:
    """True if at least one side's gap lies inside its own boundary band (each side in the dtype IT computed the logits in)."""
    ga = abs(float(torch.as_tensor(pair_a).float().reshape(-1)[0] - torch.as_tensor(pair_a).float().reshape(-1)[1]))
    gb = abs(float(torch.as_tensor(pair_b).float().reshape(-1)[0] - torch.as_tensor(pair_b).float().reshape(-1)[1]))
    return ga <= boundary_band(pair_a, dtype_a) or gb <= boundary_band(pair_b, dtype_b)
