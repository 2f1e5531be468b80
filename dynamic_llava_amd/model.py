"""Host side of the MI355X-native Dynamic-LLaVA hot path: same API surface as the reference's
`DynamicLlavaLlamaForCausalLM` (llava/model/language_model/dynamic_llava_llama.py:50-169), different engine.

What stays on PyTorch-ROCm (per BASELINE.json north_star): CLIP ViT-L/14-336 + mm_projector, the embedding
lookup and the dense decoder GEMMs (fused QKV, O, fused gate|up, down, lm_head -> hipBLASLt/MFMA).
Everything else on the path is a hand-written HIP kernel behind the C ABI (include/dynllava.h):
RMSNorm(+residual), RoPE+KV append, varlen prefill attention, ragged split-KV decode attention, SiLU*up,
vision predictor, top-k select, token compaction, text predictor + eviction decision, greedy/advance.

Design (not a translation of the reference's op sequence):
  * activations are PACKED varlen [total_tokens, H] + cu_seqlens, never padded [B, N, H];
  * after layer `sparse_layer` the packed batch is physically compacted (k image tokens per row survive);
  * KV lives in a pre-allocated slab with device-side per-row lengths (cache.py); eviction = "do not
    advance the length"; the decode step has zero host syncs and is captured in a hipGraph;
  * all prefill shapes are host-known (k is constant per row), so prefill has no device->host sync either
    apart from reading `input_ids` (which the harness hands over on the GPU).
"""
from __future__ import annotations

import copy
import math
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops as ops
from .cache import KVSlabCache
from .config import IGNORE_INDEX, IMAGE_TOKEN_INDEX, DynamicLlavaConfig

USER_IDS = [11889, 29901]  # "USER:" -- llava/model/dynamic_llava_arch.py:36


@dataclass
class CausalLMOutputWithPast:
    """Mirror of transformers.modeling_outputs.CausalLMOutputWithPast (fields the harness reads)."""

    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[KVSlabCache] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.logits, self.past_key_values) if v is not None)[i]


# ------------------------------------------------------------------------------------------------
# parameter containers: same module tree / state-dict keys as the reference, so checkpoints load
# ------------------------------------------------------------------------------------------------
class _Attn(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H, d = cfg.hidden_size, cfg.head_dim
        self.q_proj = nn.Linear(H, cfg.num_attention_heads * d, bias=False)
        self.k_proj = nn.Linear(H, cfg.num_key_value_heads * d, bias=False)
        self.v_proj = nn.Linear(H, cfg.num_key_value_heads * d, bias=False)
        self.o_proj = nn.Linear(cfg.num_attention_heads * d, H, bias=False)


class _Mlp(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.gate_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.up_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)


class _Norm(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))


class DynamicLlamaDecoderLayer(nn.Module):  # dynamic_modeling_llama.py:1221-1234
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = _Attn(cfg)
        self.mlp = _Mlp(cfg)
        self.input_layernorm = _Norm(cfg.hidden_size)
        self.post_attention_layernorm = _Norm(cfg.hidden_size)
        self.w_qkv = None  # fused [nH*d + 2*nKV*d, H]; q/k/v_proj.weight become views of it (no extra memory)
        self.w_gu = None  # fused [2*I, H]
        # round 5: second copies of q|k|v, gate|up and down_proj in matrix-core operand order for dl_linear_packed (the prefill GEMMs at <= 256 packed
        # rows, decode batches 4..32); gate|up with gate / up tiles interleaved for the SiLU * up epilogue.  +371 MB per 7B layer (q|k|v 101 + gate|up 180 +
        # down 90: 11.9 GB over 32 layers, 23 GB at 13B) of 288 GB; the state dict is untouched; model.operand_copy_bytes() reports them.
        self.wp_qkv = None
        self.wp_gu = None
        self.wp_down = None

    def pack(self, operand_copies: bool = False):
        a, m = self.self_attn, self.mlp
        self.w_qkv = torch.cat([a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data], dim=0).contiguous()
        nq, nk = a.q_proj.weight.shape[0], a.k_proj.weight.shape[0]
        a.q_proj.weight.data = self.w_qkv[:nq]
        a.k_proj.weight.data = self.w_qkv[nq : nq + nk]
        a.v_proj.weight.data = self.w_qkv[nq + nk :]
        self.w_gu = torch.cat([m.gate_proj.weight.data, m.up_proj.weight.data], dim=0).contiguous()
        I = m.gate_proj.weight.shape[0]
        m.gate_proj.weight.data = self.w_gu[:I]
        m.up_proj.weight.data = self.w_gu[I:]
        self.wp_qkv = self.wp_gu = self.wp_down = None
        if operand_copies and self.w_qkv.dtype in (torch.bfloat16, torch.float16) and self.w_qkv.shape[1] % 64 == 0 and self.w_qkv.shape[0] % 16 == 0 and I % 16 == 0:
            self.wp_qkv = ops.pack_weight_tiles(self.w_qkv)
            self.wp_gu = ops.pack_weight_tiles(self.w_gu, gate_up_pairs=True)
            if I % 64 == 0:  # down_proj reads the SiLU * up epilogue's fragment-order output and leaves fp32 partial sums for the residual-add / RMSNorm launch
                self.wp_down = ops.pack_weight_tiles(m.down_proj.weight.data.contiguous())


class _TransformerBlock(nn.Module):  # custom_transformer_layer.py:276-318 (parameters only)
    def __init__(self, dim, ff):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, ff)
        self.mlp.fc2 = nn.Linear(ff, dim)


class VisionPredictor(nn.Module):
    """dynamic_modeling_llama.py:1308-1359.  forward(x [B,n,H], image_policy) -> logits [B,n,2], computed by
    the HIP pipeline (dl_vision_predictor).  Hookable like the reference module (visualize.py:74)."""

    def __init__(self, input_dim=4096, d_model=512, nhead=8, dim_feedforward=2048, num_layers=2):
        super().__init__()
        self.input_dim, self.d_model, self.nhead, self.dim_feedforward, self.num_layers = input_dim, d_model, nhead, dim_feedforward, num_layers
        self.down_mlp = nn.Sequential(nn.LayerNorm(input_dim), nn.Linear(input_dim, d_model), nn.GELU())
        self.transformer = nn.Sequential(*[_TransformerBlock(d_model, dim_feedforward) for _ in range(num_layers)])
        self.output_mlp = nn.Sequential(
            nn.Linear(d_model, d_model // 2), nn.GELU(), nn.Linear(d_model // 2, d_model // 4), nn.GELU(), nn.Linear(d_model // 4, 2)
        )
        self._w = None
        self.last_score = None

    def _weights(self):
        key = self.down_mlp[1].weight.data_ptr()
        if self._w is None or self._w[0] != key:
            w = ops.VpWeights()
            dp = lambda t: t.data_ptr()
            w.ln_w, w.ln_b = dp(self.down_mlp[0].weight), dp(self.down_mlp[0].bias)
            w.down_w, w.down_b = dp(self.down_mlp[1].weight), dp(self.down_mlp[1].bias)
            w.out0_w, w.out0_b = dp(self.output_mlp[0].weight), dp(self.output_mlp[0].bias)
            w.out2_w, w.out2_b = dp(self.output_mlp[2].weight), dp(self.output_mlp[2].bias)
            w.out4_w, w.out4_b = dp(self.output_mlp[4].weight), dp(self.output_mlp[4].bias)
            w.num_layers = self.num_layers
            for j, blk in enumerate(self.transformer):
                b = w.blocks[j]
                b.norm1_w, b.norm1_b = dp(blk.norm1.weight), dp(blk.norm1.bias)
                b.qkv_w = dp(blk.attn.qkv.weight)
                b.proj_w, b.proj_b = dp(blk.attn.proj.weight), dp(blk.attn.proj.bias)
                b.norm2_w, b.norm2_b = dp(blk.norm2.weight), dp(blk.norm2.bias)
                b.fc1_w, b.fc1_b = dp(blk.mlp.fc1.weight), dp(blk.mlp.fc1.bias)
                b.fc2_w, b.fc2_b = dp(blk.mlp.fc2.weight), dp(blk.mlp.fc2.bias)
            self._w = (key, w)
        return self._w[1]

    def score_packed(self, hidden, cu_seqlens, img_start, n_img):
        """packed hidden [total,H] -> (logits [B,n,2], score [B,n]); image rows gathered inside the LN kernel."""
        return ops.vision_predictor(hidden, cu_seqlens, img_start, n_img, self._weights(), self.d_model, self.nhead, self.dim_feedforward)

    def forward(self, x, image_policy=None):
        B, n, H = x.shape
        x = x.contiguous().view(B * n, H)
        cu = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=x.device)
        start = torch.zeros(B, dtype=torch.int32, device=x.device)
        logits, self.last_score = self.score_packed(x, cu, start, n)
        return logits


class TextPredictor(nn.Module):
    """dynamic_modeling_llama.py:1362-1387 (parameters) + the decision of DML:2388-2391 (dl_text_predictor_decide)."""

    def __init__(self, input_dim=4096, d_model=512, **_):
        super().__init__()
        self.input_dim, self.d_model = input_dim, d_model
        self.output_mlp = nn.Sequential(
            nn.LayerNorm(input_dim), nn.Linear(input_dim, d_model), nn.GELU(), nn.Linear(d_model, d_model // 2), nn.GELU(),
            nn.Linear(d_model // 2, d_model // 4), nn.GELU(), nn.Linear(d_model // 4, 2),
        )
        self._w = None

    def _weights(self):
        key = self.output_mlp[1].weight.data_ptr()
        if self._w is None or self._w[0] != key:
            w = ops.TpWeights()
            m = self.output_mlp
            w.ln_w, w.ln_b = m[0].weight.data_ptr(), m[0].bias.data_ptr()
            w.l1_w, w.l1_b = m[1].weight.data_ptr(), m[1].bias.data_ptr()
            w.l3_w, w.l3_b = m[3].weight.data_ptr(), m[3].bias.data_ptr()
            w.l5_w, w.l5_b = m[5].weight.data_ptr(), m[5].bias.data_ptr()
            w.l7_w, w.l7_b = m[7].weight.data_ptr(), m[7].bias.data_ptr()
            self._w = (key, w)
        return self._w[1]

    def decide(self, x, workspace, logits_out, decision):
        return ops.text_predictor_decide(x, self._weights(), self.d_model, workspace, logits_out, decision)

    def forward(self, x):
        """x [..., H] -> logits [..., 2] (fp32 values of the model-dtype logits)."""
        shp = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        B = x2.shape[0]
        ws = ops.text_predictor_workspace(B, self.d_model, x.device)
        lg = torch.empty((B, 2), dtype=torch.float32, device=x.device)
        dec = torch.empty(B, dtype=torch.int32, device=x.device)
        self.decide(x2, ws, lg, dec)
        return lg.to(x.dtype).reshape(*shp, 2)


class CLIPVisionTower(nn.Module):
    """llava/model/multimodal_encoder/clip_encoder.py:7-102.  The HF CLIPVisionModel is the parameter container (state-dict keys
    unchanged); forward() runs its encoder on the library GEMMs + this package's HIP kernels."""

    def __init__(self, cfg: DynamicLlavaConfig):
        super().__init__()
        from transformers import CLIPVisionConfig, CLIPVisionModel

        self.select_layer = cfg.mm_vision_select_layer
        self.select_feature = cfg.mm_vision_select_feature
        self.vision_tower_name = cfg.mm_vision_tower
        c = cfg.clip
        self.vision_tower = CLIPVisionModel(
            CLIPVisionConfig(
                hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                num_attention_heads=c["num_attention_heads"], image_size=c["image_size"], patch_size=c["patch_size"], projection_dim=c["hidden_size"],
            )
        )
        self.vision_tower.requires_grad_(False)
        self.is_loaded = True
        # round 6 knobs: the tower's projections on dl_linear_tiles (False: the library GEMMs), up to how many images per call, k ranges of out_proj / fc2
        self.tiles_gemm = os.environ.get("DL_CLIP_TILES", "1") != "0"
        self.tiles_max_batch = 4
        self.tiles_ksplit_out, self.tiles_ksplit_fc2 = 2, 4
        self._patch_embed_as_gemm()

    def _patch_embed_as_gemm(self):
        """The ViT patch embedding is a stride-14 14x14 conv == one GEMM over unfolded patches.  MIOpen serves it with a
        ~330 us naive fallback kernel in bf16; the same weights through F.linear take ~20 us.  Still plain PyTorch."""
        import types

        conv = next(m for n, m in self.vision_tower.named_modules() if n.endswith("patch_embedding"))  # module path differs across HF versions
        ps = conv.kernel_size[0]

        def gemm_forward(mod, x):
            B, C, Hh, Ww = x.shape
            gh, gw = Hh // ps, Ww // ps
            patches = x.reshape(B, C, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ps * ps)
            y = F.linear(patches, mod.weight.reshape(mod.weight.shape[0], -1), mod.bias)
            return y.transpose(1, 2).reshape(B, -1, gh, gw)

        conv.forward = types.MethodType(gemm_forward, conv)

    def pack(self):
        """Fuse q|k|v of every encoder layer into one [3C, C] weight (+bias) for a single projection GEMM, and (16-bit dtypes) keep the four projections of
        every layer a second time in matrix-core operand order for dl_linear_tiles (+0.6 GB for ViT-L/14-336 in bf16; `tiles_bytes` says how much: the
        harness counterparts report it).  Call after the weights are loaded / cast (finalize() does)."""
        vm = next(m for n, m in self.vision_tower.named_modules() if hasattr(m, "encoder") and hasattr(m, "embeddings"))
        self._vm = [vm]  # in a list: not a registered submodule (the parameter tree / state-dict keys stay HF's)
        self._qkv = []
        self._tiles = []  # per layer: (wp_qkv, wp_out, wp_fc1, wp_fc2) or None
        self.tiles_bytes = 0
        for l in vm.encoder.layers:
            a = l.self_attn
            wq = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).contiguous()
            self._qkv.append((wq, torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).contiguous()))
            ws = (wq, a.out_proj.weight, l.mlp.fc1.weight, l.mlp.fc2.weight)
            if self.tiles_gemm and wq.is_cuda and all(ops.linear_tiles_ok(1, w.shape[0], w.shape[1], w.dtype) for w in ws):
                self._tiles.append(tuple(ops.pack_weight_tiles(w.detach().contiguous()) for w in ws))
                self.tiles_bytes += sum(t.numel() * t.element_size() for t in self._tiles[-1])
            else:
                self._tiles.append(None)
        self._tiles_src = [(w.data_ptr(), w._version) for l in vm.encoder.layers for w in (l.self_attn.q_proj.weight, l.self_attn.out_proj.weight, l.mlp.fc1.weight, l.mlp.fc2.weight)]
        self._cu = {}
        return self

    def _tiles_fresh(self):
        """The operand-order copies are detached: replacing / editing a weight after pack() must not leave the tiled path on the old values."""
        vm = self._vm[0]
        now = [(w.data_ptr(), w._version) for l in vm.encoder.layers for w in (l.self_attn.q_proj.weight, l.self_attn.out_proj.weight, l.mlp.fc1.weight, l.mlp.fc2.weight)]
        return now == self._tiles_src

    def _n_layers_needed(self):
        """hidden_states[k] is the stream after k encoder layers; select_layer = -2 needs L-1 of the L layers (HF computes all L
        and throws the last one away)."""
        L = len(self._vm[0].encoder.layers)
        k = self.select_layer if self.select_layer >= 0 else L + 1 + self.select_layer
        if not 0 <= k <= L:
            raise ValueError(f"mm_vision_select_layer={self.select_layer} out of range for {L} layers")
        return k

    @torch.no_grad()
    def forward(self, images):
        """clip_encoder.py:53-71 (`feature_select(vision_tower(images, output_hidden_states=True))`).  The encoder runs packed
        ([B*T, C] rows, cu_seqlens) on: hipBLASLt for the plain GEMMs (bias fused), dl_layernorm / dl_add_layernorm (residual add +
        next LayerNorm in one pass), dl_attn_prefill (non-causal MFMA flash attention, head_dim 64) and dl_quick_gelu -- 8 launches
        per layer instead of the ~18 of the eager module, each rounding to the model dtype where the eager module does."""
        if getattr(self, "_vm", None) is None:
            self.pack()
        vm = self._vm[0]
        cfgv = vm.config if hasattr(vm, "config") else self.config
        if cfgv.hidden_act != "quick_gelu":
            raise ops.HipOpsError(f"CLIP hidden_act={cfgv.hidden_act!r}: only quick_gelu (OpenAI CLIP) is implemented")
        x = images.to(device=self.device, dtype=self.dtype)
        B = x.shape[0]
        emb = vm.embeddings(x)  # patch GEMM + class token + position embedding (once per image; plain torch)
        T, C = emb.shape[1], emb.shape[2]
        nH = cfgv.num_attention_heads
        d = C // nH
        eps = cfgv.layer_norm_eps
        pre = getattr(vm, "pre_layrnorm", None) or getattr(vm, "pre_layernorm")
        h = ops.layernorm(emb.reshape(B * T, C).contiguous(), pre.weight, pre.bias, eps)
        cu = self._cu.get(B)
        if cu is None:
            cu = self._cu[B] = (torch.arange(B + 1, device=h.device, dtype=torch.int32) * T).contiguous()
        if not (h.dtype == torch.float32 or d in (32, 64, 128)):
            raise ops.HipOpsError(f"CLIP head_dim={d}: dl_attn_prefill tiles head dims 32 / 64 / 128 in 16-bit dtypes (no torch fallback exists)")
        layers = vm.encoder.layers[: self._n_layers_needed()]
        if not self._tiles_fresh():
            self.pack()
        if len(layers) and all(self._tiles[i] is not None for i in range(len(layers))) and B <= self.tiles_max_batch:
            self._encoder_tiles(h, layers, cu, B, T, C, nH, d, eps)
        else:
            self._encoder_library(h, layers, cu, B, T, C, nH, d, eps)
        f = h.view(B, T, C)
        if self.select_feature == "patch":
            f = f[:, 1:]
        elif self.select_feature != "cls_patch":
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        return f.to(images.dtype) if images.is_floating_point() else f

    def _encoder_library(self, h, layers, cu, B, T, C, nH, d, eps):
        """Library GEMMs (bias fused) + this package's glue kernels: fp32 models, batches past `tiles_max_batch` images (the library's large-tile kernels
        are MFMA-bound there), towers whose shapes dl_linear_tiles does not take."""
        xn = ops.layernorm(h, layers[0].layer_norm1.weight, layers[0].layer_norm1.bias, eps) if len(layers) else None
        for i, l in enumerate(layers):
            wq, bq = self._qkv[i]
            qkv = F.linear(xn, wq, bq)
            attn = torch.empty((B * T, C), dtype=h.dtype, device=h.device)
            ops.attn_prefill(qkv[:, :C], qkv[:, C : 2 * C], qkv[:, 2 * C :], attn, cu, T, nH, nH, d, causal=False)
            y = F.linear(attn, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias)
            xn = ops.add_layernorm(h, y, l.layer_norm2.weight, l.layer_norm2.bias, eps)
            g = ops.quick_gelu(F.linear(xn, l.mlp.fc1.weight, l.mlp.fc1.bias))
            y = F.linear(g, l.mlp.fc2.weight, l.mlp.fc2.bias)
            if i + 1 < len(layers):
                nl = layers[i + 1]
                xn = ops.add_layernorm(h, y, nl.layer_norm1.weight, nl.layer_norm1.bias, eps)
            else:
                ops.add_layernorm(h, y)

    def _encoder_tiles(self, h, layers, cu, B, T, C, nH, d, eps):
        """Round 6: every projection on dl_linear_tiles (own MFMA GEMM on operand-order weight copies; 7 launches per layer).  Activations between the
        launches travel in the GEMM's fragment order wherever a producer can write it: LN -> q|k|v, LN -> fc1, fc1 (+ QuickGELU in the epilogue) -> fc2;
        out_proj and fc2 leave fp32 k-range partial sums that the residual-add / LayerNorm launch adds in order (with the Linear's bias, one rounding:
        F.linear's value)."""
        M = B * T
        I = layers[0].mlp.fc1.weight.shape[0]
        xn = ops.layernorm_rows(h, layers[0].layer_norm1.weight, layers[0].layer_norm1.bias, eps, packed=True)
        attn = torch.empty((M, C), dtype=h.dtype, device=h.device)
        qkv = torch.empty((M, 3 * C), dtype=h.dtype, device=h.device)
        ks_o, ks_2 = max(1, min(self.tiles_ksplit_out, C // 64)), max(1, min(self.tiles_ksplit_fc2, I // 64))  # (tiny test towers: K = 64 is one step)
        for i, l in enumerate(layers):
            wp_qkv, wp_out, wp_fc1, wp_fc2 = self._tiles[i]
            ops.linear_tiles(xn, wp_qkv, 3 * C, bias=self._qkv[i][1], out=qkv, x_packed_mk=(M, C))
            ops.attn_prefill(qkv[:, :C], qkv[:, C : 2 * C], qkv[:, 2 * C :], attn, cu, T, nH, nH, d, causal=False)
            parts = ops.linear_tiles(attn, wp_out, C, epilogue=ops.LT_PARTS, k_split=ks_o)
            xn = ops.add_layernorm_parts(h, parts, l.self_attn.out_proj.bias, l.layer_norm2.weight, l.layer_norm2.bias, eps, packed=True)
            g = ops.linear_tiles(xn, wp_fc1, I, bias=l.mlp.fc1.bias, epilogue=ops.LT_QGELU, x_packed_mk=(M, C), y_packed=True)
            parts = ops.linear_tiles(g, wp_fc2, C, epilogue=ops.LT_PARTS, k_split=ks_2, x_packed_mk=(M, I))
            if i + 1 < len(layers):
                nl = layers[i + 1]
                xn = ops.add_layernorm_parts(h, parts, l.mlp.fc2.bias, nl.layer_norm1.weight, nl.layer_norm1.bias, eps, packed=True)
            else:
                ops.add_layernorm_parts(h, parts, l.mlp.fc2.bias)

    @torch.no_grad()
    def forward_eager(self, images):
        """The HF module as the reference runs it (tests compare the packed path against this)."""
        out = self.vision_tower(images.to(device=self.device, dtype=self.dtype), output_hidden_states=True)
        f = out.hidden_states[self.select_layer]
        return f[:, 1:] if self.select_feature == "patch" else f

    @property
    def dtype(self):
        return next(self.vision_tower.parameters()).dtype

    @property
    def device(self):
        return next(self.vision_tower.parameters()).device

    @property
    def config(self):
        return self.vision_tower.config

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2


class DynamicLlavaLlamaModel(nn.Module):
    """Parameter tree of dynamic_modeling_llama.py:1586-1647 + dynamic_llava_arch.py:41-51."""

    def __init__(self, cfg: DynamicLlavaConfig, with_vision_tower=True):
        super().__init__()
        self.config = cfg
        sc = cfg.sparse_config
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([DynamicLlamaDecoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.norm = _Norm(cfg.hidden_size)
        kw = dict(input_dim=cfg.hidden_size, d_model=sc["d_model"], nhead=sc["nhead"], dim_feedforward=sc["dim_feedforward"], num_layers=sc["num_layers"])
        if sc["use_vision_predictor"]:
            self.image_score_predictor = VisionPredictor(**kw)
        if sc["use_text_predictor"]:
            if sc["use_output_text_predictor"]:
                self.output_text_score_predictor = TextPredictor(**kw)
            if sc["use_instruct_predictor"]:
                self.instruct_score_predictor = TextPredictor(**kw)
        if with_vision_tower:
            self.vision_tower = CLIPVisionTower(cfg)
        if cfg.mm_projector_type != "mlp2x_gelu":
            raise NotImplementedError("only the LLaVA-1.5 mlp2x_gelu projector is built (multimodal_projector/builder.py:172-179)")
        self.mm_projector = nn.Sequential(nn.Linear(cfg.mm_hidden_size, cfg.hidden_size), nn.GELU(), nn.Linear(cfg.hidden_size, cfg.hidden_size))
        self.answer_indice = None  # dynamic_modeling_llama.py:1644 -- state of the no-KV-cache decode mode (never reset by the reference)

    def get_vision_tower(self):
        return getattr(self, "vision_tower", None)


# ------------------------------------------------------------------------------------------------
# decode-step state (persistent device buffers: stable pointers for the hipGraph)
# ------------------------------------------------------------------------------------------------
class _DecodeState:
    def __init__(self, model, B, device, dtype, out_cap):
        cfg = model.config
        H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        nH, nKV, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.B = B
        self.cur_ids = torch.zeros(B, dtype=torch.int64, device=device)
        self.out_ids = torch.zeros((B, max(out_cap, 1)), dtype=torch.int64, device=device)
        self.step = torch.zeros(B, dtype=torch.int32, device=device)
        self.finished = torch.zeros(B, dtype=torch.int32, device=device)
        self.decision = torch.ones(B, dtype=torch.int32, device=device)
        self.tp_logits = torch.zeros((B, 2), dtype=torch.float32, device=device)
        self.tp_ws = ops.text_predictor_workspace(B, cfg.sparse_config["d_model"], device)
        self.tp_x = torch.empty((B, H), dtype=dtype, device=device)  # snapshot of the hidden state entering layer `sparse_layer`
        self.tp_stream = torch.cuda.Stream(device=device)  # the predictor runs beside layers >= sparse_layer (graph fork/join)
        self.cu = torch.arange(0, B + 1, dtype=torch.int32, device=device)
        self.h = torch.empty((B, H), dtype=dtype, device=device)
        self.h2 = torch.empty((B, H), dtype=dtype, device=device)  # residual ping-pong partner (dl_gemv ADDNORM)
        self.x = torch.empty((B, H), dtype=dtype, device=device)
        self.qkv = torch.empty((B, (nH + 2 * nKV) * d), dtype=dtype, device=device)
        self.o = torch.empty((B, H), dtype=dtype, device=device)
        self.gu = torch.empty((B, I), dtype=dtype, device=device)  # act = silu(gate)*up, produced by the gate|up GEMV epilogue
        self.dn = torch.empty((B, H), dtype=dtype, device=device)
        # weight-streaming GEMV path for small decode batches (else torch/hipBLASLt GEMMs)
        self.use_gemv = B <= min(model.gemv_max_decode_batch, ops.gemv_max_batch(I, dtype), ops.gemv_max_batch(H, dtype))
        self.attn = torch.empty((B, nH * d), dtype=dtype, device=device)
        self.act = torch.empty((B, I), dtype=dtype, device=device)
        self.logits = torch.empty((B, V), dtype=dtype, device=device)
        # split-KV: enough workgroups to cover the chip (256 CUs) without drowning in partials
        self.n_splits = max(1, min(32, 1024 // max(1, B * nH)))
        self.attn_ws = ops.attn_decode_workspace(B, nH, d, 32, device)
        # decode batches past the GEMV range: dl_gemm_smallm (weights streamed into the matrix cores) up to smallm_max_decode_batch rows
        self.use_smallm = (not self.use_gemv) and B <= model.smallm_max_decode_batch and all(
            ops.gemm_smallm_ok(B, n, k, dtype) for n, k in (((nH + 2 * nKV) * d, H), (H, nH * d), (2 * I, H), (H, I), (V, H))
        )
        self.lin_ws = torch.empty(8 * B * max(2 * I, V), dtype=torch.float32, device=device) if self.use_smallm else None
        # round 5: decode batches of packed_decode_mlp_min_batch..32 rows (configs[2] / [3]: 32) run their MLP on dl_linear_packed -- gate|up with the SiLU * up
        # epilogue writing `act` in fragment order, down_proj leaving 4 k ranges of fp32 partial sums for the residual-add / RMSNorm launch
        # (tools/bench_linear_packed.py, M = 32: 38.4 vs 44.8 us and 25.5 vs 31.7 us against the library) -- whatever q|k|v and o_proj run on
        l0 = model.model.layers[0]
        self.use_lp_mlp = (not self.use_gemv and model.packed_decode_mlp_min_batch <= B <= 32 and model.packed_decode_mlp and getattr(l0, "wp_gu", None) is not None
                           and getattr(l0, "wp_down", None) is not None)
        if self.use_lp_mlp:
            n_el = lambda cols: int(ops.lib().dl_packed_x_bytes(B, cols)) // 2
            self.x_pk = torch.empty(n_el(H), dtype=dtype, device=device)
            self.act_pk = torch.empty(n_el(I), dtype=dtype, device=device)
            self.lp_parts = torch.empty(4 * B * H, dtype=torch.float32, device=device)
        self.graphs = {}  # captured decode steps, keyed by (slab, split factors, ...): see _run_decode_steps
        # dl_gemv_gu_tp's granules (batch 1; the predictor's stage 1 stages the row in LDS: H <= 5120)
        tpm = getattr(model.model, "output_text_score_predictor", None)
        self.tp_gran = ops.gemv_gu_tp_workspace(tpm.d_model, device) if (B == 1 and tpm is not None and dtype in (torch.bfloat16, torch.float16) and H <= 5120 and H % 8 == 0 and tpm.d_model % 32 == 0) else None
        # dl_gemv_qkv_attn's granules (batch 1, 16-bit dtypes at the decoder widths the kernel takes)
        self.qa_gran = ops.gemv_qkv_attn_workspace(nH, nKV, d, device) if (B == 1 and dtype in (torch.bfloat16, torch.float16) and d in (64, 128) and H * 2 <= 48 * 1024) else None
        self.blk_err = torch.zeros(1, dtype=torch.int32, device=device)
        # generate(): ring of pinned host words [lens (2 x B) | finished (B)] + events -- the decode loop observes the evicted lengths and the
        # EOS flags with non-blocking copies and reads them one chunk of steps late (the launch queue never drains)
        self.obs_host = torch.empty((4, 3 * B), dtype=torch.int32).pin_memory()
        self.obs_ev = [torch.cuda.Event() for _ in range(4)]
        self.n_cu = torch.cuda.get_device_properties(device).multi_processor_count


class DynamicLlavaLlamaForCausalLM(nn.Module):
    """Drop-in for the reference class of the same name (dynamic_llava_llama.py:50-169)."""

    def __init__(self, config: DynamicLlavaConfig, with_vision_tower=True):
        super().__init__()
        self.config = config
        self.model = DynamicLlavaLlamaModel(config, with_vision_tower)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self._packed = False
        self._rope = None
        self._dstate = None
        self._prefill_graphs = {}
        self._prefill_pool = None
        # captured prefill shapes kept (LRU).  288 GB of HBM: the graphs share one memory pool, an entry costs its static inputs (~0.7 MB of
        # pixels) and the graph object
        self.max_prefill_graphs = 64
        self.prefill_width_bucket = int(os.environ.get("DL_WIDTH_BUCKET", "16"))  # see _width_bucket
        self.use_hip_graph = True  # (DL_USE_HIP_GRAPH=0 in the environment forces every launch eager, whatever is assigned here: see the property)
        self.attn_inkernel_combine = True  # decode attention: split 0's workgroup merges the split-KV partials inside the launch (no combine launch)
        self.device_prompt_layout = True  # generate(): un-padded one-image-per-row prompts are laid out by a device kernel (no device->host copy)
        self.tp_side_stream = False  # run the text predictor as a parallel graph branch (measured slower: see DESIGN.md)
        self.gemv_max_decode_batch = 3  # B <= this: decode GEMMs run as hand-written weight-streaming GEMVs (dl_gemv)
        # B <= this (and past the GEMV range): dl_gemm_smallm; larger batches use the library GEMM.  tools/bench_decode_batch.py: 3.87 / 3.95 /
        # 4.24 ms per step at B = 4 / 8 / 16 against 5.1-5.3 on the library; a wash at 20-24 (4.64 / 4.80 vs 4.66 / 4.75) where the hand-written
        # path is kept for being deterministic and batch-invariant; 4 % behind at 32 (5.09 vs 4.89)
        self.smallm_max_decode_batch = 32
        self.fuse_qkv_attn = os.environ.get("DL_FUSE_QKV_ATTN", "1") == "1"
        self.fuse_gu_tp = os.environ.get("DL_FUSE_GU_TP", "1") == "1"
        # attention workgroups per head inside the fused launch: up to this many, one per 128 keys of the scheduled bound (DL_QA_SPLITS=1: always one)
        self.fused_attn_max_splits = max(1, min(4, int(os.environ.get("DL_QA_SPLITS", "4"))))  # the kernel takes 1..4 (kQaMaxSplits)
        self.gu_grid_cap = int(os.environ.get("DL_GU_GRID", "0"))  # workgroups of the batch-1 gate|up launch (0: the kernel's default, 1024)
        self.qkv_attn_grid_cap = int(os.environ.get("DL_QA_GRID", "0"))  # workgroups of the fused q|k|v + attention launch (0: the kernel's default)
        # o_proj of the post-compaction prefill layers (<= 192 rows) on dl_linear_splitk like down_proj: 17.3 vs 18.4-22 us per layer, prefill
        # 9.35 -> 9.17 ms (A/B on one box); DL_SPLITK_O=0 restores the library GEMM
        self.splitk_o_proj = os.environ.get("DL_SPLITK_O", "1") == "1"
        # round 5: q|k|v and gate|up (+ SiLU * up) of prefill layers with <= 256 packed rows on dl_linear_packed (operand-order weight copies made by
        # finalize(), activations handed over in fragment order by the norm launch that produces them).  tools/bench_linear_packed.py, M = 170:
        # q|k|v 36.2 us (two k ranges per unit set) vs the library's 41.4, gate|up + SiLU * up 57.2 vs 64.1.  DL_PACKED_GEMM=0: library GEMMs.
        self.packed_prefill_gemm = os.environ.get("DL_PACKED_GEMM", "1") == "1"
        self.packed_down_proj = os.environ.get("DL_PACKED_DOWN", "1") == "1"  # down_proj too (partial sums for dl_add_rmsnorm_parts): A/B knob
        # batched decode (4..32 rows), tools/bench_decode_batch.py: the MLP on dl_linear_packed from 4 rows on (B = 16: 4.05 -> 3.79 ms per step, 24: 4.63 -> 4.03), q|k|v too from 16 rows
        # on (24: 4.05 -> 3.97, 32: 4.21 -> 4.11); o_proj stays on dl_gemm_smallm's partial sums up to 32 rows (against the library GEMM + add: 32 rows 4.20 -> 4.11)
        self.packed_decode_qkv_min_batch = int(os.environ.get("DL_PACKED_DECODE_QKV_MIN_B", "16"))
        self.packed_decode_mlp_min_batch = int(os.environ.get("DL_PACKED_DECODE_MLP_MIN_B", "4"))
        self.packed_qkv_parts = os.environ.get("DL_PACKED_QKV_PARTS", "1") == "1"  # prefill q|k|v: partial sums added by the RoPE / KV-append launch instead of the in-launch hand-over
        self.packed_decode_mlp = os.environ.get("DL_PACKED_DECODE_MLP", "1") == "1"  # decode batches 4..32: gate|up + SiLU * up and down_proj on dl_linear_packed
        self._lp_ws = None   # hand-over workspace of the k-split launches (zeroed once; the kernel leaves its flag words zero)
        self._lp_err = None  # bit 3: a reducing wave of dl_linear_packed gave up waiting
        # split-K slices of the two WIDE projections (q|k|v, gate|up: 768 / 1376 sixteen-neuron wave tiles without any split) on dl_gemm_smallm; 0 = the
        # kernel's own choice (8 / 4).  Round 4 measured the review's proposal -- fewer slices, so the consumers re-read fewer fp32 partial slabs --
        # and it LOSES at every batch: 2 slices 4.09 / 4.39 / 4.65 / 4.94 ms per step at B = 8 / 16 / 24 / 32 against 3.71 / 4.04 / 4.60 / 4.87 (library
        # 5.03 / 5.20 / 4.64 / 4.80): many short weight streams beat few long ones by more than the partial traffic costs (tools/bench_decode_batch.py)
        self.smallm_wide_slices = int(os.environ.get("DL_SMALLM_WIDE_SLICES", "0"))
        if os.environ.get("DL_SMALLM_MAX_B"):  # tuning experiments only
            self.smallm_max_decode_batch = int(os.environ["DL_SMALLM_MAX_B"])
        self.record_timing = False  # generate(): HIP events around the prefill / decode parts -> self.last_timing (tools/bench_varlen_stream.py)
        self.last_timing = None
        self.decode_sync_every = int(os.environ.get("DL_SYNC_EVERY", "8"))  # decode steps per chunk of the schedule (KVSlabCache.sched_*)
        self.force_text_decision = None  # tests only: forward() decode keeps / evicts the step's token as given, see forward()
        self.single_split_keys_override = None  # tests only: see _single_split_max_keys
        self.min_keys_per_split = 64  # tests only: KVSlabCache.min_keys_per_split of the caches this model schedules
        self.debug_records = None  # dict filled by forward passes when set to {} (tests)
        self.eval()

    # ---- reference surface -----------------------------------------------------------------
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def encode_images(self, images):  # dynamic_llava_arch.py:163-166
        f = self.get_vision_tower()(images)
        return self._project(f.to(self.dtype))

    def _pack_projector(self):
        """Operand-order copies of the mlp2x_gelu projector's two weights for dl_linear_tiles (multimodal_projector/builder.py:172-179; +42 MB at 7B)."""
        pj = self.model.mm_projector
        self._proj_tiles = None
        vt = self.get_vision_tower()
        ws = (pj[0].weight, pj[2].weight)
        if (vt is None or vt.tiles_gemm) and all(w.is_cuda and ops.linear_tiles_ok(1, w.shape[0], w.shape[1], w.dtype) for w in ws) and pj[0].weight.shape[0] % 64 == 0:
            self._proj_tiles = tuple(ops.pack_weight_tiles(w.detach().contiguous()) for w in ws)
        self._proj_src = [(w.data_ptr(), w._version) for w in ws]

    def _project(self, f):
        """mm_projector(f): Linear -> GELU -> Linear.  16-bit models, up to `tiles_max_batch` images: both Linears on dl_linear_tiles, the GELU in the first one's
        epilogue, the intermediate in fragment order (2 launches instead of 3 library launches)."""
        pj = self.model.mm_projector
        vt = self.get_vision_tower()
        if getattr(self, "_proj_src", None) != [(w.data_ptr(), w._version) for w in (pj[0].weight, pj[2].weight)]:
            self._pack_projector()
        B = f.shape[0] if f.dim() == 3 else 1
        if self._proj_tiles is None or f.dtype != pj[0].weight.dtype or not f.is_cuda or B > (vt.tiles_max_batch if vt is not None else 4) or pj[0].bias is None or pj[2].bias is None:
            return pj(f)
        C = f.shape[-1]
        x = f.reshape(-1, C)  # a view for one image (the CLS row is skipped by the offset), a copy for several
        if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
            x = x.contiguous()
        M, H1, H2 = x.shape[0], pj[0].weight.shape[0], pj[2].weight.shape[0]
        g = ops.linear_tiles(x, self._proj_tiles[0], H1, bias=pj[0].bias, epilogue=ops.LT_GELU, y_packed=True)
        y = ops.linear_tiles(g, self._proj_tiles[1], H2, bias=pj[2].bias, x_packed_mk=(M, H1))
        return y.view(*f.shape[:-1], H2)

    def finalize(self):
        """Call once after weights are loaded / moved: fuses QKV and gate|up, builds the RoPE table."""
        ops.require_gpu()
        if self.device.type != "cuda":
            raise ops.HipOpsError("the model must live on the GPU (no CPU path exists)")
        for l in self.model.layers:
            l.pack(operand_copies=self.packed_prefill_gemm)  # (weights replaced later: call finalize() again -- the operand-order copies are made here)
        self._lp_err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._lp_ws = None
        if any(l.wp_qkv is not None for l in self.model.layers):
            # hand-over workspace of the k-split launches, sized for the largest call this model makes (256 rows, the wider of the two projections);
            # allocated HERE, once: a captured prefill must never allocate it
            cfg = self.config
            need = 256
            for n_, pr in (((cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim, False), (2 * cfg.intermediate_size, True)):
                nu_, ks_ = self._lp_config(n_ // 16, pr)
                ws_ = ops.linear_packed_workspace(256, n_, cfg.hidden_size, self.device, ops.LP_SILU_PAIR if pr else ops.LP_STORE, nu_, ks_)
                need = max(need, 0 if ws_ is None else ws_.numel())
            self._lp_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        if self.get_vision_tower() is not None:
            self.get_vision_tower().pack()
        self._pack_projector()
        self._fp_params = [w for l in self.model.layers for w in (l.self_attn.q_proj.weight, l.self_attn.k_proj.weight, l.self_attn.v_proj.weight, l.self_attn.o_proj.weight,
                                                                   l.mlp.gate_proj.weight, l.mlp.up_proj.weight, l.mlp.down_proj.weight)]
        self._fp = self._weights_fingerprint()
        self._packed = True
        self._build_rope(self.config.max_position_embeddings)
        self._dstate = None
        self._prefill_graphs = {}  # captured prefills hold the pointers of the tensors packed above
        return self

    def _build_rope(self, n_pos):
        """dynamic_modeling_llama.py:152-174,181-184: fp32 cos/sin of cat(freqs, freqs), rounded to the model dtype."""
        d, dev = self.config.head_dim, self.device
        inv_freq = 1.0 / (self.config.rope_theta ** (torch.arange(0, d, 2, device=dev).float() / d))
        t = torch.arange(n_pos, device=dev, dtype=torch.float32)
        emb = torch.cat((torch.outer(t, inv_freq),) * 2, dim=-1)
        self._rope = (emb.cos().to(self.dtype).contiguous(), emb.sin().to(self.dtype).contiguous())

    def _rope_tables(self, need):
        if self._rope is None or self._rope[0].shape[0] < need:
            self._build_rope(max(need, self.config.max_position_embeddings))
            self._dstate = None  # table pointers changed -> re-capture
        return self._rope

    # ---- multimodal glue (dynamic_llava_arch.py:169-601) ---------------------------------------
    def _segments(self, ids_row: List[int], labels_row: Optional[List[int]], n_img_feat: int):
        """Host-side restatement of ARCH:330-340, 418-489 for one row (exactly one image)."""
        img_pos = ids_row.index(IMAGE_TOKEN_INDEX)
        n = len(ids_row)
        if labels_row is None:
            ans0 = n
        else:
            ans0 = max(i for i, v in enumerate(labels_row) if v == IGNORE_INDEX) + 1
        ins = ids_row[img_pos + 1 : ans0]
        starts = [i for i in range(len(ins) - len(USER_IDS) + 1) if ins[i : i + len(USER_IDS)] == USER_IDS]
        last = starts[-1] if starts else 0
        s = img_pos
        i0 = s + n_img_feat
        a0 = i0 + (ans0 - img_pos - 1)
        tot = a0 + (n - ans0)
        return {"system": [0, s], "image": [s, i0], "instruct": [i0, a0], "answer": [a0, tot], "last_instruct": [i0 + last, a0]}

    def _layout(self, input_ids, attention_mask, labels, n_feat):
        """Host pass over the prompt(s): where the text tokens and the image features land in the PACKED batch.
        Returns a dict with `sig` (hashable shape signature -- token VALUES do not enter it), per-row lengths, segment
        dicts, and index lists.  Costs one small device->host copy of input_ids (the reference does several .item()s)."""
        ids_host = input_ids.detach().to("cpu")
        B, W = ids_host.shape
        am = None if attention_mask is None else attention_mask.detach().to("cpu").bool()
        lab = None if labels is None else labels.detach().to("cpu")
        lens, indices, text_src, text_dst, img_dst, img_rows, img_src = [], [], [], [], [], [], []
        base, img_i = 0, 0
        maxlen = getattr(self.config, "tokenizer_model_max_length", None)  # ARCH:493-506: every row is cut to this many embeddings
        truncated = False
        for b in range(B):
            cols = list(range(W)) if am is None else torch.nonzero(am[b]).flatten().tolist()
            r = [int(ids_host[b, c]) for c in cols]
            lr = None if lab is None else [int(lab[b, c]) for c in cols]
            n_images = r.count(IMAGE_TOKEN_INDEX)
            if n_feat == 0 or n_images == 0:  # ARCH:315-324: a text-only row consumes (and ignores) one image feature
                n = len(r) if maxlen is None else min(len(r), maxlen)
                truncated |= n < len(r)
                text_src += [b * W + c for c in cols[:n]]
                text_dst += list(range(base, base + n))
                lens.append(n)
                indices.append(None)
                base += n
                img_i += 1
                continue
            if n_images != 1:
                raise NotImplementedError("exactly one <image> per row (ARCH:330-332 calls .item() on the position)")
            seg = self._segments(r, lr, n_feat)
            p = seg["system"][1]
            # row-local destinations, then the cut at tokenizer_model_max_length, then the packed offsets
            t_src = [b * W + c for j, c in enumerate(cols) if j != p]
            t_dst = list(range(0, p)) + list(range(p + n_feat, len(r) - 1 + n_feat))
            i_dst = list(range(p, p + n_feat))
            n = len(r) - 1 + n_feat
            if maxlen is not None and n > maxlen:
                truncated = True
                n = maxlen
                keep_t = [k for k, dd_ in enumerate(t_dst) if dd_ < n]
                t_src, t_dst = [t_src[k] for k in keep_t], [t_dst[k] for k in keep_t]
                i_dst = [dd_ for dd_ in i_dst if dd_ < n]
                for key in seg:  # ARCH:502-506
                    seg[key] = [min(seg[key][0], n), min(seg[key][1], n)]
            text_src += t_src
            text_dst += [base + dd_ for dd_ in t_dst]
            img_dst += [base + dd_ for dd_ in i_dst]
            img_src += [img_i * n_feat + (dd_ - p) for dd_ in i_dst]
            img_rows.append(img_i)
            img_i += 1
            lens.append(n)
            indices.append(seg)
            base += n
        if all(i is None for i in indices):
            indices = None
        sig = (B, W, tuple(lens), tuple(None if (indices is None or i is None) else i["image"][0] for i in (indices or [None] * B)), n_feat, tuple(text_src))
        return dict(sig=sig, B=B, lens=lens, indices=indices, text_src=text_src, text_dst=text_dst, img_dst=img_dst, img_rows=img_rows, total=base, n_feat=n_feat,
                    img_src=img_src if truncated else None)

    def _assemble(self, lay, dev_idx, input_ids, image_features):
        """Device-only: packed embeds [total,H] from token ids + projector output (index_copy, no host sync)."""
        H = self.config.hidden_size
        # zeros, not empty: with a width bucket `total` exceeds the rows the layout writes, and the rows past the last sequence travel through every
        # row-wise launch of the prefill (they must hold finite values whatever the allocator handed out)
        embeds = torch.zeros((lay["total"], H), dtype=self.dtype, device=self.device)
        ids = input_ids.reshape(-1).index_select(0, dev_idx["text_src"])
        if lay["sig"][0] == "dev":
            # device layout (speculative: every row is ASSUMED to hold one image token).  A row with several leaves IMAGE_TOKEN_INDEX (-200)
            # among the gathered ids; that run is discarded and repeated on the host layout (the kernel raises its error flag), but the
            # gather itself must stay inside the embedding table
            ids = ids.clamp_min(0)
        embeds.index_copy_(0, dev_idx["text_dst"], self.model.embed_tokens(ids))
        if lay["img_dst"] and lay.get("img_src") is not None:  # rows cut inside their image span: only some features are placed
            f = image_features.to(self.dtype).reshape(-1, H).index_select(0, dev_idx["img_src"])
            embeds.index_copy_(0, dev_idx["img_dst"], f)
        elif lay["img_dst"]:
            f = image_features.to(self.dtype)
            if len(lay["img_rows"]) != f.shape[0] or lay["img_rows"] != list(range(f.shape[0])):
                f = f.index_select(0, dev_idx["img_rows"])
            embeds.index_copy_(0, dev_idx["img_dst"], f.reshape(-1, H))
        return embeds

    def _dev_idx(self, lay):
        dev = self.device
        t = lambda x: torch.tensor(x, dtype=torch.long, device=dev)
        d = {"text_src": t(lay["text_src"]), "text_dst": t(lay["text_dst"]), "img_dst": t(lay["img_dst"]), "img_rows": t(lay["img_rows"])}
        if lay.get("img_src") is not None:
            d["img_src"] = t(lay["img_src"])
        return d

    def _n_feat(self, images, image_features):
        if image_features is not None:
            return image_features.shape[1]
        if images is None:
            return 0
        if type(images) is list or images.ndim == 5:
            raise NotImplementedError("anyres / multi-image lists are not on the LLaVA-1.5 Dynamic-LLaVA path")
        return self.get_vision_tower().num_patches

    def _prepare_packed(self, input_ids, attention_mask, labels, images, image_features=None):
        """-> (packed embeds [total,H], lens [B], indices list[dict] or None)."""
        lay = self._layout(input_ids.to(self.device), attention_mask, labels, self._n_feat(images, image_features))
        if image_features is None and images is not None:
            image_features = self.encode_images(images)
        embeds = self._assemble(lay, self._dev_idx(lay), input_ids.to(self.device), image_features)
        return embeds, lay["lens"], lay["indices"]

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images, image_sizes=None):
        """Reference-format wrapper (dynamic_llava_arch.py:169-178, 594-601): right-padded [B,N,H] embeds."""
        if self.get_vision_tower() is None or images is None or input_ids.shape[1] == 1:
            return (input_ids, position_ids, attention_mask, past_key_values, None, labels), (None,)
        if labels is not None:
            raise NotImplementedError("labels / loss are training-side (DML:2713-2800), out of scope")
        embeds, lens, indices = self._prepare_packed(input_ids, attention_mask, labels, images)
        B, N = len(lens), max(lens)
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"  # ARCH:529-555: rows right-aligned, indices shifted by the pad
        out = embeds.new_zeros((B, N, embeds.shape[-1]))
        o = 0
        for b, n in enumerate(lens):
            if left:
                out[b, N - n :] = embeds[o : o + n]
                if indices is not None and indices[b] is not None:
                    for key in indices[b]:
                        indices[b][key] = [indices[b][key][0] + N - n, indices[b][key][1] + N - n]
            else:
                out[b, :n] = embeds[o : o + n]
            o += n
        new_mask = None
        if attention_mask is not None:
            new_mask = torch.zeros((B, N), dtype=attention_mask.dtype, device=attention_mask.device)
            for b, n in enumerate(lens):
                if left:
                    new_mask[b, N - n :] = 1
                else:
                    new_mask[b, :n] = 1
        new_pos = None
        if position_ids is not None:
            new_pos = torch.zeros((B, N), dtype=position_ids.dtype, device=position_ids.device)
            for b, n in enumerate(lens):
                ar = torch.arange(n, dtype=position_ids.dtype, device=position_ids.device)
                if left:
                    new_pos[b, N - n :] = ar
                else:
                    new_pos[b, :n] = ar
        return (None, new_pos, new_mask, past_key_values, out, None), (indices,)

    # ---- decoder engine -----------------------------------------------------------------------
    def _check_ready(self):
        if not self._packed:
            raise ops.HipOpsError("call model.finalize() after loading weights (done by the builders)")
        # the operand-order copies are detached from the parameters.  A load_state_dict() / an in-place edit / a replaced `.data` after
        # finalize() would otherwise leave the packed prefill and the packed decode batches on the OLD weights while the GEMV / library paths use the new ones
        # -- path-dependent results with no error.  (data_ptr, _version) of every decoder projection weight is compared per call (~50 us) and the model
        # re-finalized when one moved.
        if self._weights_fingerprint() != self._fp:
            self._packed = False
            self.finalize()

    def _prefill_knob_key(self):
        """The knobs that decide which launches a captured prefill contains."""
        vt = self.get_vision_tower()
        return (self.packed_prefill_gemm, self.packed_down_proj, self.packed_qkv_parts, self.splitk_o_proj, self.prefill_width_bucket, self.device_prompt_layout,
                None if vt is None else (vt.tiles_gemm, vt.tiles_max_batch, vt.tiles_ksplit_out, vt.tiles_ksplit_fc2))

    def _weights_fingerprint(self):
        return tuple(v for p in self._fp_params for v in (p.data_ptr(), p._version))

    def parameter_bytes(self) -> int:
        """Bytes of the parameters and buffers the state dict holds (what the reference's `model memory` print measures, BIMG:59-67)."""
        seen, n = set(), 0
        for t in list(self.parameters()) + list(self.buffers()):
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                n += t.numel() * t.element_size()
        return n

    def operand_copy_bytes(self) -> dict:
        """Bytes this engine keeps BESIDE the parameters: second copies of weights in matrix-core operand order (decoder q|k|v, gate|up, down_proj for
        dl_linear_packed; the CLIP tower's and the projector's projections for dl_linear_tiles) and the tower's fused q|k|v.  The reference holds none of
        these: the harness counterparts report them separately so that `model memory` stays comparable."""
        nb = lambda t: 0 if t is None else t.numel() * t.element_size()
        dec = sum(nb(l.wp_qkv) + nb(l.wp_gu) + nb(l.wp_down) for l in self.model.layers)
        vt = self.get_vision_tower()
        clip_tiles = int(getattr(vt, "tiles_bytes", 0) or 0) if vt is not None else 0
        clip_fused = sum(nb(w) + nb(b) for w, b in getattr(vt, "_qkv", [])) if vt is not None else 0
        proj = sum(nb(t) for t in (getattr(self, "_proj_tiles", None) or ()))
        return {"decoder_operand_order": dec, "clip_operand_order": clip_tiles, "clip_fused_qkv": clip_fused, "projector_operand_order": proj,
                "total": dec + clip_tiles + clip_fused + proj}

    def _instruct_on(self, indices, B):
        sc = self.config.sparse_config
        return bool(sc["use_text_predictor"] and sc["use_instruct_predictor"]) and indices is not None and len(indices) == B and all(i is not None for i in indices)

    def _plan_prefill(self, lens, indices):
        """Everything about a prefill that the host knows up front (all shapes: k is the same for every row), plus the
        device-side metadata tensors.  Built OUTSIDE hipGraph capture; `_prefill_run` is then pure device work."""
        cfg, sc = self.config, self.config.sparse_config
        dev = self.device
        B = len(lens)
        vision_on = bool(sc["use_vision_predictor"]) and indices is not None and all(i is not None for i in indices) and len(indices) == B
        vision_on = vision_on and sc["sparse_layer"] < cfg.num_hidden_layers  # the layer loop never reaches the sparsification point otherwise (DML:1826)
        n_img = k = 0
        if vision_on:
            n_img = indices[0]["image"][1] - indices[0]["image"][0]
            if any(i["image"][1] - i["image"][0] != n_img for i in indices):
                raise NotImplementedError("all images must have the same token count (DML:1774-1778 assumes it too)")
            k = int(n_img * sc["vision_keep_rate"])  # DML:1899-1901
        i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=dev)
        cu_list = [0]
        for n in lens:
            cu_list.append(cu_list[-1] + n)
        p = dict(B=B, lens=list(lens), vision_on=vision_on, n_img=n_img, k=k, cu_list=cu_list, cu=i32(cu_list), zeros=i32([0] * B), max_len=max(lens))
        p["instruct_on"] = self._instruct_on(indices, B) and sc["sparse_layer"] < cfg.num_hidden_layers
        if p["instruct_on"]:
            # DML:2269 -- the reference asserts B == 1 on this branch
            assert B == 1, "Using text predictor must keep the batch size to 1"
            drop_v = (n_img - k) if vision_on else 0
            p["li"] = (indices[0]["last_instruct"][0] - drop_v, indices[0]["last_instruct"][1] - drop_v)
        p["instruct_drop"] = 0
        p["instruct_dev"] = None  # device-side {kept rows, last row} of the instruct compaction (generate(): no host copy)
        p["nocache"] = False
        p["nocache_lens"] = None
        lens2 = list(lens)
        if vision_on:
            lens2 = [n - (n_img - k) for n in lens]
            cu2 = [0]
            for n in lens2:
                cu2.append(cu2[-1] + n)
            p.update(cu2_list=cu2, cu2=i32(cu2), img_start=i32([ix["image"][0] for ix in indices]), max_len2=max(lens2))
        else:
            p.update(cu2_list=cu_list, cu2=p["cu"], max_len2=p["max_len"])
        SL, L = sc["sparse_layer"], cfg.num_hidden_layers
        # rows the launches are SIZED for (>= the packed rows that exist).  Equally long rows are sized for their width bucket whatever path
        # built the plan (device layout in generate(), host layout, forward()): the library GEMMs pick their kernel by row count, so the
        # same request computes the same bits on every path; the rows past cu[B] hold zeros / padding that nobody consumes.
        p["total"], p["total2"] = cu_list[-1], p["cu2_list"][-1]
        if vision_on and not p["instruct_on"] and len(set(lens)) == 1:
            W_ = lens[0] - n_img + 1
            pad = (self._width_bucket(W_, n_img) - W_) * B
            p["total"], p["total2"] = p["total"] + pad, p["total2"] + pad
        p["lens2"] = lens2
        p["lens_dev"] = i32([list(lens), lens2 if (SL < L) else list(lens)])
        p["last_rows"] = torch.tensor([c - 1 for c in p["cu2_list"][1:]], dtype=torch.long, device=dev)
        return p

    def _prefill_run(self, p, embeds, cache: KVSlabCache, indices, last_only: bool):
        """Packed prefill, device work only.  Returns the normed hidden state (all rows, or the last row of each sequence)."""
        cfg, sc = self.config, self.config.sparse_config
        dev, dt = self.device, self.dtype
        B = p["B"]
        nH, nKV, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        eps = cfg.rms_norm_eps
        L, SL = cfg.num_hidden_layers, sc["sparse_layer"]
        vision_on, n_img, k = p["vision_on"], p["n_img"], p["k"]
        cos, sin = self._rope
        cu, cu_list, max_len, total = p["cu"], p["cu_list"], p["max_len"], (p["cu_list"][-1] if p["nocache"] else p["total"])
        zeros_b = p["zeros"]
        pos = None  # layers < SL: position = in-row index
        h = embeds.to(dt).contiguous()
        if h.shape[0] < total:  # sized for the width bucket (see _plan_prefill): the extra rows are zeros behind the last sequence
            h = torch.cat([h, h.new_zeros((total - h.shape[0], h.shape[1]))], dim=0)
        elif h.data_ptr() == embeds.data_ptr():
            h = h.clone()  # the residual stream is updated in place; never touch the caller's tensor
        rec = self.debug_records
        # dl_linear_packed for q|k|v and gate|up of the layers whose packed row count fits its one tile (<= 256 rows: the post-compaction layers of a
        # B = 1 request).  `x_pk`: x is in fragment order (written that way by the norm launch that produced it).
        lp_ok = lambda rows_, layer_: (self.packed_prefill_gemm and layer_.wp_qkv is not None and 0 < rows_ <= ops.LP_MAX_ROWS and dt in (torch.bfloat16, torch.float16))
        x_pk = lp_ok(total, self.model.layers[0]) and not (SL == 0 and (vision_on or p["instruct_on"] or p["nocache"]))
        x = ops.rmsnorm(h, self.model.layers[0].input_layernorm.weight, eps, packed=x_pk)
        attn_buf = None
        qkv_buf = None
        for i, layer in enumerate(self.model.layers):
            if i == SL and vision_on:
                # ---- F1..F5: predictor -> top-k -> compaction (DML:1826-1994) on the un-normed residual stream ----
                vp = self.model.image_score_predictor
                if len(vp._forward_hooks) or len(vp._forward_pre_hooks):  # keep the reference's hook point alive
                    dense = torch.stack([h[cu_list[b] + indices[b]["image"][0] : cu_list[b] + indices[b]["image"][1]] for b in range(B)])
                    logits = vp(dense, torch.ones(B, n_img, 1, dtype=dt, device=dev))
                    score = vp.last_score
                else:
                    logits, score = vp.score_packed(h, cu, p["img_start"], n_img)
                keep = ops.topk_select(score, k)
                # compaction + this layer's input RMSNorm in one launch (unless a text-predictor compaction still follows at this layer)
                fuse_norm = not p["instruct_on"] and not p["nocache"]
                total2 = p["cu2_list"][-1] if p["nocache"] else p["total2"]
                if fuse_norm:
                    h, pos, x_fused = ops.compact_tokens(h, keep, cu, p["cu2"], p["img_start"], n_img, k, total2, layer.input_layernorm.weight, eps)
                else:
                    h, pos = ops.compact_tokens(h, keep, cu, p["cu2"], p["img_start"], n_img, k, total2)
                if rec is not None:
                    rec.update(vision_logit=logits, vision_score=score, keep_index=keep, position_ids=pos[: p["cu2_list"][-1]], cu_after=p["cu2"])
                cu, cu_list, max_len, total = p["cu2"], p["cu2_list"], p["max_len2"], total2
            if i == SL and p["instruct_on"]:
                # ---- SURVEY 8f N2 / DML:2261-2375: prefill, first instruct -- the instruct predictor drops tokens of the last
                # instruct span (its final token always stays).  The kept count is data dependent: one device->host copy, as in
                # the reference (torch.where).  B == 1 only, like the reference.
                li0, li1 = p["li"]
                n_span = li1 - 1 - li0
                if n_span > 0:
                    tp = self.model.instruct_score_predictor
                    dec = torch.empty(n_span, dtype=torch.int32, device=dev)
                    lg = torch.empty((n_span, 2), dtype=torch.float32, device=dev)
                    tp.decide(h[li0 : li1 - 1], ops.text_predictor_workspace(n_span, tp.d_model, dev), lg, dec)
                    if p.get("device_instruct"):
                        # generate(): the kept count stays on the device.  Every following launch is sized for the UPPER bound (no row
                        # dropped) and reads the true length from device memory (cu); the rows past it are zeros that nobody consumes.
                        # Host-visible bookkeeping (the reference shifts its index dicts by the drop count, DML:2365-2375) is internal here.
                        h, pos, cu, counts = ops.compact_rows_by_mask(h, pos, dec, li0, n_span)
                        p["instruct_dev"] = counts
                        continue_host = False
                    else:
                        continue_host = True
                if n_span > 0 and continue_host:
                    keep_rel = torch.nonzero(dec).flatten()
                    idx = torch.cat([torch.arange(0, li0, device=dev), keep_rel + li0, torch.arange(li1 - 1, total, device=dev)])
                    if pos is None:
                        pos = torch.arange(total, dtype=torch.int32, device=dev)
                    h = h.index_select(0, idx)
                    pos = pos.index_select(0, idx)
                    total = int(idx.numel())
                    p["instruct_drop"] = n_span - int(keep_rel.numel())
                    cu_list, max_len = [0, total], total
                    cu = torch.tensor(cu_list, dtype=torch.int32, device=dev)
                    if rec is not None:
                        rec.update(instruct_logit=lg, instruct_keep=keep_rel, position_ids=pos, cu_after=cu)
            if i == SL and p["nocache"] and not p["instruct_on"] and indices is not None and sc["use_text_predictor"] and sc["use_output_text_predictor"]:
                # ---- SURVEY 8f N3 / DML:2393-2504: decode WITHOUT KV cache.  The answer tokens [answer_indice, -1) of every row are
                # compacted by top-k of the RAW keep logit with k = max kept count over the batch (data dependent: one host copy).
                # First call: answer_indice == row length, so the last token is duplicated -- reproduced on purpose.
                L_row = cu_list[1] - cu_list[0]
                if any(cu_list[b + 1] - cu_list[b] != L_row for b in range(B)):
                    raise NotImplementedError("use_cache=False expects equally long rows (the reference uses row 0's answer_indice for all, DML:2402-2409)")
                if self.model.answer_indice is None:
                    self.model.answer_indice = indices[0]["instruct"][1] - ((n_img - k) if vision_on else 0)
                ai = self.model.answer_indice
                n_span = max(0, L_row - 1 - ai)
                num_keep = 0
                keep = torch.zeros((B, 0), dtype=torch.int64, device=dev)
                if n_span > 0:
                    tp = self.model.output_text_score_predictor
                    rows = (torch.tensor(cu_list[:-1], device=dev)[:, None] + ai + torch.arange(n_span, device=dev)[None, :]).reshape(-1)
                    dec = torch.empty(B * n_span, dtype=torch.int32, device=dev)
                    lg = torch.empty((B * n_span, 2), dtype=torch.float32, device=dev)
                    tp.decide(h.index_select(0, rows), ops.text_predictor_workspace(B * n_span, tp.d_model, dev), lg, dec)
                    num_keep = int(dec.view(B, n_span).sum(dim=1).max().item())
                    if num_keep > 0:
                        keep = ops.topk_select(lg[:, 0].to(dt).view(B, n_span).contiguous(), num_keep)
                    if rec is not None:
                        rec.update(nocache_logit=lg.view(B, n_span, 2), nocache_keep=keep)
                left = torch.arange(min(ai, L_row), device=dev)
                idx = torch.cat([torch.cat([left, ai + keep[b], torch.tensor([L_row - 1], device=dev)]) + cu_list[b] for b in range(B)])
                if pos is None:
                    pos = torch.cat([torch.arange(L_row, dtype=torch.int32, device=dev) for _ in range(B)])
                h = h.index_select(0, idx)
                pos = pos.index_select(0, idx)
                L_new = int(left.numel()) + num_keep + 1
                cu_list = [b * L_new for b in range(B + 1)]
                cu = torch.tensor(cu_list, dtype=torch.int32, device=dev)
                max_len, total = L_new, B * L_new
                p["nocache_lens"] = [L_new] * B
                if rec is not None:
                    rec.update(position_ids=pos, cu_after=cu)
            use_lp = lp_ok(total, layer)
            if i == SL and vision_on and not p["instruct_on"] and not p["nocache"]:
                # (the compaction launch normalises the rows it moves; for the packed GEMM they are normalised again into fragment order: one ~5 us
                # launch at one layer buys that layer's two GEMMs)
                x, x_pk = (ops.rmsnorm(h, layer.input_layernorm.weight, eps, packed=True), True) if use_lp else (x_fused, False)
            elif i == SL and (vision_on or p["instruct_on"] or p["nocache"]):
                x_pk = use_lp
                x = ops.rmsnorm(h, layer.input_layernorm.weight, eps, packed=x_pk)
            use_lp = use_lp and x_pk
            Nq = layer.w_qkv.shape[0]
            nu_q, ks_q = self._lp_config(Nq // 16, False)
            if use_lp and self.packed_qkv_parts and ks_q > 1:
                # the two k ranges of q|k|v leave fp32 partial sums instead of meeting inside the GEMM launch (its hand-over is 8-11 us of a 35 us launch); the
                # RoPE / KV-append launch adds them -- the same sum, rounded once -- and writes q, k, v for the attention
                if qkv_buf is None or qkv_buf.shape[0] != total or qkv_buf.shape[1] != Nq:
                    qkv_buf = torch.zeros((total, Nq), dtype=dt, device=dev)  # (rows past the last sequence are never written, nor read)
                parts_q = ops.linear_packed(x, layer.wp_qkv, Nq, out=self._qkv_parts_ws(ks_q * ops.LP_MAX_ROWS * Nq), epilogue=ops.LP_PARTS, units_per_workgroup=nu_q, k_split=ks_q,
                                            x_packed_mk=(total, h.shape[1]))
                qkv = qkv_buf
                ops.rope_kv_write(qkv, cos, sin, cu, pos, zeros_b, zeros_b, cache.k[i], cache.v[i], nH, nKV, d, parts=parts_q)
            else:
                qkv = self._lp_linear(x, total, layer.wp_qkv, Nq, h.shape[1]) if use_lp else F.linear(x, layer.w_qkv)
                ops.rope_kv_write(qkv, cos, sin, cu, pos, zeros_b, zeros_b, cache.k[i], cache.v[i], nH, nKV, d)
            if attn_buf is None or attn_buf.shape[0] != total:
                # one zero-filled buffer per row count, shared by the layers (the attention launch writes the rows of real sequences only: padding rows
                # of a width bucket stay zero instead of holding whatever the allocator handed out -- ADVICE r4)
                attn_buf = torch.zeros((total, nH * d), dtype=dt, device=dev)
            attn = attn_buf
            ops.attn_prefill(qkv[:, : nH * d], qkv[:, nH * d : (nH + nKV) * d], qkv[:, (nH + nKV) * d :], attn, cu, max_len, nH, nKV, d, True)
            if self.splitk_o_proj and dt in (torch.bfloat16, torch.float16) and attn.shape[0] <= 192 and attn.shape[1] >= 1024 and attn.shape[1] % 64 == 0 and h.shape[1] % 64 == 0:
                x = ops.add_rmsnorm_parts(h, ops.linear_splitk(attn, layer.self_attn.o_proj.weight, self._splitk_ws(h.shape[1]), 8), layer.post_attention_layernorm.weight, eps, packed=use_lp)
            else:
                o = F.linear(attn, layer.self_attn.o_proj.weight)
                x = ops.add_rmsnorm(h, o, layer.post_attention_layernorm.weight, eps, packed=use_lp)
            lp_down = use_lp and layer.wp_down is not None and self.packed_down_proj
            if use_lp:  # gate|up with silu(gate) * up in the epilogue: one launch, no [rows, 2 I] round trip
                act = self._lp_linear(x, total, layer.wp_gu, layer.w_gu.shape[0], h.shape[1], ops.LP_SILU_PAIR, y_packed=lp_down)
            else:
                act = ops.silu_mul(F.linear(x, layer.w_gu))
            nw_next = self.model.norm.weight if i + 1 == L else (None if i + 1 == SL and (vision_on or p["instruct_on"] or p["nocache"])  # residual add only: layer SL's norm runs after compaction
                                                               else self.model.layers[i + 1].input_layernorm.weight)
            pk_next = nw_next is not None and i + 1 < L and lp_ok(total, self.model.layers[i + 1])  # the next layer's q|k|v reads this norm's output
            if lp_down:
                # down_proj on the operand-order copy: 4 k ranges per unit set, fp32 partial sums added in range order by the residual-add / RMSNorm launch
                I_ = layer.w_gu.shape[0] // 2
                nu_, ks_ = self._lp_config_parts(h.shape[1] // 16, total)
                parts_ = ops.linear_packed(act, layer.wp_down, h.shape[1], out=self._splitk_ws(h.shape[1])[: ks_ * total * h.shape[1]], epilogue=ops.LP_PARTS, units_per_workgroup=nu_,
                                           k_split=ks_, x_packed_mk=(total, I_))
                x_new = ops.add_rmsnorm_parts(h, parts_, nw_next, eps, packed=pk_next)
            elif dt in (torch.bfloat16, torch.float16) and act.shape[0] <= 192 and act.shape[1] >= 1024 and act.shape[1] % 64 == 0 and h.shape[1] % 64 == 0:
                # down_proj at <= 192 packed rows (the compacted layers at B=1): the library streams [H, I] at 1.8 TB/s there; dl_linear_splitk
                # cuts K into 8 slices and the residual-add / RMSNorm launch adds them in order (tools/bench_linear_splitk.py: 44 vs 54 us)
                x_new = ops.add_rmsnorm_parts(h, ops.linear_splitk(act, layer.mlp.down_proj.weight, self._splitk_ws(h.shape[1]), 8), nw_next, eps, packed=pk_next)
            else:
                x_new = ops.add_rmsnorm(h, F.linear(act, layer.mlp.down_proj.weight), nw_next, eps, packed=pk_next)
            x = x if nw_next is None else x_new
            x_pk = pk_next if nw_next is not None else x_pk
        cache.lens.copy_(p["lens_dev"])  # layers < SL hold the full prompt, layers >= SL the compacted one
        if p.get("instruct_dev") is not None:  # device-side instruct compaction (B == 1): kept rows / last row index live on the device
            cache.lens[1].copy_(p["instruct_dev"][:1])
            if last_only:
                x = x.index_select(0, p["instruct_dev"][1:2])
            return x
        if p["instruct_drop"]:
            cache.lens[1] -= p["instruct_drop"]
        if last_only:
            x = x.index_select(0, p["last_rows"] - p["instruct_drop"])
        return x

    # DL_USE_HIP_GRAPH=0: debugging switch that wins over any assignment -- e.g. the GPU tests under PYTORCH_NO_CUDA_MEMORY_CACHING=1 (every tensor its own
    # allocation, so that an out-of-bounds read faults instead of landing in a neighbour; stream capture is impossible without the caching allocator)
    _force_eager = os.environ.get("DL_USE_HIP_GRAPH", "1") == "0"

    @property
    def use_hip_graph(self):
        return self._use_hip_graph and not self._force_eager

    @use_hip_graph.setter
    def use_hip_graph(self, v):
        self._use_hip_graph = bool(v)

    @staticmethod
    def _lp_config(n_units: int, pairs: bool):
        """(units per workgroup, k ranges) of a dl_linear_packed launch: one workgroup per CU; two k ranges per unit set where that still leaves at
        most 8 units per workgroup (q|k|v: every CU then pulls half of X through its L1 beside the weight stream -- the bound of this kernel,
        DESIGN.md section 4), else one (gate|up at 7B / 13B: 6 / 8 units, no hand-over)."""
        for ks in (2, 1):
            for nu in (1, 2, 3, 4, 6, 8):
                if pairs and nu % 2:
                    continue
                if -(-n_units // nu) * ks <= 256:
                    return nu, ks
        return 8, 1

    @staticmethod
    def _lp_config_parts(n_units: int, rows: int = 0):
        """(units per workgroup, k ranges) of a partial-sum launch (narrow N: o_proj / down_proj): as many k ranges as keep one workgroup per CU with at
        most 4 units each -- every CU then pulls 1 / k_split of X through its L1 (256 units at 7B: 4 units x 4 ranges).  With more than 128 rows, where X is
        what the launch waits for, 8 units x 8 ranges when that is exactly one workgroup per CU (7B down_proj at M = 170: 32.6 -> 28.5 us, with the consumer's
        eight slices 39.5 -> 36.6; a tie at 117 rows, slower at 32)."""
        if 128 < rows <= 192 and n_units % 8 == 0 and n_units // 8 * 8 == 256:
            return 8, 8
        for ks in (4, 2, 1):
            for nu in (1, 2, 3, 4):
                if -(-n_units // nu) * ks <= 256:
                    return nu, ks
        return 4, 1

    def _lp_linear(self, x_pk, rows, wp, N, K, epilogue=ops.LP_STORE, y_packed=False):
        """x [rows, K] in fragment order @ W^T on the operand-order copy wp."""
        nu, ks = self._lp_config(N // 16, epilogue == ops.LP_SILU_PAIR)
        return ops.linear_packed(x_pk, wp, N, epilogue=epilogue, units_per_workgroup=nu, k_split=ks, workspace=self._lp_ws if ks > 1 else None, err=self._lp_err,
                                 x_packed_mk=(rows, K), y_packed=y_packed)

    def _qkv_parts_ws(self, n):
        """fp32 partial sums of the q|k|v projection (k ranges x LP_MAX_ROWS x columns: one size per model, so that captured graphs keep a valid pointer)."""
        ws = getattr(self, "_qkv_parts_buf", None)
        if ws is None or ws.numel() < n:
            ws = self._qkv_parts_buf = torch.empty(n, dtype=torch.float32, device=self.device)
        return ws

    def _splitk_ws(self, H):
        """fp32 split-K partials of dl_linear_splitk (8 slices x <= 192 rows x H), allocated once."""
        ws = getattr(self, "_splitk_buf", None)
        if ws is None or ws.numel() < 8 * 192 * H:
            ws = self._splitk_buf = torch.empty(8 * 192 * H, dtype=torch.float32, device=self.device)
        return ws

    def _prefill_host_update(self, p, cache, indices):
        """Host mirrors of what `_prefill_run` did on the device (also the reference's in-place index shift, DML:1986-1994)."""
        cache.full_len_host = list(p["lens"])
        cache.seen_tokens = max(p["lens"])
        cache.sparse_cap = cache.logical_cap - (max(p["lens"]) - max(p["lens2"])) - p["instruct_drop"]  # host-known upper bound of the evicted group's lengths
        cache.prefill_sparse_max = max(p["lens2"]) - p["instruct_drop"]  # longest row of layers >= sparse_layer after the prefill (upper bound when the instruct compaction stayed on the device)
        cache.set_bounds(None, None)
        cache.sched_begin(max(p["lens"]), cache.prefill_sparse_max, self.decode_sync_every)
        if p["instruct_drop"]:  # DML:2365-2375
            for ix in indices:
                ix["instruct"][1] -= p["instruct_drop"]
                ix["last_instruct"][1] -= p["instruct_drop"]
                ix["answer"][0] -= p["instruct_drop"]
                ix["answer"][1] -= p["instruct_drop"]
        if p["vision_on"]:
            drop = p["n_img"] - p["k"]
            for ix in indices:
                ix["image"][1] -= drop
                for key in ("instruct", "last_instruct", "answer"):
                    ix[key][0] -= drop
                    ix[key][1] -= drop

    def _prefill(self, embeds, lens, indices, cache: Optional[KVSlabCache], reserve: int, last_only: bool):
        """Eager packed prefill (forward() API and first-time shapes).  Returns (x, cache, lens_after, cu_after)."""
        cfg, sc = self.config, self.config.sparse_config
        p = self._plan_prefill(lens, indices)
        if cache is None:
            cache = KVSlabCache(cfg.num_hidden_layers, sc["sparse_layer"], p["B"], cfg.num_key_value_heads, cfg.head_dim, max(lens) + reserve, self.dtype, self.device)
        elif max(cache.full_len_host) != 0:
            raise NotImplementedError("multi-token forward on a non-empty cache (new-instruct round, DML:2506-2521) is SURVEY 8f row N2")
        self._rope_tables(max(lens) + reserve)
        x = self._prefill_run(p, embeds, cache, indices, last_only)
        self._prefill_host_update(p, cache, indices)
        if p["instruct_drop"]:
            n = p["lens2"][0] - p["instruct_drop"]
            return x, cache, [n], [0, n]
        return x, cache, p["lens2"], p["cu2_list"]

    # ---- one decode step; every buffer persistent, no host sync -> hipGraph-capturable ----
    def knobs(self) -> dict:
        """The resolved values of every tuning knob and test hook that decides WHICH kernels a request runs on (constructor defaults, DL_* environment
        values, attributes set later) -- recorded in the bench line so that a run can be reproduced.  The last three are test hooks:
        they must be None / 0 outside tests/."""
        return {
            "use_hip_graph": self.use_hip_graph, "attn_inkernel_combine": self.attn_inkernel_combine, "device_prompt_layout": self.device_prompt_layout,
            "tp_side_stream": self.tp_side_stream, "gemv_max_decode_batch": self.gemv_max_decode_batch, "smallm_max_decode_batch": self.smallm_max_decode_batch,
            "fuse_qkv_attn": self.fuse_qkv_attn, "fuse_gu_tp": self.fuse_gu_tp, "fused_attn_max_splits": self.fused_attn_max_splits, "gu_grid_cap": self.gu_grid_cap,
            "qkv_attn_grid_cap": self.qkv_attn_grid_cap, "splitk_o_proj": self.splitk_o_proj, "packed_prefill_gemm": self.packed_prefill_gemm, "packed_down_proj": self.packed_down_proj, "packed_qkv_parts": self.packed_qkv_parts, "packed_decode_mlp": self.packed_decode_mlp, "packed_decode_mlp_min_batch": self.packed_decode_mlp_min_batch, "packed_decode_qkv_min_batch": self.packed_decode_qkv_min_batch,
            "smallm_wide_slices": self.smallm_wide_slices, "decode_sync_every": self.decode_sync_every, "prefill_width_bucket": self.prefill_width_bucket,
            "max_prefill_graphs": self.max_prefill_graphs,
            "clip_tiles_gemm": getattr(self.get_vision_tower(), "tiles_gemm", None), "clip_tiles_max_batch": getattr(self.get_vision_tower(), "tiles_max_batch", None),
            "clip_tiles_ksplit": [getattr(self.get_vision_tower(), "tiles_ksplit_out", None), getattr(self.get_vision_tower(), "tiles_ksplit_fc2", None)],
            "test_hook_force_text_decision": self.force_text_decision is not None, "test_hook_single_split_keys_override": self.single_split_keys_override,
            "test_hook_min_keys_per_split": getattr(self, "min_keys_per_split", None),
        }

    def check_device_errors(self):
        """Raises if a launch with in-kernel hand-offs (dl_gemv_qkv_attn, dl_gemv_gu_tp) gave up on a wait since the last check (such a launch
        poisons its output instead of hanging).  Costs one device->host copy: call it where a sync is acceptable."""
        if self._lp_err is not None and self._lp_ws is not None:
            code = int(self._lp_err.item())
            if code != 0:
                self._lp_err.zero_()
                self._lp_ws.zero_()
                raise ops.HipOpsError("in-kernel hand-off aborted: dl_linear_packed (a k range's partial tiles never arrived)")
        st = self._dstate
        if st is not None:
            code = int(st.blk_err.item())
            if code != 0:
                st.blk_err.zero_()
                what = [n for bit, n in ((1, "dl_gemv_qkv_attn (attention never received its projection outputs)"), (2, "dl_gemv_gu_tp (a predictor stage never received its inputs)")) if code & bit]
                if code & ~3:
                    what.append(f"unknown error bits {code & ~3:#x}")
                raise ops.HipOpsError("in-kernel hand-off aborted: " + "; ".join(what))

    def _decode_step_kernels(self, st: _DecodeState, cache: KVSlabCache, advance: bool):
        if st.use_gemv:
            self._decode_step_gemv(st, cache)
        else:
            self._decode_step_gemm(st, cache)
        if advance:
            sc = self.config.sparse_config
            use_tp = bool(sc["use_text_predictor"] and sc["use_output_text_predictor"]) and sc["sparse_layer"] < self.config.num_hidden_layers
            ops.decode_advance(
                st.logits, st.cur_ids, st.out_ids, st.step, st.finished, self._eos, self._pad, cache.len_full, cache.len_sparse,
                st.decision if use_tp else None, min_new_tokens=getattr(self, "_min_new", 0),
            )

    def _decode_step_gemv(self, st: _DecodeState, cache: KVSlabCache):
        """Small-batch decode step: 5 weight-streaming launches per layer (dl_gemv with fused residual-add+RMSNorm /
        SiLU*up prologues) + RoPE/KV append + split-KV attention.  The residual stream ping-pongs between st.h / st.h2."""
        cfg, sc = self.config, self.config.sparse_config
        nH, nKV, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        eps, L, SL = cfg.rms_norm_eps, cfg.num_hidden_layers, sc["sparse_layer"]
        cos, sin = self._rope
        use_tp = bool(sc["use_text_predictor"] and sc["use_output_text_predictor"]) and SL < L
        torch.index_select(self.model.embed_tokens.weight, 0, st.cur_ids, out=st.h)
        h_cur, h_alt, delta = st.h, st.h2, None
        A = ops.GEMV_ADDNORM
        for i, layer in enumerate(self.model.layers):
            lens = cache.len_of_layer(i)
            ns = cache.n_splits(i, st.B * nH)
            # q|k|v projection + single-split attention of a batch-1 layer in ONE launch (dl_gemv_qkv_attn: the attention workgroups fetch their
            # K/V rows while the weights stream and receive the projection as granules).  Same bodies as the two launches below, so the
            # results are bit-identical to them WHEN the stand-alone attention also runs four waves (KVSlabCache.eight_wave_single_split =
            # False, as the kernel tests set it); by default the stand-alone single-split launch of a small batch runs eight waves -- another
            # (equally valid) summation order, so DL_FUSE_QKV_ATTN=0 is an A/B of speed, not of bits (tokens / KV lengths: tested equal)
            fused_attn = self.fuse_qkv_attn and st.B == 1 and ns == 1 and st.qa_gran is not None
            if fused_attn:
                ops.gemv_qkv_attn(layer.w_qkv, st.qkv, h_cur, h_alt, delta, layer.input_layernorm.weight, eps, cos, sin, cache.len_full, lens, cache.k[i], cache.v[i],
                                  st.attn, st.qa_gran, i & 0xff, nH, nKV, d, err=st.blk_err, grid_cap=self.qkv_attn_grid_cap,
                                  n_splits=cache.fused_attn_splits(i, self.fused_attn_max_splits))
                if delta is not None:
                    h_cur, h_alt = h_alt, h_cur
            else:
                ops.gemv(layer.w_qkv, st.qkv, mode=A, h_in=h_cur, h_out=h_alt, delta=delta, norm_w=layer.input_layernorm.weight, eps=eps)
                if delta is not None:
                    h_cur, h_alt = h_alt, h_cur
            # the predictor as extra workgroups of this layer's gate|up launch (dl_gemv_gu_tp): its input is that launch's h_in
            fused_tp = i == SL and use_tp and self.fuse_gu_tp and not self.tp_side_stream and st.B == 1 and st.tp_gran is not None
            if i == SL and use_tp and not fused_tp:  # F6: decision on the hidden state entering layer SL (DML:2377-2391)
                # only the end-of-step length advance consumes the decision: run the predictor on a side stream (a parallel
                # branch of the captured graph) on a snapshot of the residual stream, off the layer chain's critical path
                if self.tp_side_stream:
                    st.tp_x.copy_(h_cur)
                    st.tp_stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st.tp_stream):
                        self.model.output_text_score_predictor.decide(st.tp_x, st.tp_ws, st.tp_logits, st.decision)
                else:
                    self.model.output_text_score_predictor.decide(h_cur, st.tp_ws, st.tp_logits, st.decision)
            # F8+F10+F9: RoPE, KV append at slot len[b] and ragged attention in one launch (1024-thread workgroups; split-KV
            # only when the row is long enough to need more than one workgroup per head)
            if not fused_attn:
                ops.attn_decode_rope(st.qkv, cos, sin, cache.len_full, lens, cache.k[i], cache.v[i], st.attn, st.attn_ws, ns, nH, nKV, d, keys_in_flight=cache.keys_in_flight(ns, st.B * nH), chunk_keys=cache.spec_chunk(ns),
                                     call_tag=(i & 0xff) if self.attn_inkernel_combine and L >= 2 else -1)
            ops.gemv(layer.self_attn.o_proj.weight, st.o, x=st.attn)
            if fused_tp:
                tp = self.model.output_text_score_predictor
                ops.gemv_gu_tp(layer.w_gu, st.gu, h_cur, h_alt, st.o, layer.post_attention_layernorm.weight, eps, tp._weights(), tp.d_model, st.tp_ws, st.tp_logits,
                               st.decision, cache.len_full, st.tp_gran, i & 0xff, err=st.blk_err)
            else:
                ops.gemv(layer.w_gu, st.gu, mode=A | ops.GEMV_OUT_SILU_PAIR, h_in=h_cur, h_out=h_alt, delta=st.o, norm_w=layer.post_attention_layernorm.weight, eps=eps, grid_cap=self.gu_grid_cap)
            h_cur, h_alt = h_alt, h_cur
            ops.gemv(layer.mlp.down_proj.weight, st.dn, x=st.gu)
            delta = st.dn
        ops.gemv(self.lm_head.weight, st.logits, mode=A, h_in=h_cur, h_out=h_alt, delta=delta, norm_w=self.model.norm.weight, eps=eps)
        if use_tp and self.tp_side_stream:
            torch.cuda.current_stream().wait_stream(st.tp_stream)  # join before anything reads st.decision

    def _decode_step_gemm(self, st: _DecodeState, cache: KVSlabCache):
        """Decode step for batches past the GEMV range (round 5, `profiles/r05_decode_batch_paths.txt`).  Up to smallm_max_decode_batch (32) rows:
        o_proj -- and q|k|v below packed_decode_qkv_min_batch (16) rows -- on dl_gemm_smallm (row-major weights streamed into the matrix cores, fp32
        split-K partials added by the residual-add / RMSNorm launch); from packed_decode_mlp_min_batch (4) rows the MLP, from 16 rows q|k|v too, on
        dl_linear_packed (operand-order weight copies; SiLU * up in the epilogue, down_proj as 4 k ranges of partial sums; the norm launches write the
        GEMMs' input in fragment order): 7 launches per layer.  Larger batches, or a model without operand copies: library GEMMs."""
        cfg, sc = self.config, self.config.sparse_config
        nH, nKV, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        eps, L, SL = cfg.rms_norm_eps, cfg.num_hidden_layers, sc["sparse_layer"]
        cos, sin = self._rope
        use_tp = bool(sc["use_text_predictor"] and sc["use_output_text_predictor"]) and SL < L
        sm, ws = st.use_smallm, st.lin_ws
        torch.index_select(self.model.embed_tokens.weight, 0, st.cur_ids, out=st.h)
        lp_qkv = st.use_lp_mlp and st.B >= self.packed_decode_qkv_min_batch and getattr(self.model.layers[0], "wp_qkv", None) is not None
        if lp_qkv:
            nu_q, ks_q = self._lp_config(st.qkv.shape[1] // 16, False)
        ops.rmsnorm(st.h, self.model.layers[0].input_layernorm.weight, eps, out=st.x_pk if lp_qkv else st.x, packed=lp_qkv)
        for i, layer in enumerate(self.model.layers):
            if i == SL and use_tp:  # F6: decision on the hidden state entering layer SL (DML:2377-2391)
                self.model.output_text_score_predictor.decide(st.h, st.tp_ws, st.tp_logits, st.decision)
            lens = cache.len_of_layer(i)
            if lp_qkv:
                qkv = ops.linear_packed(st.x_pk, layer.wp_qkv, st.qkv.shape[1], out=st.qkv, units_per_workgroup=nu_q, k_split=ks_q, workspace=self._lp_ws if ks_q > 1 else None, err=self._lp_err,
                                        x_packed_mk=(st.B, st.h.shape[1]))
            else:
                qkv = ops.gemm_smallm(st.x, layer.w_qkv, out=st.qkv, workspace=ws, n_slices=self.smallm_wide_slices) if sm else F.linear(st.x, layer.w_qkv)
            ns = cache.n_splits(i, st.B * nH)
            ops.attn_decode_rope(qkv, cos, sin, cache.len_full, lens, cache.k[i], cache.v[i], st.attn, st.attn_ws, ns, nH, nKV, d, keys_in_flight=cache.keys_in_flight(ns, st.B * nH), chunk_keys=cache.spec_chunk(ns),
                                 call_tag=(i & 0xff) if self.attn_inkernel_combine and L >= 2 else -1)
            nw = self.model.norm.weight if i + 1 == L else self.model.layers[i + 1].input_layernorm.weight
            lp = st.use_lp_mlp
            x_mlp = st.x_pk if lp else st.x  # the packed MLP reads its input in fragment order: the norm launch writes it that way
            if sm:  # (o_proj on dl_linear_packed's partial sums instead: a tie at 8..32 rows, measured and dropped)
                parts, _ = ops.gemm_smallm_parts(st.attn, layer.self_attn.o_proj.weight, ws)
                ops.add_rmsnorm_parts(st.h, parts, layer.post_attention_layernorm.weight, eps, out=x_mlp, packed=lp)
            else:
                o = F.linear(st.attn, layer.self_attn.o_proj.weight)
                ops.add_rmsnorm(st.h, o, layer.post_attention_layernorm.weight, eps, out=x_mlp, packed=lp)
            if lp:
                H_, I2 = st.h.shape[1], layer.w_gu.shape[0]
                nu_g, ks_g = self._lp_config(I2 // 16, True)
                ops.linear_packed(st.x_pk, layer.wp_gu, I2, out=st.act_pk, epilogue=ops.LP_SILU_PAIR, units_per_workgroup=nu_g, k_split=ks_g, workspace=self._lp_ws if ks_g > 1 else None,
                                  err=self._lp_err, x_packed_mk=(st.B, H_), y_packed=True)
                nu_d, ks_d = self._lp_config_parts(H_ // 16)
                parts = ops.linear_packed(st.act_pk, layer.wp_down, H_, out=st.lp_parts, epilogue=ops.LP_PARTS, units_per_workgroup=nu_d, k_split=ks_d, x_packed_mk=(st.B, I2 // 2))
                nxt_pk = lp_qkv and i + 1 < L  # the final norm feeds lm_head: row-major
                ops.add_rmsnorm_parts(st.h, parts, nw, eps, out=st.x_pk if nxt_pk else st.x, packed=nxt_pk)
            elif sm:
                parts, _ = ops.gemm_smallm_parts(st.x, layer.w_gu, ws, n_slices=self.smallm_wide_slices)
                ops.silu_mul_parts(parts, st.act)
                parts, _ = ops.gemm_smallm_parts(st.act, layer.mlp.down_proj.weight, ws)
                ops.add_rmsnorm_parts(st.h, parts, nw, eps, out=st.x)
            else:
                ops.silu_mul(F.linear(st.x, layer.w_gu), out=st.act)
                dn = F.linear(st.act, layer.mlp.down_proj.weight)
                ops.add_rmsnorm(st.h, dn, nw, eps, out=st.x)
        if sm:
            ops.gemm_smallm(st.x, self.lm_head.weight, out=st.logits, workspace=ws)
        else:
            torch.matmul(st.x, self.lm_head.weight.t(), out=st.logits)

    def _pooled_cache(self, B, t_need):
        """generate() owns its cache, so the slab is reused across calls: stable pointers keep the captured hipGraphs valid."""
        cfg = self.config
        c = getattr(self, "_cache_pool", None)
        if c is None or c.batch != B or c.t_cap < t_need or c.dtype != self.dtype or c.sparse_layer != cfg.sparse_config["sparse_layer"]:
            # slots are allocated in steps of 128: a stream of requests of slightly different lengths (VQAL:123-196) keeps ONE slab -- and with it
            # every captured graph that holds pointers into it -- instead of re-allocating whenever a prompt is a few tokens longer than any before
            old_ptr = None if c is None else c.slab.data_ptr()
            c = None
            self._cache_pool = None
            c = KVSlabCache(cfg.num_hidden_layers, cfg.sparse_config["sparse_layer"], B, cfg.num_key_value_heads, cfg.head_dim, -(-int(t_need) // 128) * 128, self.dtype, self.device)
            self._cache_pool = c
            if old_ptr is not None:  # graphs captured on the slab that has just been freed can never be replayed again
                self._prefill_graphs = {k: v for k, v in self._prefill_graphs.items() if old_ptr not in k}
                if self._dstate is not None:
                    self._dstate.graphs = {k: v for k, v in self._dstate.graphs.items() if old_ptr not in k}
        c.lens.zero_()
        c.full_len_host = [0] * B
        c.seen_tokens = 0
        c.logical_cap = int(t_need)  # a pooled (possibly larger) slab must compute exactly like a fresh one of the requested size
        c.sparse_cap = c.logical_cap
        c.set_bounds(None, None)
        return c

    def _single_split_max_keys(self, st):
        """-> (largest row, in keys, that the fused q|k|v + attention launch takes; largest row it takes with ONE attention workgroup per head).
        Stand-alone launches: 256 keys as one workgroup per (row, head) (cache.py).  Inside dl_gemv_qkv_attn the slab part of the attention runs while
        the q|k|v weights still stream, so the break-even against `dl_gemv` + a split launch moves out with the stream's length, and further with
        several attention workgroups per head (round 4).  tools/bench_qkv_attn.py on 1x MI355X, one launch with 1 / 4 workgroups per head vs the two
        launches: 7B (100.7 MB of q|k|v, 17.8 us) 22.8 / 23.4 vs 26.8 at 256 keys, 25.2 / 23.6 vs 28.0 at 448, 27.8 / 25.7 vs 28.5 at 640, 29.1 / 28.2 vs
        28.3 at 768; 13B (157 MB, 28 us) 32.2 / 33.6 vs 37.5 at 384, 36.0 / 34.1 vs 38.1 at 640, 36.5 / 36.5 vs 39.0 at 768, 39.7 / 42.1 vs 39.9 at 1024."""
        from .cache import _SINGLE_SPLIT_MAX_KEYS
        if self.single_split_keys_override is not None:  # tests: force the schedule to change inside a short generation
            return int(self.single_split_keys_override), int(self.single_split_keys_override)
        if not (self.fuse_qkv_attn and st.B == 1 and st.use_gemv and st.qa_gran is not None):
            return _SINGLE_SPLIT_MAX_KEYS, _SINGLE_SPLIT_MAX_KEYS
        w = self.model.layers[0].w_qkv
        big = (w.numel() * w.element_size()) >= 130e6  # 13B-class stream
        if self.fused_attn_max_splits <= 1:
            return (576 if big else 384), (576 if big else 384)
        return (768 if big else 704), (576 if big else 256)

    def _width_bucket(self, W: int, n_feat: int) -> int:
        """Prompt-width bucket of the device-layout prefill: the smallest width >= W whose COMPACTED row count (W - 1 + kept image tokens: the
        M of 30 of the 32 layers' GEMMs) is a multiple of `prefill_width_bucket` -- 16 by default, one MFMA tile of rows, so a bucket never
        adds a row tile to those GEMMs that the true width would not have needed.  0 / 1 disables bucketing."""
        g = int(self.prefill_width_bucket or 0)
        if g <= 1:
            return W
        sc = self.config.sparse_config
        kept = int(n_feat * sc["vision_keep_rate"]) if (sc["use_vision_predictor"] and sc["sparse_layer"] < self.config.num_hidden_layers) else n_feat
        rows = W - 1 + kept
        return W + (-rows) % g

    def _evict_prefill_entries(self):
        """Bound the prefill-shape cache: at most `max_prefill_graphs` captured graphs and as many seen-once entries (oldest first)."""
        cap = self.max_prefill_graphs
        graphs = [k for k, e in self._prefill_graphs.items() if e["graph"] is not None]
        seen = [k for k, e in self._prefill_graphs.items() if e["graph"] is None]
        for k in graphs[: max(0, len(graphs) - cap + 1)] + seen[: max(0, len(seen) - cap + 1)]:
            self._prefill_graphs.pop(k)

    def _get_dstate(self, B, out_cap):
        st = self._dstate
        if st is None or st.B != B or st.out_ids.shape[1] < out_cap:
            st = self._dstate = _DecodeState(self, B, self.device, self.dtype, out_cap)
            self._prefill_graphs = {}
        return st

    @staticmethod
    def _capture(fn, warm):
        """Warm `fn` up on a side stream (lazy hipBLASLt / allocator state), then capture it into a hipGraph."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            warm()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    def _run_decode_steps(self, st, cache, n_steps):
        """Enqueue n greedy steps (graph replay when enabled)."""
        # what the captured launches depend on: the slab (pointers, strides), the split-KV factor of each length group (the only thing
        # the REQUESTED capacity changes -- keying on logical_cap / sparse_cap themselves would re-capture for every new prompt
        # length of a variable-length workload such as the VQA loader), tables, stop ids and the switches that pick kernels
        cfg = self.config
        nH, SL = cfg.num_attention_heads, cfg.sparse_config["sparse_layer"]
        splits = (cache.n_splits(0, st.B * nH), cache.n_splits(min(SL, cfg.num_hidden_layers - 1), st.B * nH), cache.n_splits(cfg.num_hidden_layers - 1, st.B * nH))
        fused_ns = (cache.fused_attn_splits(0, self.fused_attn_max_splits), cache.fused_attn_splits(cfg.num_hidden_layers - 1, self.fused_attn_max_splits)) if (st.B == 1 and st.qa_gran is not None) else (1, 1)
        key = (cache.slab.data_ptr(), cache.t_cap, splits, fused_ns, self._rope[0].data_ptr(), self._eos, self._pad, getattr(self, "_min_new", 0),
               repr(cfg.sparse_config), self.attn_inkernel_combine, self.tp_side_stream, self.smallm_max_decode_batch, self.gemv_max_decode_batch, self.fuse_qkv_attn, self.fuse_gu_tp, KVSlabCache.eight_wave_single_split,
               self.fused_attn_max_splits, self.qkv_attn_grid_cap, self.gu_grid_cap, self.packed_decode_qkv_min_batch)
        if not self.use_hip_graph:
            for _ in range(n_steps):
                self._decode_step_kernels(st, cache, True)
            return
        g = st.graphs.get(key)
        if g is None:
            # the warm-up executes one real step: snapshot / restore the state it advances
            snap = (st.cur_ids.clone(), st.out_ids.clone(), st.step.clone(), st.finished.clone(), cache.lens.clone(), st.decision.clone())

            def warm():
                self._decode_step_kernels(st, cache, True)

            g, _ = self._capture(lambda: self._decode_step_kernels(st, cache, True), warm)
            st.cur_ids.copy_(snap[0]); st.out_ids.copy_(snap[1]); st.step.copy_(snap[2]); st.finished.copy_(snap[3]); cache.lens.copy_(snap[4]); st.decision.copy_(snap[5])
            if len(st.graphs) >= 12:  # a long generation walks through a few split factors as its rows grow (one capture each)
                st.graphs.pop(next(iter(st.graphs)))
            st.graphs[key] = g
        for _ in range(n_steps):
            g.replay()

    # ---- public API -----------------------------------------------------------------------------
    @torch.no_grad()
    def forward(
        self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
        output_attentions=None, output_hidden_states=None, images=None, image_sizes=None, return_dict=None, input_embeds_indices=None,
        image_features=None,
    ):
        """dynamic_llava_llama.py:68-115 + dynamic_modeling_llama.py:2631-2813 (inference; no labels / loss).
        logits: fp32 [B, N', V] for ALL positions like the reference (DML:2709-2710); rows are right-padded with
        zeros when their lengths differ.

        Host synchronisation of DECODE steps (a one-token call on a non-empty cache): the step's launches are scheduled by the same rule generate()
        follows, which reads the evicted layers' longest row back from the device ONCE PER CHUNK of steps -- steps 1, 5, 9, then every
        `decode_sync_every` (8) steps block on a device->host copy and check the error word of the fused launches (check_device_errors()).  All other
        steps enqueue and return.  A caller that captures forward() steps in its own hipGraph / stream pipeline sets `model.decode_sync_every = 0`:
        then no decode step ever reads the device (launches are sized for the longest possible rows; logits in the same rounding class, not
        bit-identical to generate()'s), and calling check_device_errors() at a convenient sync point is the caller's job."""
        self._check_ready()
        if labels is not None:
            raise NotImplementedError("labels / loss are training-side (DML:2713-2800), out of scope")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not produced by the fused path")
        if use_cache is False:
            return self._forward_nocache(input_ids, attention_mask, past_key_values, inputs_embeds, images, image_features, input_embeds_indices)
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You have to specify either input_ids or inputs_embeds")  # DML:1686-1695
        cache = past_key_values
        if cache is not None and not isinstance(cache, KVSlabCache):
            cache = KVSlabCache.from_legacy_cache(cache, self.config.sparse_config["sparse_layer"], device=self.device)
        decode = cache is not None and max(cache.full_len_host) > 0
        if decode:
            if input_ids is None:
                raise NotImplementedError("inputs_embeds on a non-empty cache")
            if input_ids.shape[1] != 1:
                return self._forward_chunk(input_ids, attention_mask, cache)
            B = input_ids.shape[0]
            st = self._get_dstate(B, 0)
            (cache.single_split_max_keys, cache.fused_single_keys), cache.min_keys_per_split = self._single_split_max_keys(st), self.min_keys_per_split
            cache.ensure_capacity(2)
            self._rope_tables(max(cache.full_len_host) + 2)
            st.cur_ids.copy_(input_ids[:, 0])
            self._eos, self._pad = -1, 0
            # the decode schedule (KVSlabCache.sched_*): the same rule generate() follows, evaluated step by step -- a forward()-driven loop
            # replays the same kernels as generate() on the same request.  Costs one device->host copy of the lengths per CHUNK of steps
            if self.decode_sync_every <= 0:
                # sync-free decode steps: nothing is read back from the device -- the launches are scheduled from the host-known dense
                # length (every row of the evicting layers is at most that long).  For callers that capture forward() steps in their own graph or
                # pipeline; the kernels chosen are the ones for the LONGEST possible rows, so a request decoded this way and through generate()
                # may differ in the last bits of its logits (same rounding class).  check_device_errors() is then the caller's job.
                if cache.sched_active():
                    cache.sched_drop()
                cache.set_bounds(max(cache.full_len_host) + 1, max(cache.full_len_host) + 1)
            else:
                if not cache.sched_active():  # a cache of unknown history (imported legacy tuple, a chunk appended): start from what is there now
                    cache.sched_begin(max(cache.full_len_host), int(cache.lens[1].max()), self.decode_sync_every)
                if cache.sched_at_boundary():
                    if cache._sch["chunks"] > 0:
                        cache.sched_observe(cache._sch["produced"], int(cache.lens[1].max()))
                        self.check_device_errors()  # the queue has just been drained anyway: a fused launch that gave up must not go unnoticed in a forward() loop either
                    cache.sched_chunk()
            st.attn_ws.zero_()  # callers may interleave caches at equal positions on this state: clear the merge granules every call (see generate())
            if st.qa_gran is not None:
                st.qa_gran.zero_()
            if st.tp_gran is not None:
                st.tp_gran.zero_()
            self._decode_step_kernels(st, cache, False)
            sc_ = self.config.sparse_config
            use_tp = bool(sc_["use_text_predictor"] and sc_["use_output_text_predictor"]) and sc_["sparse_layer"] < self.config.num_hidden_layers
            cache.lens[0] += 1
            dec_ = st.decision if use_tp else 1
            if use_tp and self.force_text_decision is not None:
                # test hook (the oracle has the same one): continue a comparison past a keep/evict logit pair that sits on the decision boundary.
                # Only the bookkeeping is overridden -- the step's own logits never depend on its decision (the token always attends itself,
                # DML:1061-1076); debug_records keeps the predictor's own decision and logits
                dec_ = torch.as_tensor(self.force_text_decision).to(device=self.device, dtype=torch.int32).reshape(st.decision.shape)
            cache.lens[1] += dec_
            cache.full_len_host = [n + 1 for n in cache.full_len_host]
            cache.seen_tokens += 1
            if cache.sched_active():
                cache.sched_advance(1)
            if self.debug_records is not None:
                self.debug_records.update(text_decision=st.decision.clone() if use_tp else None, text_logit=st.tp_logits.clone())
            logits = st.logits.to(torch.float32, copy=True).unsqueeze(1)  # never alias the persistent step buffer
            return CausalLMOutputWithPast(logits=logits, past_key_values=cache)
        # ---- prefill ----
        if inputs_embeds is not None:
            embeds, lens, indices = self._unpad_embeds(inputs_embeds, attention_mask, input_embeds_indices)
        else:
            embeds, lens, indices = self._prepare_packed(input_ids, attention_mask, None, images, image_features)
        x, cache, lens2, cu_list = self._prefill(embeds, lens, indices, cache, reserve=256, last_only=False)
        logits_packed = F.linear(x, self.lm_head.weight).float()[: cu_list[-1]]  # (x may be sized for a width bucket: rows past the last sequence are padding)
        B = len(lens2)
        if len(set(lens2)) == 1:
            logits = logits_packed.view(B, lens2[0], -1)
        else:
            logits = logits_packed.new_zeros((B, max(lens2), logits_packed.shape[-1]))
            for b in range(B):
                logits[b, : lens2[b]] = logits_packed[cu_list[b] : cu_list[b + 1]]
        return CausalLMOutputWithPast(logits=logits, past_key_values=cache)

    def _unpad_embeds(self, inputs_embeds, attention_mask, input_embeds_indices):
        """Padded [B, N, H] embeddings (what prepare_inputs_labels_for_multimodal returns: right- OR left-padded, ARCH:529-579) ->
        packed rows + per-row lengths + row-relative segment dicts."""
        B, N = inputs_embeds.shape[:2]
        if attention_mask is None:
            return inputs_embeds.reshape(B * N, -1).to(self.dtype).contiguous(), [N] * B, input_embeds_indices
        am = attention_mask.bool()
        lens = am.sum(dim=1).tolist()
        first = am.int().argmax(dim=1).tolist()  # first valid column of every row (0 when right-padded)
        embeds = torch.cat([inputs_embeds[b, first[b] : first[b] + lens[b]] for b in range(B)], dim=0).to(self.dtype).contiguous()
        indices = input_embeds_indices
        if indices is not None and any(first):
            indices = [None if ix is None else {k: [v[0] - first[b], v[1] - first[b]] for k, v in ix.items()} for b, ix in enumerate(indices)]
        return embeds, lens, indices

    def _forward_nocache(self, input_ids, attention_mask, past_key_values, inputs_embeds, images, image_features, input_embeds_indices):
        """`model(total_input_ids, images=..., use_cache=False)` -- the whole sequence is re-run every step
        (llava/dynamic_eval/bench_test/dynamic_llava_long_text_time_with_no_cache.py:336-343); no cache is returned."""
        if past_key_values is not None:
            raise NotImplementedError("use_cache=False with past_key_values")
        cfg, sc = self.config, self.config.sparse_config
        if inputs_embeds is not None:
            embeds, lens, indices = self._unpad_embeds(inputs_embeds, attention_mask, input_embeds_indices)
        else:
            embeds, lens, indices = self._prepare_packed(input_ids, attention_mask, None, images, image_features)
        p = self._plan_prefill(lens, indices)
        p["nocache"] = True
        need = max(lens) + 2
        c = getattr(self, "_scratch_cache", None)  # K/V are still written (the kernels are fused), into a scratch slab that is dropped
        if c is None or c.batch != p["B"] or c.t_cap < need or c.dtype != self.dtype:
            c = self._scratch_cache = KVSlabCache(cfg.num_hidden_layers, sc["sparse_layer"], p["B"], cfg.num_key_value_heads, cfg.head_dim, need + 64, self.dtype, self.device)
        self._rope_tables(need)
        x = self._prefill_run(p, embeds, c, indices, False)
        lens2 = p["nocache_lens"] or p["lens2"]
        if len(set(lens2)) != 1:
            raise NotImplementedError("use_cache=False expects equally long rows")
        logits = F.linear(x, self.lm_head.weight).float().view(p["B"], lens2[0], -1)
        return CausalLMOutputWithPast(logits=logits, past_key_values=None)

    def _forward_chunk(self, input_ids, attention_mask, cache: KVSlabCache):
        """T > 1 new tokens on a non-empty cache -- the multi-round "new instruct" call (DML:2506-2521: the instruct
        predictor decides which of the chunk's tokens are stored in layers >= sparse_layer, the last one always) or, without the
        instruct predictor, plain chunked prefill.  Every chunk token attends to the cache and causally to the chunk
        (CU:256-268 `get_cache`), then only the kept K/V rows stay in the slab (CU:165-241, without the zero padding)."""
        cfg, sc = self.config, self.config.sparse_config
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            raise NotImplementedError("padded chunks on a cache")
        dev, dt = self.device, self.dtype
        B, T = input_ids.shape
        nH, nKV, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        eps, L, SL = cfg.rms_norm_eps, cfg.num_hidden_layers, sc["sparse_layer"]
        instruct = bool(sc["use_text_predictor"] and sc["use_instruct_predictor"]) and SL < L
        cache.ensure_capacity(T + 1)
        cos, sin = self._rope_tables(max(cache.full_len_host) + T + 1)
        # no device->host copy (the reference syncs per row per layer, CU:197-199): the un-evicted length bounds both length groups
        bound = max(cache.full_len_host) + T
        total = B * T
        cu = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=dev)
        h = self.model.embed_tokens(input_ids.reshape(-1).to(dev)).clone()
        x = ops.rmsnorm(h, self.model.layers[0].input_layernorm.weight, eps)
        keep_idx = None
        for i, layer in enumerate(self.model.layers):
            if i == SL and instruct:
                tp = self.model.instruct_score_predictor
                dec = torch.empty(total, dtype=torch.int32, device=dev)
                lg = torch.empty((total, 2), dtype=torch.float32, device=dev)
                tp.decide(h, ops.text_predictor_workspace(total, tp.d_model, dev), lg, dec)
                dec = dec.view(B, T)
                dec[:, -1] = 1  # DML:2521
                keep_idx = dec.contiguous()  # int32 [B, T] on the device: which chunk rows stay in layers >= SL
                if self.debug_records is not None:
                    self.debug_records.update(text_decision=dec.clone(), text_logit=lg.view(B, T, 2).clone())
            g = cache.group(i)
            lens = cache.lens[g]
            qkv = F.linear(x, layer.w_qkv)
            ops.rope_kv_write(qkv, cos, sin, cu, None, cache.len_full, lens, cache.k[i], cache.v[i], nH, nKV, d)
            attn = torch.empty((total, nH * d), dtype=dt, device=dev)
            ops.attn_prefill_cached(qkv[:, : nH * d], cache.k[i], cache.v[i], lens, attn, cu, T, bound, nH, nKV, d)
            o = F.linear(attn, layer.self_attn.o_proj.weight)
            x = ops.add_rmsnorm(h, o, layer.post_attention_layernorm.weight, eps)
            act = ops.silu_mul(F.linear(x, layer.w_gu))
            dn = F.linear(act, layer.mlp.down_proj.weight)
            nw = self.model.norm.weight if i + 1 == L else self.model.layers[i + 1].input_layernorm.weight
            x = ops.add_rmsnorm(h, dn, nw, eps)
        cache.lens[0] += T
        if keep_idx is not None:
            # keep only the chosen rows of this chunk, packed in place right after the old ones: ONE launch for all layers >= SL
            # (every layer's attention has already read its un-packed chunk rows), then the kept counts are added on the device
            ops.kv_pack_rows(cache.k[SL], cache.v[SL], cache.slab.stride(0), L - SL, keep_idx, cache.lens[1], cache.t_cap)
            cache.lens[1] += keep_idx.sum(dim=1).to(torch.int32)
        else:
            cache.lens[1] += T
        cache.full_len_host = [n + T for n in cache.full_len_host]
        cache.seen_tokens += T  # cache.sparse_cap stays a valid (host-known) upper bound of the evicted group's lengths
        cache.sched_drop()  # the next decode step re-starts the schedule from the lengths it finds
        logits = F.linear(x, self.lm_head.weight).float().view(B, T, -1)
        return CausalLMOutputWithPast(logits=logits, past_key_values=cache)

    def _first_token(self, st, x_last, min_new):
        if st.use_gemv and x_last.dim() == 2 and x_last.is_contiguous():
            # up to three rows: the weight-streaming GEMV the decode steps use for the same matrix (41 vs 61 us for the library's skinny GEMM at B=1)
            ops.gemv(self.lm_head.weight, st.logits, x=x_last)
        else:
            torch.matmul(x_last, self.lm_head.weight.t(), out=st.logits)
        self._prefill_logits_buf.copy_(st.logits)
        # first token: argmax only (the prompt's KV lengths are already in place); EOS is banned while step < min_new (HF semantics)
        ops.decode_advance(st.logits, st.cur_ids, st.out_ids, st.step, st.finished, self._eos, self._pad, None, None, None, min_new_tokens=min_new)

    def _gen_kwargs(self, kwargs, lens):
        """Shared parsing of the HF generate() kwargs this path honours (DLL:117-152 forwards **kwargs to HF): rejects what is not
        built instead of silently ignoring it."""
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")  # DLL:128-129
        if (kwargs.get("num_beams", 1) or 1) != 1:
            raise NotImplementedError("beam search is not built (harness default num_beams=1)")
        max_new = kwargs.get("max_new_tokens")
        if max_new is None:
            max_new = 20 if kwargs.get("max_length") is None else int(kwargs["max_length"]) - max(lens)
        if max_new < 1:
            raise ValueError(f"max_new_tokens must be >= 1 (got {max_new})")
        eos = kwargs.get("eos_token_id", self.config.eos_token_id)
        if isinstance(eos, (list, tuple)):
            eos = [int(e) for e in eos]
            eos = None if not eos else (eos[0] if len(eos) == 1 else eos)
        elif eos is not None:
            eos = int(eos)
        pad = kwargs.get("pad_token_id", self.config.pad_token_id)
        min_new = int(kwargs.get("min_new_tokens", 0) or 0)
        return int(max_new), min(min_new, int(max_new)), eos, (0 if pad is None else int(pad))

    @torch.no_grad()
    def _generate_sample(self, inputs, images, greedy=False, **kwargs):
        """`greedy=True` (round 4): the same plain loop over forward() taking the argmax -- the route for requests the device-side greedy loop does not
        take (more than three EOS ids: dl_decode_advance compares three).  do_sample=True: temperature / top_k / top_p sampling as HF's logits warpers define them (TemperatureLogitsWarper,
        TopKLogitsWarper, TopPLogitsWarper: the smallest set of most probable tokens whose mass reaches top_p is kept), drawn with
        torch's generator -- a plain loop over forward() (no hipGraph: this is the convenience path, the harness default is greedy).
        Token streams cannot match HF's draw for draw (different RNG consumption); the distribution per step is the same."""
        temperature = float(kwargs.get("temperature", 1.0) or 1.0)
        top_k = int(kwargs.get("top_k", 0) or 0)
        top_p = kwargs.get("top_p")
        gen = kwargs.get("generator")
        out = self.forward(inputs.to(self.device), attention_mask=kwargs.get("attention_mask"), images=images, image_features=kwargs.get("image_features"))
        cache = out.past_key_values
        max_new, min_new, eos, pad = self._gen_kwargs(kwargs, cache.full_len_host)
        eos_set = [] if eos is None else (eos if isinstance(eos, list) else [eos])
        # last VALID position of every row: the returned logits are right-padded with zeros when rows differ in length, and the
        # last layer's KV length is the (compacted) row length
        last = cache[1][-1].to(self.device).long() - 1
        logits = out.logits[torch.arange(out.logits.shape[0], device=self.device), last]
        self.last_prefill_logits = logits.float().clone()
        B = logits.shape[0]
        finished = torch.zeros(B, dtype=torch.bool, device=self.device)
        toks, scores = [], []
        for step in range(max_new):
            z = logits.float() / temperature
            if step < min_new and eos_set:
                z[:, eos_set] = float("-inf")
            if top_k > 0:
                kth = torch.topk(z, min(top_k, z.shape[-1]), dim=-1).values[:, -1:]
                z = z.masked_fill(z < kth, float("-inf"))
            if top_p is not None and float(top_p) < 1.0:
                sz, si = torch.sort(z, dim=-1, descending=False)
                cum = sz.softmax(dim=-1).cumsum(dim=-1)
                remove = cum <= (1.0 - float(top_p))
                remove[:, -1] = False  # always keep the most probable token
                z = z.masked_fill(remove.scatter(1, si, remove), float("-inf"))
            scores.append(z)
            nxt = z.argmax(dim=-1) if greedy else torch.multinomial(z.softmax(dim=-1), 1, generator=gen)[:, 0]
            nxt = torch.where(finished, torch.full_like(nxt, pad), nxt)
            toks.append(nxt)
            for e in eos_set:
                finished = finished | (nxt == e)
            if eos_set and bool(finished.all()):
                break
            if step + 1 < max_new:
                out = self.forward(nxt[:, None], past_key_values=cache)
                cache = out.past_key_values
                logits = out.logits[:, -1]
        seq = torch.stack(toks, dim=1)
        self.last_cache = cache
        if kwargs.get("return_dict_in_generate"):
            res = {"sequences": seq, "past_key_values": cache}
            if kwargs.get("output_scores"):
                res["scores"] = tuple(scores)
            return res
        return seq

    @torch.no_grad()
    def _generate_on_cache(self, inputs, images, **kwargs):
        """generate(new_ids, past_key_values=cache): greedy continuation of a dialogue whose earlier turns (image included) are in `cache` (what
        a previous call returned: `return_dict_in_generate=True`, or forward()).  `inputs` holds ONLY the new turn's token ids [B, T] -- with
        the image placeholder of the first turn the reference's `input_ids[:, past_length:]` slicing (DML:2831-2853) has no consistent meaning,
        and its own multi-round harness drives forward() (model_lvis_multi_round_for_ppl.py:108-220).  The new chunk goes through the
        chunk-on-cache path (DML:2506-2521: the instruct predictor decides which of its K/V rows stay in layers >= sparse_layer), the new
        tokens through single decode steps; plain loop over forward(), no hipGraph."""
        if images is not None or kwargs.get("image_features") is not None:
            raise NotImplementedError("generate(past_key_values=...) continues a dialogue: the image belongs to the first call")
        cache = kwargs["past_key_values"]
        if not isinstance(cache, KVSlabCache):
            cache = KVSlabCache.from_legacy_cache(cache, self.config.sparse_config["sparse_layer"], device=self.device)
        inputs = inputs.to(self.device)
        if inputs.shape[0] != cache.batch:
            raise ValueError(f"{inputs.shape[0]} rows of new tokens for a cache of {cache.batch} rows")
        if bool((inputs == IMAGE_TOKEN_INDEX).any()):
            # HF's convention (prepare_inputs_for_generation, DML:2835-2848) passes the FULL dialogue and slices input_ids[:, past_length:]; with
            # the image placeholder in the ids and 576 features in the cache that slice is meaningless, so it is refused instead of appending
            # the whole prompt to the cache a second time
            raise ValueError("generate(past_key_values=...) takes ONLY the new turn's token ids; these ids contain the image placeholder, i.e. the full dialogue "
                             "(the first turn, image included, is already in the cache)")
        # max_length counts what is already cached plus the new turn (the reference's own length accounting includes the 576 image tokens)
        max_new, min_new, eos, pad = self._gen_kwargs({k: v for k, v in kwargs.items() if k != "past_key_values"}, [cache.seen_tokens + inputs.shape[1]])
        eos_set = [] if eos is None else (eos if isinstance(eos, list) else [eos])
        out = self.forward(inputs, attention_mask=kwargs.get("attention_mask"), past_key_values=cache)
        cache = out.past_key_values
        logits = out.logits[:, -1].float()
        self.last_prefill_logits = logits.clone()
        finished = torch.zeros(inputs.shape[0], dtype=torch.bool, device=self.device)
        toks, scores = [], []
        for step in range(max_new):
            z = logits.clone()
            if step < min_new and eos_set:
                z[:, eos_set] = float("-inf")
            scores.append(z)
            nxt = z.argmax(dim=-1)
            nxt = torch.where(finished, torch.full_like(nxt, pad), nxt)
            toks.append(nxt)
            for e in eos_set:
                finished = finished | (nxt == e)
            if eos_set and bool(finished.all()):
                break
            if step + 1 < max_new:
                out = self.forward(nxt[:, None], past_key_values=cache)
                cache = out.past_key_values
                logits = out.logits[:, -1].float()
        seq = torch.stack(toks, dim=1)
        self.last_cache = cache
        if kwargs.get("return_dict_in_generate"):
            res = {"sequences": seq, "past_key_values": cache}
            if kwargs.get("output_scores"):
                res["scores"] = tuple(scores)
            return res
        return seq

    @torch.no_grad()
    def generate(self, inputs=None, images=None, image_sizes=None, **kwargs):
        """dynamic_llava_llama.py:117-152: greedy decoding; returns the NEW tokens only [B, T_new] (HF behaviour when
        generation is driven by inputs_embeds).  Supported kwargs: max_new_tokens / max_length, min_new_tokens (EOS banned from
        the argmax until then, as HF's MinNewTokensLengthLogitsProcessor), do_sample (True: temperature / top_k / top_p sampling
        through _generate_sample), num_beams(1), use_cache(True), eos_token_id, pad_token_id, attention_mask,
        return_dict_in_generate (+ output_scores), image_features (pre-computed projector output, testing).
        Steady state (same prompt SHAPE as a previous call): the whole prefill -- CLIP, projector, embedding assembly,
        32 layers, first-token argmax -- is one hipGraph replay and every decode step is another; the host only copies
        the new token ids / pixels into static buffers."""
        self._check_ready()
        if kwargs.get("do_sample", False):  # model_vqa_loader.py:162-175 passes do_sample = temperature > 0 (default 0: greedy)
            if "inputs_embeds" in kwargs:
                raise NotImplementedError("`inputs_embeds` is not supported")  # DLL:128-129
            return self._generate_sample(inputs, images, **kwargs)
        if kwargs.get("past_key_values") is not None:
            return self._generate_on_cache(inputs, images, **kwargs)
        attention_mask = kwargs.get("attention_mask")
        image_features = kwargs.get("image_features")
        sync_every = int(kwargs.get("sync_every", self.decode_sync_every))  # decode steps enqueued between two observations of the device state
        want_dict = bool(kwargs.get("return_dict_in_generate"))
        want_scores = want_dict and bool(kwargs.get("output_scores"))
        inputs = inputs.to(self.device)
        tm = None
        if self.record_timing:
            tm = {"ev": [torch.cuda.Event(enable_timing=True) for _ in range(3)], "path": None}
            tm["ev"][0].record()
        n_feat = self._n_feat(images, image_features)
        sc_ = self.config.sparse_config
        vp_ = getattr(self.model, "image_score_predictor", None)
        dev_layout = (self.device_prompt_layout and not kwargs.get("_host_layout") and attention_mask is None and n_feat > 0 and inputs.shape[1] >= 1
                      and self.use_hip_graph and self.debug_records is None and not (sc_["use_text_predictor"] and sc_["use_instruct_predictor"])
                      and not (vp_ is not None and (len(vp_._forward_hooks) or len(vp_._forward_pre_hooks)))
                      and getattr(self.config, "tokenizer_model_max_length", None) is None)
        if dev_layout:
            # ARCH:309-490 on the device.  Every row is assumed to hold exactly one image token (checked by the kernel; a
            # violation is seen at the final synchronisation and the call is repeated with the host layout): all shapes then follow
            # from (B, W), and where the image sits is the kernel's business, inside the captured graph.
            # Round 4: W is rounded up to a WIDTH BUCKET -- the captured launches are sized for the bucket, the true width travels as a device
            # scalar and the layout kernel packs the sequences at their true lengths (cu_seqlens, KV lengths, last rows: device memory the
            # launches already read).  A request stream with a new width on every call (VQAL:123-196) then replays a handful of graphs.
            Bq, Wq = inputs.shape
            Wb = self._width_bucket(Wq, n_feat)
            n_row, n_row_b = Wq - 1 + n_feat, Wb - 1 + n_feat
            fake = [{"system": [0, 0], "image": [0, n_feat], "instruct": [n_feat, n_row_b], "answer": [n_row_b, n_row_b], "last_instruct": [n_feat, n_row_b]} for _ in range(Bq)]
            lay = dict(sig=("dev", Bq, Wb, n_feat), B=Bq, lens=[n_row] * Bq, lens_bucket=[n_row_b] * Bq, indices=fake, text_src=[0], text_dst=[0], img_dst=[0], img_rows=list(range(Bq)),
                       total=Bq * n_row_b, n_feat=n_feat, width=Wq, bucket=Wb)
        else:
            lay = self._layout(inputs, attention_mask, None, n_feat)
        lens, indices, B = lay["lens"], lay["indices"], lay["B"]
        max_new, min_new, eos, pad = self._gen_kwargs(kwargs, lens)
        if isinstance(eos, list) and len(eos) > 3:
            # three ids are compared on the device; a longer EOS set takes the plain forward() loop (same kernels per step, host-side stop test)
            return self._generate_sample(inputs, images, greedy=True, **{k: v for k, v in kwargs.items() if k not in ("do_sample", "sync_every")})
        cache = self._pooled_cache(B, max(lens) + max_new + 1)
        self._rope_tables(max(lens) + max_new + 1)
        st = self._get_dstate(B, max_new)
        (cache.single_split_max_keys, cache.fused_single_keys), cache.min_keys_per_split = self._single_split_max_keys(st), self.min_keys_per_split
        st.step.zero_(); st.finished.zero_(); st.decision.fill_(1)
        # the in-kernel split merge of the decode attention validates its granules by tag = (position of the new token, layer): within one
        # request positions only grow, so a slot left by an earlier step never matches -- but a slot left by an EARLIER REQUEST at the same
        # position would.  Tag 0 is never expected: one clear per request makes every older granule unmatchable.
        st.attn_ws.zero_()
        if st.qa_gran is not None:
            st.qa_gran.zero_()  # and of dl_gemv_qkv_attn
        if st.tp_gran is not None:
            st.tp_gran.zero_()  # and of dl_gemv_gu_tp
        self._eos = -1 if eos is None else (tuple(eos) if isinstance(eos, list) else eos)  # one id or a tuple of up to three (the EOS set)
        self._pad = pad
        self._min_new = min_new
        if getattr(self, "_prefill_logits_buf", None) is None or self._prefill_logits_buf.shape != st.logits.shape:
            self._prefill_logits_buf = torch.empty(st.logits.shape, dtype=torch.float32, device=self.device)
        vp = getattr(self.model, "image_score_predictor", None)
        hooked = vp is not None and (len(vp._forward_hooks) or len(vp._forward_pre_hooks))
        graphable = self.use_hip_graph and self.debug_records is None and not hooked  # (the instruct predictor's data-dependent row count stays on the device)
        prefill_path = "eager"
        if graphable:
            # the instruct predictor's compaction span is baked into the captured plan, and `sig` ignores token values: two
            # equally long multi-turn prompts whose last "USER:" sits elsewhere must not share a graph
            li_key = tuple(tuple(ix["last_instruct"]) for ix in indices) if self._instruct_on(indices, B) else None
            key = (lay["sig"], None if images is None else tuple(images.shape), None if image_features is None else tuple(image_features.shape),
                   cache.slab.data_ptr(), cache.t_cap, self._rope[0].data_ptr(), self._eos, self._pad, min_new, repr(self.config.sparse_config), li_key,
                   self._prefill_knob_key())  # (a knob toggled at run time must not replay a graph captured under the other setting)
            ent = self._prefill_graphs.get(key)
            if ent is None:
                # first sighting of a prompt shape: run it eagerly ONCE, through the same closure a capture would record (a request stream
                # such as the VQA loader's, VQAL:123-196, presents many widths; capturing each on sight cost two prefills + a capture per miss)
                self._evict_prefill_entries()
                ids0 = inputs.contiguous().clone()
                if dev_layout and lay["bucket"] != lay["width"]:
                    ids0 = torch.zeros((B, lay["bucket"]), dtype=inputs.dtype, device=self.device)
                ent = dict(ids=ids0, images=None if images is None else images.to(self.device).clone(),
                           feats=None if image_features is None else image_features.to(self.device).clone(),
                           plan=self._plan_prefill(lay.get("lens_bucket", lens), indices), indices=copy.deepcopy(indices), graph=None)
                ent["plan"]["device_instruct"] = True
                if dev_layout:
                    ent["w_true"] = torch.zeros(1, dtype=torch.int32, device=self.device)
                    ent["didx"] = ops.prompt_layout(ent["ids"], n_feat, IMAGE_TOKEN_INDEX, USER_IDS)  # allocates the outputs (its launch saw an empty bucket buffer:
                    ent["didx"]["err"].zero_()                                                        # forget that verdict)
                    ent["plan"]["img_start"] = ent["didx"]["img_start"]  # written by the layout kernel inside the graph
                else:
                    ent["didx"] = self._dev_idx(lay)

                def run(ent=ent, lay=lay, dev_layout=dev_layout, n_feat=n_feat, min_new=min_new, cache=cache, st=st):
                    if dev_layout:
                        p_ = ent["plan"]  # the plan's shapes are the bucket's; its device metadata is (re)written here at the true width
                        ops.prompt_layout_into(ent["ids"], n_feat, IMAGE_TOKEN_INDEX, USER_IDS, ent["didx"], w_true=ent["w_true"], n_drop=(p_["n_img"] - p_["k"]) if p_["vision_on"] else 0,
                                               cu=p_["cu"], cu2=p_["cu2"], lens=p_["lens_dev"], last_rows=p_["last_rows"])
                    f = ent["feats"] if ent["feats"] is not None else (self.encode_images(ent["images"]) if ent["images"] is not None else None)
                    emb = self._assemble(lay, ent["didx"], ent["ids"], f)
                    x = self._prefill_run(ent["plan"], emb, cache, copy.deepcopy(ent["indices"]), True)
                    self._first_token(st, x, min_new)

                ent["run"] = run
                self._prefill_graphs[key] = ent
                if dev_layout:
                    ent["ids"][:, : inputs.shape[1]].copy_(inputs)
                    ent["w_true"].fill_(inputs.shape[1])
                st.step.zero_(); st.finished.zero_()
                run()
            else:
                self._prefill_graphs[key] = self._prefill_graphs.pop(key)  # most recently used last
                ent["ids"][:, : inputs.shape[1]].copy_(inputs)
                if dev_layout:
                    ent["w_true"].fill_(inputs.shape[1])
                if images is not None:
                    ent["images"].copy_(images)
                if image_features is not None:
                    ent["feats"].copy_(image_features)
                if ent["graph"] is None:
                    # second sighting: capture (the eager run of the first sighting was the warm-up: library heuristics, allocator); all
                    # prefill graphs record into ONE memory pool -- they never run concurrently and leave nothing behind in it (logits, ids
                    # and K/V land in persistent buffers), so a hundred cached shapes cost one shape's activations
                    prefill_path = "graph-capture"
                    if self._prefill_pool is None or not any(e["graph"] is not None for e in self._prefill_graphs.values()):
                        # (the allocator drops a private pool with its last graph: a handle whose graphs are all gone must not be reused)
                        self._prefill_pool = torch.cuda.graph_pool_handle()
                    g_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_, pool=self._prefill_pool):
                        ent["run"]()
                    ent["graph"] = g_
                    ent["run"] = None
                else:
                    prefill_path = "graph-replay"
                st.step.zero_(); st.finished.zero_()
                ent["graph"].replay()
            p_host = ent["plan"]
            if dev_layout and lay["bucket"] != lay["width"]:  # host mirrors follow the true lengths, not the bucket's
                drop_ = (p_host["n_img"] - p_host["k"]) if p_host["vision_on"] else 0
                p_host = dict(p_host, lens=list(lens), lens2=[n - drop_ for n in lens] if self.config.sparse_config["sparse_layer"] < self.config.num_hidden_layers else list(lens))
            self._prefill_host_update(p_host, cache, indices)
        else:
            f = image_features if image_features is not None else (self.encode_images(images) if images is not None else None)
            embeds = self._assemble(lay, self._dev_idx(lay), inputs, f)
            x, cache, _, _ = self._prefill(embeds, lens, indices, cache, reserve=max_new + 1, last_only=True)
            self._first_token(st, x, min_new)
        self.last_prefill_logits = self._prefill_logits_buf
        if tm is not None:
            tm["ev"][1].record()
            tm["path"] = prefill_path
        scores = []

        eos_ids = [] if self._eos == -1 else (list(self._eos) if isinstance(self._eos, tuple) else [self._eos])

        def _score(step_idx):  # HF `scores`: the processed logits of that step (EOS at -inf while step < min_new_tokens)
            z = (self._prefill_logits_buf if step_idx == 0 else st.logits).float().clone()
            if step_idx < min_new and eos_ids:
                z[:, eos_ids] = float("-inf")
            scores.append(z)

        if want_scores:
            _score(0)
        # ---- decode loop: chunks of captured steps, scheduled from the lengths the predictor actually leaves (DML:2377-2391, CU:153-164) ----
        # Before a chunk the host knows: the un-evicted group's length exactly, and the evicted group's as OBSERVED after an earlier chunk
        # (non-blocking copy into pinned memory, consumed one chunk late so that the host never waits for steps it has just enqueued) plus
        # the steps enqueued since.  Those bounds pick the split-KV factor / the fused q|k|v+attention launch (cache.n_splits), i.e. which
        # captured graph is replayed; the values are data, never timing, so the schedule is deterministic for a given request.
        produced, chunks = 1, 0
        cache.sched_begin(max(cache.full_len_host), getattr(cache, "prefill_sparse_max", None) or max(cache.full_len_host), sync_every)
        pending = []  # (ring slot, produced-when-copied)
        all_done = False
        while produced < max_new and not all_done:
            # EOS of a short answer (VQA: a few tokens) must not cost two chunks of wasted steps: the first chunks' flags are read blocking (one
            # ~50 us queue drain each), later ones one chunk late.  The SCHEDULE never uses an observation newer than one chunk old either way.
            keep_newest = 0 if (eos_ids and chunks <= 2) else 1
            while len(pending) > keep_newest:
                slot, at = pending.pop(0)
                st.obs_ev[slot].synchronize()
                row = st.obs_host[slot]
                cache.sched_observe(at, int(row[B : 2 * B].max()))
                if eos_ids and int(row[2 * B :].min()) != 0:
                    all_done = True
            if all_done:
                break
            n = min(cache.sched_chunk(), max_new - produced)
            if want_scores:  # the step buffer is read after every step; the schedule (chunks, bounds) is the same as without scores
                for i_ in range(n):
                    self._run_decode_steps(st, cache, 1)
                    _score(produced + i_)
            else:
                self._run_decode_steps(st, cache, n)
            produced += n
            chunks += 1
            cache.sched_advance(n)
            if produced < max_new:
                slot = chunks % 4
                st.obs_host[slot, : 2 * B].copy_(cache.lens.view(-1), non_blocking=True)
                if eos_ids:
                    st.obs_host[slot, 2 * B :].copy_(st.finished, non_blocking=True)
                st.obs_ev[slot].record()
                pending.append((slot, produced))
        if B == 1:
            self.check_device_errors()
        if dev_layout and int(ent["didx"]["err"].item()) != 0:
            # a row without exactly one image token (text-only row, several images): what was computed is meaningless -- repeat the
            # call with the host-side layout, which handles (or rejects) those rows like the reference does
            ent["didx"]["err"].zero_()
            return self.generate(inputs, images=images, image_sizes=image_sizes, _host_layout=True, **kwargs)
        # host mirrors of what the device loop advanced: every row's un-evicted length grows by one per decode step
        cache.full_len_host = [n + produced - 1 for n in cache.full_len_host]
        cache.seen_tokens += produced - 1
        out = st.out_ids[:, :produced].clone()
        if eos_ids:  # HF stops as soon as every row has emitted EOS: trim the columns produced after that
            hit = out == eos_ids[0]
            for e_ in eos_ids[1:]:
                hit = hit | (out == e_)
            fin = hit.int().cumsum(dim=1).clamp(max=1)
            all_done = fin.min(dim=0).values
            if bool(all_done.any().item()):
                first = int(torch.argmax(all_done).item())
                out = out[:, : first + 1]
                scores = scores[: first + 1]
        self.last_cache = cache
        if tm is not None:
            tm["ev"][2].record()
            tm["new_tokens"] = int(out.shape[1])
            self.last_timing = tm
        if want_dict:
            # the caller keeps this cache (the reference returns an independent one per call): detach it from the pool, the next
            # generate() allocates a fresh slab instead of overwriting this one
            self._cache_pool = None
            ptr = cache.slab.data_ptr()  # captured graphs hold raw pointers into the slab the caller now owns (and may free): drop them
            self._prefill_graphs = {k: v for k, v in self._prefill_graphs.items() if ptr not in k}
            st.graphs = {k: v for k, v in st.graphs.items() if ptr not in k}
            res = {"sequences": out, "past_key_values": cache}
            if want_scores:
                res["scores"] = tuple(scores)
            return res
        return out
