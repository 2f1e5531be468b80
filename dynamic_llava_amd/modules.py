"""Parameter tree of the model and the vision side: decoder-layer containers (state-dict keys identical to the reference), the vision / text predictors (DML:1308-1387), the CLIP tower on this package's kernels (clip_encoder.py:7-102).  Split out of model.py in round 6 (no behaviour change)."""
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
        self.wp_o = None  # round 6: o_proj for dl_linear_tiles (row tiles x 8 units x k ranges as fp32 partial sums; +33.5 MB per 7B layer)

    def pack(self, operand_copies: bool = False, o_copy: bool = False):
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
        self.wp_qkv = self.wp_gu = self.wp_down = self.wp_o = None
        if operand_copies and self.w_qkv.dtype in (torch.bfloat16, torch.float16) and self.w_qkv.shape[1] % 64 == 0 and self.w_qkv.shape[0] % 16 == 0 and I % 16 == 0:
            self.wp_qkv = ops.pack_weight_tiles(self.w_qkv)
            self.wp_gu = ops.pack_weight_tiles(self.w_gu, gate_up_pairs=True)
            wo = a.o_proj.weight.data
            if o_copy and ops.linear_tiles_ok(1, wo.shape[0], wo.shape[1], wo.dtype) and wo.shape[0] % 128 == 0:
                self.wp_o = ops.pack_weight_tiles(wo.contiguous())
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
        self.tiles_max_batch = 2  # tools/clip_tower_batch_time.py: 0.85 of the library path at one image, 0.95 at two (160-row tiles, one round of workgroups); 1.26 / 1.17 at three / four
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
        # 16-bit towers of up to 2048 channels: the wave-per-row LayerNorm launches (round 6).  One 256-thread workgroup per 1024-channel row leaves half its threads
        # without a vector: 59 us per launch at 32 images where the bytes need 30 (tools/clip_tower_batch_time.py)
        wave_ln = h.dtype in (torch.bfloat16, torch.float16) and C % 8 == 0 and C <= 2048
        ln, add_ln = (ops.layernorm_rows, ops.add_layernorm_rows) if wave_ln else (ops.layernorm, ops.add_layernorm)
        xn = ln(h, layers[0].layer_norm1.weight, layers[0].layer_norm1.bias, eps) if len(layers) else None
        for i, l in enumerate(layers):
            wq, bq = self._qkv[i]
            qkv = F.linear(xn, wq, bq)
            attn = torch.empty((B * T, C), dtype=h.dtype, device=h.device)
            ops.attn_prefill(qkv[:, :C], qkv[:, C : 2 * C], qkv[:, 2 * C :], attn, cu, T, nH, nH, d, causal=False)
            y = F.linear(attn, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias)
            xn = add_ln(h, y, l.layer_norm2.weight, l.layer_norm2.bias, eps)
            g = ops.quick_gelu(F.linear(xn, l.mlp.fc1.weight, l.mlp.fc1.bias))
            y = F.linear(g, l.mlp.fc2.weight, l.mlp.fc2.bias)
            if i + 1 < len(layers):
                nl = layers[i + 1]
                xn = add_ln(h, y, nl.layer_norm1.weight, nl.layer_norm1.bias, eps)
            else:
                add_ln(h, y)

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
