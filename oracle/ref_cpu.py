"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's sparsified prefill+decode path.

This file restates, in plain eager PyTorch, what Osilly/dynamic_llava computes on its inference hot
path.  It is the *checker* for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg) and is never imported by anything under dynamic_llava_amd/.  Every function cites the reference
lines it follows (paths relative to /root/reference/; DML = llava/model/language_model/
dynamic_modeling_llama.py, CU = .../cache_utils.py, CTL = .../custom_transformer_layer.py,
ARCH = llava/model/dynamic_llava_arch.py, DLL = .../dynamic_llava_llama.py).

Parity status: PINNED.  tests/test_oracle_golden.py checks this file against golden vectors that
oracle/make_golden.py produced by importing and running the reference itself in the build container
(bit-exact on CPU for fp32 and bf16: same torch ops in the same order), and, when /root/reference is
present, tests/test_oracle_vs_reference.py re-runs the reference live.

Pinned choice (the reference leaves it unspecified): top-k ties.  The reference uses a non-stable
`torch.argsort(descending=True)` (DML:1902-1908); `tie_break="stable"` (default) = among equal scores the
lower original index wins.  `tie_break="torch"` calls argsort exactly as the reference does.

Reference quirks reproduced on purpose (so that this file matches the reference, not "what it meant"):
  * B>1 decode zero-pads evicted rows and attends to the pad slots (CU:201-241 + DML:1114-1122);
  * row 0's image range is applied to every row (DML:1917-1933);
  * the 2-D padding mask is honoured at layer 0 only (`attention_mask = None`, DML:2554).
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100
USER_IDS = [11889, 29901]  # ARCH:36 special_text["USER:"]


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def rmsnorm(x, w, eps):
    """DML:134-139 -- fp32 statistics, cast back to the input dtype, THEN multiply by the weight."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_table(head_dim, n_pos, base, dtype, device="cpu"):
    """DML:152-174,181-184 -- fp32 table of cat(freqs, freqs), rounded to the model dtype on use."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float().to(device) / head_dim))
    t = torch.arange(n_pos, device=device, dtype=inv_freq.dtype)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    """DML:253-257"""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2 :]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """DML:260-285 (q,k: [B, nH, T, d]; position_ids [B, T] or [1, T])."""
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def causal_mask_for_sdpa(attention_mask_2d, batch, q_len, past_len, dtype, device):
    """transformers==4.37.2 `_prepare_4d_causal_attention_mask_for_sdpa` as called at DML:1805-1810.
    (third-party, pinned by the reference's pyproject.toml:17; restated from its published source.)
    Returns None (=> SDPA is_causal when q_len > 1) or an additive [B,1,q,kv] mask."""
    kv_len = q_len + past_len
    if attention_mask_2d is not None:
        if bool(torch.all(attention_mask_2d == 1)):
            if q_len == 1 or kv_len == q_len:
                return None
        minv = torch.finfo(dtype).min
        m = torch.full((q_len, q_len), minv, dtype=dtype, device=device)
        cond = torch.arange(q_len, device=device)
        m.masked_fill_(cond < (cond + 1).view(q_len, 1), 0)
        if past_len > 0:
            m = torch.cat([torch.zeros(q_len, past_len, dtype=dtype, device=device), m], dim=-1)
        causal = m[None, None].expand(batch, 1, q_len, kv_len)
        exp = attention_mask_2d[:, None, None, :].expand(batch, 1, q_len, kv_len).to(dtype)
        inv = 1.0 - exp
        pad = inv.masked_fill(inv.to(torch.bool), minv)
        out = causal.masked_fill(pad.bool(), minv)
        if q_len > 1:  # AttentionMaskConverter._unmask_unattended: fully masked rows -> 0
            out = out.mul(~torch.all(out == minv, dim=-1, keepdim=True))
        return out
    if q_len > 1 and kv_len != q_len:
        minv = torch.finfo(dtype).min
        m = torch.full((q_len, q_len), minv, dtype=dtype, device=device)
        cond = torch.arange(q_len, device=device)
        m.masked_fill_(cond < (cond + 1).view(q_len, 1), 0)
        m = torch.cat([torch.zeros(q_len, past_len, dtype=dtype, device=device), m], dim=-1)
        return m[None, None].expand(batch, 1, q_len, kv_len)
    return None


class OracleCache:
    """CU:63-320 DynamicCachePlus -- list-of-tensors KV with per-layer `true_cache_length`."""

    def __init__(self):
        self.key_cache: List[torch.Tensor] = []
        self.value_cache: List[torch.Tensor] = []
        self.true_cache_length: List[torch.Tensor] = []

    def __len__(self):
        return len(self.key_cache)

    def get_seq_length(self, layer_idx=0):  # CU:272-276 (padded length)
        if len(self.key_cache) <= layer_idx:
            return 0
        return self.key_cache[layer_idx].shape[-2]

    def get_cache(self, k, v, layer_idx):  # CU:256-268
        if len(self.key_cache) <= layer_idx:
            return k, v
        return torch.cat([self.key_cache[layer_idx], k], dim=-2), torch.cat([self.value_cache[layer_idx], v], dim=-2)

    def update(self, k, v, layer_idx, cache_decision=None):  # CU:109-253
        B, _, N, _ = k.shape
        if len(self.key_cache) <= layer_idx:
            self.key_cache.append(k)
            self.value_cache.append(v)
            if cache_decision is not None:
                self.true_cache_length.append(cache_decision.sum(dim=-1))
            else:
                self.true_cache_length.append(torch.tensor([N]).repeat(B))
        elif cache_decision is not None:
            if B == 1 and N == 1:
                if bool(cache_decision[0, 0]):
                    self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
                    self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
                    self.true_cache_length[layer_idx] += N
            else:
                ks, vs = [], []
                for b in range(B):
                    keep = cache_decision[b]
                    tl = int(self.true_cache_length[layer_idx][b])
                    ks.append(torch.cat([self.key_cache[layer_idx][b, :, :tl, :], k[b, :, keep, :]], dim=-2))
                    vs.append(torch.cat([self.value_cache[layer_idx][b, :, :tl, :], v[b, :, keep, :]], dim=-2))
                    self.true_cache_length[layer_idx][b] += int(keep.sum().item())
                mx = max(x.shape[-2] for x in ks)
                for b in range(B):
                    cur = ks[b].shape[-2]
                    z = torch.zeros((ks[b].shape[0], mx - cur, ks[b].shape[-1]), dtype=ks[b].dtype, device=ks[b].device)
                    ks[b] = torch.cat([ks[b], z], dim=-2)
                    vs[b] = torch.cat([vs[b], z], dim=-2)
                self.key_cache[layer_idx] = torch.stack(ks)
                self.value_cache[layer_idx] = torch.stack(vs)
        else:
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
            self.true_cache_length[layer_idx] += N
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def to_legacy_cache(self):  # CU:295-302
        return tuple((self.key_cache[i], self.value_cache[i]) for i in range(len(self))), self.true_cache_length

    @classmethod
    def from_legacy_cache(cls, pkv):  # CU:304-318
        c = cls()
        if pkv is not None:
            for i in range(len(pkv[0])):
                k, v = pkv[0][i]
                c.update(k, v, i)
            c.true_cache_length = pkv[1]
        return c


# --------------------------------------------------------------------------------------------
# predictors
# --------------------------------------------------------------------------------------------
def _lin(x, sd, p, bias=True):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def vision_predictor(sd, prefix, x, image_policy, nhead, num_layers):
    """DML:1348-1359 + CTL:153-180,320-323.  x [B,n,H] -> logits [B,n,2]."""
    h = F.gelu(_lin(_ln(x, sd, prefix + "down_mlp.0"), sd, prefix + "down_mlp.1"))
    h = h * image_policy
    for j in range(num_layers):
        q = prefix + f"transformer.{j}."
        B, N, C = h.shape
        y = _ln(h, sd, q + "norm1")
        qkv = F.linear(y, sd[q + "attn.qkv.weight"]).reshape(B, N, 3, nhead, C // nhead).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], dropout_p=0.0)
        a = a.transpose(1, 2).reshape(B, N, C)
        h = h + _lin(a, sd, q + "attn.proj")
        y = _ln(h, sd, q + "norm2")
        h = h + _lin(F.gelu(_lin(y, sd, q + "mlp.fc1")), sd, q + "mlp.fc2")
    B, N, C = h.shape
    local_x = h[:, :, : C // 2]
    global_x = (h[:, :, C // 2 :] * image_policy).sum(dim=1, keepdim=True) / torch.sum(image_policy, dim=1, keepdim=True)
    z = torch.cat([local_x, global_x.expand(B, N, C // 2)], dim=-1)
    z = F.gelu(_lin(z, sd, prefix + "output_mlp.0"))
    z = F.gelu(_lin(z, sd, prefix + "output_mlp.2"))
    return _lin(z, sd, prefix + "output_mlp.4")


def text_predictor(sd, prefix, x):
    """DML:1374-1387.  x [B,T,H] -> logits [B,T,2]."""
    p = prefix + "output_mlp."
    z = _ln(x, sd, p + "0")
    z = F.gelu(_lin(z, sd, p + "1"))
    z = F.gelu(_lin(z, sd, p + "3"))
    z = F.gelu(_lin(z, sd, p + "5"))
    return _lin(z, sd, p + "7")


def topk_keep_index(score, k, tie_break="stable"):
    """DML:1902-1908.  score [B,n] (model dtype) -> ascending kept indices [B,k] int64."""
    if tie_break == "torch":
        order = torch.argsort(score, dim=1, descending=True)
    else:
        order = torch.argsort(score, dim=1, descending=True, stable=True)
    keep, _ = torch.sort(order[:, :k], dim=1, descending=False)
    return keep


# --------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------
class Oracle:
    def __init__(self, cfg, sd, dtype=torch.float32, device="cpu", clip=None, tie_break="stable"):
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        self.sd = {k: v.to(device=self.device, dtype=dtype) for k, v in sd.items()}
        self.clip = clip.to(device=self.device, dtype=dtype) if clip is not None else None
        self.tie_break = tie_break
        self.nH = cfg.num_attention_heads
        self.nKV = cfg.num_key_value_heads
        self.d = cfg.hidden_size // self.nH
        self.records = {}
        self.answer_indice = None  # DML:1644 -- state of the no-KV-cache decode mode, never reset by the reference
        self._table_len = 0
        self._ensure_table(cfg.max_position_embeddings)

    def _ensure_table(self, n):
        if n > self._table_len:
            self.cos, self.sin = rope_table(self.d, n, self.cfg.rope_theta, self.dtype, self.device)
            self._table_len = n

    # ---- multimodal glue --------------------------------------------------------------------
    def encode_images(self, images):
        """ARCH:163-166 + multimodal_encoder/clip_encoder.py:43-71 (hidden_states[-2], drop CLS)
        + multimodal_projector/builder.py:172-179 (mlp2x_gelu)."""
        with torch.no_grad():
            out = self.clip(images.to(device=self.device, dtype=self.dtype), output_hidden_states=True)
        f = out.hidden_states[self.cfg.mm_vision_select_layer][:, 1:].to(images.dtype if images.is_floating_point() else self.dtype)
        f = f.to(self.dtype)
        return _lin(F.gelu(_lin(f, self.sd, "model.mm_projector.0")), self.sd, "model.mm_projector.2")

    def embed(self, ids):
        return F.embedding(ids, self.sd["model.embed_tokens.weight"])

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images, image_features=None):
        """ARCH:169-601, single-image-per-row, non-anyres path (what LLaVA-1.5 eval exercises)."""
        if images is None and image_features is None or input_ids.shape[1] == 1:
            return (input_ids, position_ids, attention_mask, past_key_values, None, labels), (None,)
        if image_features is None:
            image_features = self.encode_images(images)
        _labels, _position_ids, _attention_mask = labels, position_ids, attention_mask
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        else:
            attention_mask = attention_mask.bool()
        if position_ids is None:
            position_ids = torch.arange(0, input_ids.shape[1], dtype=torch.long, device=input_ids.device)
        if labels is None:
            labels = torch.full_like(input_ids, IGNORE_INDEX)
        ids_l = [a[m] for a, m in zip(input_ids, attention_mask)]
        lab_l = [a[m] for a, m in zip(labels, attention_mask)]
        new_embeds, new_labels, indices = [], [], []
        img_idx = 0
        for b in range(len(ids_l)):
            cur, cl = ids_l[b], lab_l[b]
            n_img = int((cur == IMAGE_TOKEN_INDEX).sum())
            if n_img == 0:  # ARCH:315-324
                new_embeds.append(torch.cat([self.embed(cur), image_features[img_idx][0:0]], dim=0))
                new_labels.append(cl)
                img_idx += 1
                continue
            img_pos = int(torch.where(cur == IMAGE_TOKEN_INDEX)[0].item())  # ARCH:330-332 (exactly one image)
            ins0 = img_pos + 1
            ans0 = int(torch.where(cl == -100)[0][-1].item()) + 1  # ARCH:334
            sys_ids, ins_ids, ans_ids = cur[:img_pos], cur[ins0:ans0], cur[ans0:]
            txt = self.embed(torch.cat([sys_ids, ins_ids, ans_ids]))  # ARCH:364-366
            sys_e = txt[:img_pos]
            img_e = image_features[img_idx].to(self.device)
            img_idx += 1
            ins_e = txt[ins0 - 1 : ans0 - 1]
            ans_e = txt[ans0 - 1 : cur.shape[0] - 1]
            e = torch.cat([sys_e, img_e, ins_e, ans_e])
            new_embeds.append(e)
            new_labels.append(
                torch.cat([cl[:img_pos], torch.full((img_e.shape[0],), IGNORE_INDEX, dtype=cl.dtype, device=cl.device), cl[ins0:ans0], cl[ans0:]])
            )
            # ARCH:418-454 -- last "USER:" occurrence inside the instruct span
            ins_list = ins_ids.tolist()
            starts = [i for i in range(len(ins_list) - len(USER_IDS) + 1) if ins_list[i : i + len(USER_IDS)] == USER_IDS]
            last_ins = starts[-1] if starts else 0
            s = sys_e.shape[0]
            i0 = s + img_e.shape[0]
            a0 = i0 + ins_e.shape[0]
            indices.append(
                {"system": [0, s], "image": [s, i0], "instruct": [i0, a0], "answer": [a0, e.shape[0]], "last_instruct": [i0 + last_ins, a0]}
            )
        tmax = getattr(self.cfg, "tokenizer_model_max_length", None)  # ARCH:493-506
        if tmax is not None:
            new_embeds = [x[:tmax] for x in new_embeds]
            new_labels = [x[:tmax] for x in new_labels]
            for x in indices:
                for key, value in x.items():
                    value[0] = min(value[0], tmax)
                    value[1] = min(value[1], tmax)
        left = getattr(self.cfg, "tokenizer_padding_side", "right") == "left"  # ARCH:529-555
        max_len = max(x.shape[0] for x in new_embeds)
        B = len(new_embeds)
        padded = []
        dev = new_labels[0].device
        lab_p = torch.full((B, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype, device=dev)
        am = torch.zeros((B, max_len), dtype=attention_mask.dtype, device=dev)
        pid = torch.zeros((B, max_len), dtype=position_ids.dtype, device=dev)
        for i, (e, l) in enumerate(zip(new_embeds, new_labels)):  # right padding (ARCH:558-577) / left padding (ARCH:529-555)
            n = e.shape[0]
            z = torch.zeros((max_len - n, e.shape[1]), dtype=e.dtype, device=e.device)
            if left:
                padded.append(torch.cat((z, e), dim=0))
                if i < len(indices):  # as the reference: input_embeds_indices[i] (one entry per row that held an image)
                    for key, value in indices[i].items():
                        value[0] += max_len - n
                        value[1] += max_len - n
                if n > 0:
                    lab_p[i, -n:] = l
                    am[i, -n:] = True
                    pid[i, -n:] = torch.arange(0, n, dtype=pid.dtype, device=dev)
                continue
            padded.append(torch.cat((e, z), dim=0))
            if n > 0:
                lab_p[i, :n] = l
                am[i, :n] = True
                pid[i, :n] = torch.arange(0, n, dtype=pid.dtype, device=dev)
        embeds = torch.stack(padded, dim=0)
        return (
            None,
            None if _position_ids is None else pid,
            None if _attention_mask is None else am.to(dtype=_attention_mask.dtype),
            past_key_values,
            embeds,
            None if _labels is None else lab_p,
        ), (indices,)

    # ---- decoder --------------------------------------------------------------------------
    def _attn(self, i, x, attention_mask, position_ids, cache, init_n, sparse_layer, text_decision):
        """DML:1009-1129."""
        sd, p = self.sd, f"model.layers.{i}.self_attn."
        B, T, _ = x.shape
        q = F.linear(x, sd[p + "q_proj.weight"]).view(B, T, self.nH, self.d).transpose(1, 2)
        k = F.linear(x, sd[p + "k_proj.weight"]).view(B, T, self.nKV, self.d).transpose(1, 2)
        v = F.linear(x, sd[p + "v_proj.weight"]).view(B, T, self.nKV, self.d).transpose(1, 2)
        kv_seq_len = T
        pos_len = init_n
        if cache is not None:
            kv_seq_len += cache.get_seq_length(i)
            if i >= sparse_layer:
                pos_len = cache.get_seq_length(sparse_layer - 1)  # DML:1031-1037
            else:
                pos_len += cache.get_seq_length(i)  # DML:1039-1041
        self._ensure_table(pos_len)
        q, k = apply_rope(q, k, self.cos[:pos_len], self.sin[:pos_len], position_ids)
        if cache is not None:
            if text_decision is not None:  # DML:1061-1076: attend over cache+new, then store conditionally
                tk, tv = k, v
                k, v = cache.get_cache(k, v, i)
                cache.update(tk, tv, i, text_decision)
            else:
                k, v = cache.update(k, v, i)
        if self.nKV != self.nH:
            rep = self.nH // self.nKV
            k = k[:, :, None].expand(B, self.nKV, rep, k.shape[-2], self.d).reshape(B, self.nH, k.shape[-2], self.d)
            v = v[:, :, None].expand(B, self.nKV, rep, v.shape[-2], self.d).reshape(B, self.nH, v.shape[-2], self.d)
        if attention_mask is not None and attention_mask.size() != (B, 1, T, kv_seq_len):
            raise ValueError(f"Attention mask should be of size {(B, 1, T, kv_seq_len)}, but is {attention_mask.size()}")
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=attention_mask is None and T > 1)
        o = o.transpose(1, 2).contiguous().reshape(B, T, self.nH * self.d)
        return F.linear(o, sd[p + "o_proj.weight"])

    def _layer(self, i, h, attention_mask, position_ids, cache, init_n, sparse_layer, text_decision):
        """DML:1271-1295."""
        sd, p, eps = self.sd, f"model.layers.{i}.", self.cfg.rms_norm_eps
        r = h
        x = rmsnorm(h, sd[p + "input_layernorm.weight"], eps)
        h = r + self._attn(i, x, attention_mask, position_ids, cache, init_n, sparse_layer, text_decision)
        r = h
        x = rmsnorm(h, sd[p + "post_attention_layernorm.weight"], eps)
        m = F.linear(F.silu(F.linear(x, sd[p + "mlp.gate_proj.weight"])) * F.linear(x, sd[p + "mlp.up_proj.weight"]), sd[p + "mlp.down_proj.weight"])
        return r + m

    def model_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, use_cache=True, input_embeds_indices=None):
        """DML:1656-2594, eval-mode branches: vision block (DML:1826-1994) and the cached
        output-text decision (DML:2377-2391).  Returns (normed hidden [B,T',H], legacy cache)."""
        cfg, sc = self.cfg, self.cfg.sparse_config
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You have to specify exactly one of input_ids or inputs_embeds")
        B, seq_length = (input_ids if input_ids is not None else inputs_embeds).shape[:2]
        cache = None
        past_len = 0
        if use_cache:
            cache = past_key_values if isinstance(past_key_values, OracleCache) else OracleCache.from_legacy_cache(past_key_values)
            use_legacy = not isinstance(past_key_values, OracleCache)
            past_len = cache.get_seq_length(0)
        if position_ids is None:
            position_ids = torch.arange(past_len, seq_length + past_len, dtype=torch.long, device=self.device).unsqueeze(0)
        if inputs_embeds is None:
            inputs_embeds = self.embed(input_ids)
        h = inputs_embeds
        init_n = h.shape[1]
        text_decision = None
        sparse_layer = sc["sparse_layer"]
        rec = self.records = {"text_decision": None}
        vision_on = sc["use_vision_predictor"] and input_embeds_indices is not None and B == len(input_embeds_indices)
        if vision_on:
            init_image_n = input_embeds_indices[0]["image"][1] - input_embeds_indices[0]["image"][0]
            image_prev_decision = torch.ones(B, init_image_n, 1, dtype=h.dtype, device=h.device)
        for i in range(cfg.num_hidden_layers):
            if use_cache:
                past_len = cache.get_seq_length(i)
            mask = causal_mask_for_sdpa(attention_mask, B, seq_length, past_len, h.dtype, h.device)  # DML:1805-1810
            if vision_on and i == sparse_layer:
                s0, s1 = input_embeds_indices[0]["image"]
                img = torch.stack([h[b, input_embeds_indices[b]["image"][0] : input_embeds_indices[b]["image"][1], :] for b in range(B)], dim=0)
                logit = vision_predictor(self.sd, "model.image_score_predictor.", img, image_prev_decision, sc["nhead"], sc["num_layers"]).reshape(B, -1, 2)
                score = F.log_softmax(logit, dim=-1)[:, :, 0]  # DML:1867,1898
                k = int(init_image_n * sc["vision_keep_rate"])  # DML:1899-1901
                keep = topk_keep_index(score, k, self.tie_break)
                if getattr(self, "force_keep_index", None) is not None:  # test hook: continue a comparison past a near-tied top-k boundary
                    keep = self.force_keep_index.to(keep.device)
                rec.update(vision_logit=logit, vision_score=score, keep_index=keep, predictor_input=img)
                img_h = h[:, s0:s1, :]
                kept = img_h.gather(dim=1, index=keep[..., None].expand(B, k, h.shape[2]))
                h = torch.cat([h[:, :s0, :], kept, h[:, s1:, :]], dim=1)  # DML:1944-1951
                position_ids = torch.cat(
                    [
                        torch.arange(0, s0, device=h.device).repeat(B, 1),
                        keep + s0,
                        torch.arange(s1, init_n, device=h.device).repeat(B, 1),
                    ],
                    dim=1,
                ).to(torch.long)  # DML:1963-1983
                rec["position_ids"] = position_ids
                drop = init_image_n - k
                for d_ in input_embeds_indices:  # DML:1986-1994 (in place, like the reference)
                    d_["image"][1] -= drop
                    for key in ("instruct", "last_instruct", "answer"):
                        d_[key][0] -= drop
                        d_[key][1] -= drop
            if sc["use_text_predictor"] and i == sparse_layer:
                if (not past_len) and input_embeds_indices is not None and B == len(input_embeds_indices) and sc["use_instruct_predictor"]:
                    # DML:2261-2375 -- prefill, first instruct: drop the instruct tokens the predictor rejects (last one always stays)
                    assert B == 1, "Using text predictor must keep the batch size to 1"
                    li0, li1 = input_embeds_indices[0]["last_instruct"]
                    span = h[:, li0 : li1 - 1, :]
                    il = text_predictor(self.sd, "model.instruct_score_predictor.", span).reshape(B, -1, 2)
                    keep_i = torch.where(il[0, :, 0] > il[0, :, 1])[0].unsqueeze(0)
                    kept = span.gather(dim=1, index=keep_i[..., None].expand(B, -1, h.shape[2]))
                    h = torch.cat([h[:, :li0, :], kept, h[:, li1 - 1 :, :]], dim=1)
                    position_ids = torch.cat([position_ids[:, :li0], position_ids[:, li0 : li1 - 1].gather(dim=1, index=keep_i), position_ids[:, li1 - 1 :]], dim=1)
                    rec.update(instruct_logit=il, instruct_keep=keep_i, position_ids=position_ids)
                    drop_i = li1 - 1 - li0 - keep_i.shape[1]
                    for d_ in input_embeds_indices:
                        d_["instruct"][1] -= drop_i
                        d_["last_instruct"][1] -= drop_i
                        d_["answer"][0] -= drop_i
                        d_["answer"][1] -= drop_i
                elif past_len and h.shape[1] > 1 and sc["use_instruct_predictor"]:
                    # DML:2506-2521 -- a new instruct chunk on top of a cache (multi-round dialogue): every new token attends to the
                    # cache and to the chunk (causally); only the tokens the instruct predictor keeps (the last one always) are stored
                    tl = text_predictor(self.sd, "model.instruct_score_predictor.", h).reshape(B, -1, 2)
                    text_decision = tl[:, :, 0] > tl[:, :, 1]
                    text_decision[:, -1:] = True
                    rec.update(text_logit=tl, text_decision=text_decision)
                elif (not past_len) and input_embeds_indices is not None and B == len(input_embeds_indices) and sc["use_output_text_predictor"] and not use_cache:
                    # DML:2393-2504 -- output infer stage WITHOUT KV cache: the whole sequence is re-run every step and the answer
                    # tokens [answer_indice, -1) are compacted by top-k of the RAW keep logit, k = max kept count over the batch.
                    # (On the first call answer_indice == T', so left = everything and right = h[:, -1:] duplicates the last token.)
                    if self.answer_indice is None:
                        self.answer_indice = input_embeds_indices[0]["instruct"][1]
                    ai = self.answer_indice
                    span = h[:, ai:-1, :]
                    tl = text_predictor(self.sd, "model.output_text_score_predictor.", span).reshape(B, -1, 2)
                    text_decision = tl[:, :, 0] > tl[:, :, 1]
                    num_keep = int(text_decision.sum(dim=1).max()) if text_decision.numel() else 0
                    order = torch.argsort(tl[:, :, 0], dim=1, descending=True, stable=(self.tie_break != "torch"))
                    keep_t, _ = torch.sort(order[:, :num_keep], dim=1, descending=False)
                    kept = span.gather(dim=1, index=keep_t[..., None].expand(B, num_keep, h.shape[2]))
                    h = torch.cat([h[:, :ai, :], kept, h[:, -1:, :]], dim=1)
                    if position_ids.shape[0] != B:
                        position_ids = position_ids.expand(B, -1)
                    position_ids = torch.cat([position_ids[:, :ai], position_ids[:, ai:-1].gather(dim=1, index=keep_t), position_ids[:, -1:]], dim=1)
                    rec.update(nocache_logit=tl, nocache_keep=keep_t, position_ids=position_ids)
                    text_decision = None  # only used for the compaction above; layers run without a cache
                elif past_len and h.shape[1] == 1 and sc["use_output_text_predictor"]:  # DML:2377-2391
                    tl = text_predictor(self.sd, "model.output_text_score_predictor.", h).reshape(B, -1, 2)
                    text_decision = tl[:, :, 0] > tl[:, :, 1]
                    if getattr(self, "force_text_decision", None) is not None:  # test hook: continue a comparison past a keep/evict logit pair that sits on the boundary
                        text_decision = self.force_text_decision.to(text_decision.device).bool().reshape(text_decision.shape)
                    rec.update(text_logit=tl, text_decision=text_decision)
            h = self._layer(i, h, mask, position_ids, cache, init_n, sparse_layer, text_decision)
            attention_mask = None  # DML:2554
        h = rmsnorm(h, self.sd["model.norm.weight"], cfg.rms_norm_eps)
        nxt = None
        if use_cache:
            nxt = cache.to_legacy_cache() if use_legacy else cache
        return h, nxt

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, images=None, image_features=None, use_cache=True, input_embeds_indices=None):
        """DLL:68-115 + DML:2631-2813 (no labels).  Returns (logits fp32 [B,T',V], past_key_values)."""
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, _), (input_embeds_indices,) = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, past_key_values, None, images, image_features
            )
        h, pkv = self.model_forward(input_ids, attention_mask, position_ids, past_key_values, inputs_embeds, use_cache, input_embeds_indices)
        logits = F.linear(h, self.sd["lm_head.weight"]).float()  # DML:2709-2710
        return logits, pkv

    @torch.no_grad()
    def greedy(self, input_ids, images=None, image_features=None, max_new_tokens=16, eos_token_id=2, trace=None):
        """What `generate(do_sample=False, num_beams=1, use_cache=True)` does (DLL:117-152 + HF greedy
        search, restated as the reference's own hand-rolled loop BLTM:310-337): new tokens only."""
        B = input_ids.shape[0]
        logits, pkv = self.forward(input_ids, images=images, image_features=image_features)
        if trace is not None:
            trace.append(dict(logits=logits[:, -1].clone(), **{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.records.items()}))
        out = []
        unfinished = torch.ones(B, dtype=torch.long, device=logits.device)
        for step in range(max_new_tokens):
            nxt = logits[:, -1, :].argmax(dim=-1)
            if eos_token_id is not None:
                nxt = nxt * unfinished + 0 * (1 - unfinished)  # pad_token_id = 0 after EOS
                unfinished = unfinished * (nxt != eos_token_id).long()
            out.append(nxt)
            if int(unfinished.max()) == 0 or step == max_new_tokens - 1:
                break
            logits, pkv = self.forward(nxt[:, None], past_key_values=pkv)
            if trace is not None:
                lens = [t.clone() for t in pkv[1]]
                trace.append(dict(logits=logits[:, -1].clone(), text_decision=self.records.get("text_decision"), true_cache_length=lens, kv_len_last=pkv[0][-1][0].shape[-2]))
        return torch.stack(out, dim=1), pkv


# --------------------------------------------------------------------------------------------
# N5: training-time ops (checker for dynamic_llava_amd/train_ops.py)
# --------------------------------------------------------------------------------------------
def softmax_with_policy(attn, policy, eps=1e-6):
    """DML:913-930.  attn [B,H,N,N] scores (mask already added), policy [B,N,1] keep decisions (differentiable).
    A dropped key j keeps weight only on the diagonal (i == j); `eps / N` is added to EVERY entry, masked ones included."""
    B, N, _ = policy.size()
    attn_policy = policy.reshape(B, 1, 1, N)
    eye = torch.eye(N, dtype=attn_policy.dtype, device=attn_policy.device).view(1, 1, N, N)
    attn_policy = attn_policy + (1.0 - attn_policy) * eye
    max_att = torch.max(attn, dim=-1, keepdim=True)[0]
    attn = attn - max_att
    attn = attn.to(torch.float32).exp_() * attn_policy.to(torch.float32)
    attn = (attn + eps / N) / (attn.sum(dim=-1, keepdim=True) + eps)
    return attn.type_as(max_att)


def sdpa_with_policy(query, key, value, attn_mask=None, is_causal=False, scale=None, policy=None):
    """DML:933-970 with dropout_p = 0 (`attention_dropout` is 0.0 in every shipped config).  [B,H,L,d] tensors."""
    B = query.size(0)
    L, S = query.size(-2), key.size(-2)
    scale_factor = 1 / math.sqrt(query.size(-1)) if scale is None else scale
    if attn_mask is not None:
        attn_bias = torch.zeros_like(attn_mask, dtype=query.dtype)
    else:
        attn_bias = torch.zeros(B, 1, L, S, dtype=query.dtype, device=query.device)
    if is_causal:
        assert attn_mask is None
        temp_mask = torch.ones(B, 1, L, S, dtype=torch.bool, device=query.device).tril(diagonal=0)
        attn_bias.masked_fill_(temp_mask.logical_not(), float("-inf"))
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            attn_bias.masked_fill_(attn_mask.logical_not(), float("-inf"))
        else:
            attn_bias += attn_mask
    attn_weight = query @ key.transpose(-2, -1) * scale_factor
    attn_weight += attn_bias
    if policy is not None:
        attn_weight = softmax_with_policy(attn_weight, policy=policy)
    else:
        attn_weight = torch.softmax(attn_weight, dim=-1)
    return attn_weight @ value


def gumbel_hard_keep(log_probs, gumbels, tau, prev_decision):
    """DML:1868-1876: `F.gumbel_softmax(log_probs, tau, hard=True)[:, :, 0:1] * prev_decision` with the Gumbel noise
    passed in (torch draws it as `-empty_like(logits).exponential_().log()`); straight-through estimator."""
    y = (log_probs + gumbels) / tau
    y_soft = y.softmax(-1)
    index = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(log_probs).scatter_(-1, index, 1.0)
    ret = y_hard - y_soft.detach() + y_soft
    return ret[:, :, 0:1] * prev_decision
