"""Prefill side of the engine: prompt layout (ARCH:169-601 on the host and on the device), the prefill planner, the packed-varlen layer loop with the sparsification at layer `sparse_layer` (DML:1656-2594), the prefill-shape graph cache, chunk-on-cache and no-KV-cache forwards.  A mixin of DynamicLlavaLlamaForCausalLM (model.py); split out in round 6 (no behaviour change)."""
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
from .modules import CausalLMOutputWithPast

USER_IDS = [11889, 29901]  # "USER:" -- llava/model/dynamic_llava_arch.py:36


class PrefillEngine:
    """Prompt layout, prefill planner and runner (methods of DynamicLlavaLlamaForCausalLM)."""

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

    def _prefill_knob_key(self):
        """The knobs that decide which launches a captured prefill contains."""
        vt = self.get_vision_tower()
        return (self.packed_prefill_gemm, self.packed_down_proj, self.packed_qkv_parts, self.splitk_o_proj, self.tiles_o_proj, self.prefill_width_bucket, self.device_prompt_layout,
                None if vt is None else (vt.tiles_gemm, vt.tiles_max_batch, vt.tiles_ksplit_out, vt.tiles_ksplit_fc2))

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
            if self.tiles_o_proj and getattr(layer, "wp_o", None) is not None and dt in (torch.bfloat16, torch.float16) and 128 < attn.shape[0] <= 256 and attn.shape[1] % 64 == 0:
                # round 6: o_proj [H, H] on dl_linear_tiles -- two blocks of row tiles x 128 neurons x 4 k ranges, one round of workgroups, fp32 partial sums added in range
                # order by the residual-add / RMSNorm launch: 23.6 us with the consumer at M = 170 against 26.4 for dl_linear_splitk (half as many slice bytes); up to 128
                # rows it is no better than what was there (22.9 vs 24.0, library 21.4 at M = 117) and stays off
                shp, ks_ = self._tiles_o_config(attn.shape[0], h.shape[1])
                parts_ = ops.linear_tiles(attn, layer.wp_o, h.shape[1], out=self._splitk_ws(h.shape[1])[: ks_ * total * h.shape[1]], epilogue=ops.LT_PARTS, tile_shape=shp, k_split=ks_)
                x = ops.add_rmsnorm_parts(h, parts_, layer.post_attention_layernorm.weight, eps, packed=use_lp)
            elif self.splitk_o_proj and dt in (torch.bfloat16, torch.float16) and attn.shape[0] <= 192 and attn.shape[1] >= 1024 and attn.shape[1] % 64 == 0 and h.shape[1] % 64 == 0:
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
        if ws is None or ws.numel() < 8 * 256 * H:
            ws = self._splitk_buf = torch.empty(8 * 256 * H, dtype=torch.float32, device=self.device)
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
