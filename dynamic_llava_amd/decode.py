"""Decode side of the engine: the persistent step state (stable pointers for the hipGraph), the three step builders (GEMV / small-M / library), the pooled KV slab, the schedule that picks a captured step from the observed lengths, device-error reporting.  A mixin of DynamicLlavaLlamaForCausalLM (model.py); split out in round 6 (no behaviour change)."""
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
            self.qkv_parts = torch.empty(2 * B * (nH + 2 * nKV) * d, dtype=torch.float32, device=device)  # q|k|v's two k ranges, added by the attention launch
        self.o_parts = torch.empty(8 * B * H, dtype=torch.float32, device=device) if (self.use_smallm and B <= 32 and getattr(l0, "wp_o", None) is not None) else None  # o_proj's k-range slices (dl_linear_tiles)
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


class DecodeScheduler:
    """Decode-step builders and their scheduling (methods of DynamicLlavaLlamaForCausalLM)."""

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
        qkv_parts = False
        if lp_qkv:
            nu_q, ks_q = self._lp_config(st.qkv.shape[1] // 16, False)
            qkv_parts = self.packed_decode_qkv_parts and st.B <= self.packed_decode_qkv_parts_max_batch and ks_q == 2 and self.dtype in (torch.bfloat16, torch.float16)
        ops.rmsnorm(st.h, self.model.layers[0].input_layernorm.weight, eps, out=st.x_pk if lp_qkv else st.x, packed=lp_qkv)
        for i, layer in enumerate(self.model.layers):
            if i == SL and use_tp:  # F6: decision on the hidden state entering layer SL (DML:2377-2391)
                self.model.output_text_score_predictor.decide(st.h, st.tp_ws, st.tp_logits, st.decision)
            lens = cache.len_of_layer(i)
            ns = cache.n_splits(i, st.B * nH)
            tag = (i & 0xff) if self.attn_inkernel_combine and L >= 2 else -1
            if lp_qkv and qkv_parts and cache.keys_in_flight(ns, st.B * nH) == 64:
                # round 6 (verdict r5 item 2c): the projection's two k ranges stay fp32 partial sums and the RoPE / append / attention launch adds them (each (row, head)
                # workgroup the 3 x head_dim values it reads): no hand-over inside the projection's launch.  In the step (profiles/r06_decode_qkv_parts.txt): the
                # projection 22.7 -> 20.5 us at 32 rows; the attention launch pays for the wider loads in front of its first score (24.3 -> 28.2 us in the first build,
                # most of it removed by taking q's RoPE partner half from the neighbouring lane instead of loading it): -2.2 % per step at 16 rows, -1.8 % at 24; at 32 rows
                # -0.6 % on equal prompts, +0.5 % on configs[2]'s ragged batch => up to packed_decode_qkv_parts_max_batch (24) rows
                parts_q = ops.linear_packed(st.x_pk, layer.wp_qkv, st.qkv.shape[1], out=st.qkv_parts, epilogue=ops.LP_PARTS, units_per_workgroup=nu_q, k_split=ks_q, x_packed_mk=(st.B, st.h.shape[1]))
                ops.attn_decode_rope_parts(parts_q, cos, sin, cache.len_full, lens, cache.k[i], cache.v[i], st.attn, st.attn_ws, ns, nH, nKV, d, chunk_keys=cache.spec_chunk(ns), call_tag=tag)
            else:
                if lp_qkv:
                    qkv = ops.linear_packed(st.x_pk, layer.wp_qkv, st.qkv.shape[1], out=st.qkv, units_per_workgroup=nu_q, k_split=ks_q, workspace=self._lp_ws if ks_q > 1 else None, err=self._lp_err,
                                            x_packed_mk=(st.B, st.h.shape[1]))
                else:
                    qkv = ops.gemm_smallm(st.x, layer.w_qkv, out=st.qkv, workspace=ws, n_slices=self.smallm_wide_slices) if sm else F.linear(st.x, layer.w_qkv)
                ops.attn_decode_rope(qkv, cos, sin, cache.len_full, lens, cache.k[i], cache.v[i], st.attn, st.attn_ws, ns, nH, nKV, d, keys_in_flight=cache.keys_in_flight(ns, st.B * nH), chunk_keys=cache.spec_chunk(ns),
                                     call_tag=tag)
            nw = self.model.norm.weight if i + 1 == L else self.model.layers[i + 1].input_layernorm.weight
            lp = st.use_lp_mlp
            x_mlp = st.x_pk if lp else st.x  # the packed MLP reads its input in fragment order: the norm launch writes it that way
            if sm and self.tiles_o_proj and st.o_parts is not None and st.B >= self.tiles_o_proj_min_decode_batch and getattr(layer, "wp_o", None) is not None:
                # round 6: o_proj on dl_linear_tiles -- 1-2 row tiles x 128 neurons x 8 k ranges (256 workgroups), each consumer wave streaming its own operand-order
                # weight fragments five steps ahead; the slices go to the same residual-add / RMSNorm launch (dl_gemm_smallm's partial sums: 13.0 us at 32 rows)
                shp, ks_ = self._tiles_o_config(st.B, st.h.shape[1])
                parts = ops.linear_tiles(st.attn, layer.wp_o, st.h.shape[1], out=st.o_parts[: ks_ * st.B * st.h.shape[1]], epilogue=ops.LT_PARTS, tile_shape=shp, k_split=ks_)
                ops.add_rmsnorm_parts(st.h, parts, layer.post_attention_layernorm.weight, eps, out=x_mlp, packed=lp)
            elif sm:  # (o_proj on dl_linear_packed's partial sums instead: a tie at 8..32 rows, measured and dropped)
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
               self.fused_attn_max_splits, self.qkv_attn_grid_cap, self.gu_grid_cap, self.packed_decode_qkv_min_batch, self.packed_decode_qkv_parts, self.packed_decode_qkv_parts_max_batch)
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

    def _first_token(self, st, x_last, min_new):
        if st.use_gemv and x_last.dim() == 2 and x_last.is_contiguous():
            # up to three rows: the weight-streaming GEMV the decode steps use for the same matrix (41 vs 61 us for the library's skinny GEMM at B=1)
            ops.gemv(self.lm_head.weight, st.logits, x=x_last)
        else:
            torch.matmul(x_last, self.lm_head.weight.t(), out=st.logits)
        self._prefill_logits_buf.copy_(st.logits)
        # first token: argmax only (the prompt's KV lengths are already in place); EOS is banned while step < min_new (HF semantics)
        ops.decode_advance(st.logits, st.cur_ids, st.out_ids, st.step, st.finished, self._eos, self._pad, None, None, None, min_new_tokens=min_new)
