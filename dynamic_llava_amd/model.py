"""Host side of the MI355X-native Dynamic-LLaVA hot path: same API surface as the reference's
`DynamicLlavaLlamaForCausalLM` (llava/model/language_model/dynamic_llava_llama.py:50-169), different engine.

What stays on PyTorch-ROCm (per BASELINE.json north_star): CLIP ViT-L/14-336 + mm_projector, the embedding
lookup and the dense decoder GEMMs (fused QKV, O, fused gate|up, down, lm_head -> hipBLASLt/MFMA).
Everything else on the path is a hand-written HIP kernel behind the C ABI (include/dynllava.h):
RMSNorm(+residual), RoPE+KV append, varlen prefill attention, ragged split-KV decode attention, SiLU*up,
vision predictor, top-k select, token compaction, text predictor + eviction decision, greedy/advance.

Round 6: this file is the API shell (construction, finalize(), forward(), generate() and their glue); the parameter tree and the vision side live in
modules.py, the prompt layout / prefill planner / layer loop in prefill.py (PrefillEngine), the decode-step builders and their schedule in decode.py
(DecodeScheduler) -- mixins of the one class the reference's harness sees.

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
import operator
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops as ops
from .cache import KVSlabCache
from .config import IGNORE_INDEX, IMAGE_TOKEN_INDEX, DynamicLlavaConfig

from .decode import DecodeScheduler, _DecodeState  # noqa: F401
from .modules import (CLIPVisionTower, CausalLMOutputWithPast, DynamicLlamaDecoderLayer, DynamicLlavaLlamaModel, TextPredictor,  # noqa: F401 (re-exported:
                      VisionPredictor)  # tests, tools and the package's lazy attributes import these names from here)
from .prefill import USER_IDS, PrefillEngine  # noqa: F401

_DATA_PTR, _VERSION = operator.methodcaller("data_ptr"), operator.attrgetter("_version")


class DynamicLlavaLlamaForCausalLM(PrefillEngine, DecodeScheduler, nn.Module):
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
        # round 6: o_proj at <= 256 rows (post-compaction prefill layers, decode batches from `tiles_o_proj_min_decode_batch` rows) on dl_linear_tiles' partial sums
        # OFF by default: 10 % on the launch pair at M = 170, not visible in the prefill (8.26-8.42 ms with and without, box to box) and 1.07 GB of copies at 7B
        self.tiles_o_proj = os.environ.get("DL_TILES_O", "0") == "1"
        # (tools/bench_linear_tiles.py --oproj, with the consumer launch: M = 170 23.6 vs 26.4 us for dl_linear_splitk, M = 192 24.7 vs 26.4, M = 117 22.9 vs 24.0 but the library's 21.4;
        # 32 rows 17.3 vs 18.1 for dl_gemm_smallm, 16 rows 16.8 vs 16.1: used for 129..256 prefill rows; decode batches stay where they were -- 33 = off)
        self.tiles_o_proj_min_decode_batch = int(os.environ.get("DL_TILES_O_MIN_B", "33"))
        self.packed_down_proj = os.environ.get("DL_PACKED_DOWN", "1") == "1"  # down_proj too (partial sums for dl_add_rmsnorm_parts): A/B knob
        # batched decode (4..32 rows), tools/bench_decode_batch.py: the MLP on dl_linear_packed from 4 rows on (B = 16: 4.05 -> 3.79 ms per step, 24: 4.63 -> 4.03), q|k|v too from 16 rows
        # on (24: 4.05 -> 3.97, 32: 4.21 -> 4.11); o_proj stays on dl_gemm_smallm's partial sums up to 32 rows (against the library GEMM + add: 32 rows 4.20 -> 4.11)
        self.packed_decode_qkv_min_batch = int(os.environ.get("DL_PACKED_DECODE_QKV_MIN_B", "16"))
        self.packed_decode_qkv_parts = os.environ.get("DL_PACKED_DECODE_QKV_PARTS", "1") != "0"  # ... as fp32 partial sums of its two k ranges, added by dl_attn_decode_rope_parts
        self.packed_decode_qkv_parts_max_batch = int(os.environ.get("DL_PACKED_DECODE_QKV_PARTS_MAX_B", "24"))  # (tools/bench_decode_qkv_parts.py: -2.2 % per step at 16 rows, -1.8 % at 24; at 32 rows -0.6 % on equal prompts but +0.5 % on configs[2]'s ragged batch)
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
        if self._proj_tiles is None or f.dtype != pj[0].weight.dtype or not f.is_cuda or B > (vt.tiles_max_batch if vt is not None else 1) or pj[0].bias is None or pj[2].bias is None:
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
            l.pack(operand_copies=self.packed_prefill_gemm, o_copy=self.tiles_o_proj)  # (weights replaced later: call finalize() again -- the operand-order copies are made here)
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

    # ---- decoder engine -----------------------------------------------------------------------
    def _check_ready(self, weights=True):
        if not self._packed:
            raise ops.HipOpsError("call model.finalize() after loading weights (done by the builders)")
        if not weights:
            return
        # the operand-order copies are detached from the parameters.  A load_state_dict() / an in-place edit / a replaced `.data` after
        # finalize() would otherwise leave the packed prefill and the packed decode batches on the OLD weights while the GEMV / library paths use the new ones
        # -- path-dependent results with no error.  (data_ptr, _version) of every decoder projection weight is compared and the model
        # re-finalized when one moved.  Checked by generate() and by every forward() that starts a sequence (0.1 ms: 224 parameters at 7B).
        if self._weights_fingerprint() != self._fp:
            self._packed = False
            self.finalize()

    @staticmethod
    def _tiles_o_config(rows, n_out):
        """(tile_shape, k_split) of o_proj on dl_linear_tiles' partial-sum form: up to 8 row tiles as ONE row block, more as two; 8 units (128 neurons) per workgroup;
        as many k ranges (<= 8) as keep one round of workgroups on the 256 CUs."""
        rt = (rows + 15) // 16
        n_mb = 1 if rt <= 8 else 2
        tm = -(-rt // n_mb)
        tm = {5: 6}.get(tm, tm)  # (5 row tiles are built with every epilogue under another shape code; 6 wastes one tile and keeps the table small)
        n_nb = -(-(n_out // 16) // 8)
        ks = max(1, min(8, 256 // (n_mb * n_nb)))
        return 100 * tm + 42 + (20000 if tm <= 2 else 0), ks

    def _weights_fingerprint(self):
        ps = self._fp_params
        return tuple(map(_DATA_PTR, ps)), tuple(map(_VERSION, ps))  # (C-level maps: ~45 us for the 224 projection weights of a 7B model)

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
        dec = sum(nb(l.wp_qkv) + nb(l.wp_gu) + nb(l.wp_down) + nb(l.wp_o) for l in self.model.layers)
        vt = self.get_vision_tower()
        clip_tiles = int(getattr(vt, "tiles_bytes", 0) or 0) if vt is not None else 0
        clip_fused = sum(nb(w) + nb(b) for w, b in getattr(vt, "_qkv", [])) if vt is not None else 0
        proj = sum(nb(t) for t in (getattr(self, "_proj_tiles", None) or ()))
        return {"decoder_operand_order": dec, "clip_operand_order": clip_tiles, "clip_fused_qkv": clip_fused, "projector_operand_order": proj,
                "total": dec + clip_tiles + clip_fused + proj}

    # DL_USE_HIP_GRAPH=0: debugging switch that wins over any assignment -- e.g. the GPU tests under PYTORCH_NO_CUDA_MEMORY_CACHING=1 (every tensor its own
    # allocation, so that an out-of-bounds read faults instead of landing in a neighbour; stream capture is impossible without the caching allocator)
    _force_eager = os.environ.get("DL_USE_HIP_GRAPH", "1") == "0"

    @property
    def use_hip_graph(self):
        return self._use_hip_graph and not self._force_eager

    @use_hip_graph.setter
    def use_hip_graph(self, v):
        self._use_hip_graph = bool(v)

    # ---- one decode step; every buffer persistent, no host sync -> hipGraph-capturable ----
    def knobs(self) -> dict:
        """The resolved values of every tuning knob and test hook that decides WHICH kernels a request runs on (constructor defaults, DL_* environment
        values, attributes set later) -- recorded in the bench line so that a run can be reproduced.  The last three are test hooks:
        they must be None / 0 outside tests/."""
        return {
            "use_hip_graph": self.use_hip_graph, "attn_inkernel_combine": self.attn_inkernel_combine, "device_prompt_layout": self.device_prompt_layout,
            "tp_side_stream": self.tp_side_stream, "gemv_max_decode_batch": self.gemv_max_decode_batch, "smallm_max_decode_batch": self.smallm_max_decode_batch,
            "fuse_qkv_attn": self.fuse_qkv_attn, "fuse_gu_tp": self.fuse_gu_tp, "fused_attn_max_splits": self.fused_attn_max_splits, "gu_grid_cap": self.gu_grid_cap,
            "qkv_attn_grid_cap": self.qkv_attn_grid_cap, "splitk_o_proj": self.splitk_o_proj, "packed_prefill_gemm": self.packed_prefill_gemm, "packed_down_proj": self.packed_down_proj, "packed_qkv_parts": self.packed_qkv_parts, "packed_decode_mlp": self.packed_decode_mlp, "packed_decode_mlp_min_batch": self.packed_decode_mlp_min_batch, "packed_decode_qkv_min_batch": self.packed_decode_qkv_min_batch, "packed_decode_qkv_parts": self.packed_decode_qkv_parts, "packed_decode_qkv_parts_max_batch": self.packed_decode_qkv_parts_max_batch,
            "smallm_wide_slices": self.smallm_wide_slices, "decode_sync_every": self.decode_sync_every, "prefill_width_bucket": self.prefill_width_bucket,
            "max_prefill_graphs": self.max_prefill_graphs,
            "tiles_o_proj": self.tiles_o_proj, "tiles_o_proj_min_decode_batch": self.tiles_o_proj_min_decode_batch,
            "clip_tiles_gemm": getattr(self.get_vision_tower(), "tiles_gemm", None), "clip_tiles_max_batch": getattr(self.get_vision_tower(), "tiles_max_batch", None),
            "clip_tiles_ksplit": [getattr(self.get_vision_tower(), "tiles_ksplit_out", None), getattr(self.get_vision_tower(), "tiles_ksplit_fc2", None)],
            "test_hook_force_text_decision": self.force_text_decision is not None, "test_hook_single_split_keys_override": self.single_split_keys_override,
            "test_hook_min_keys_per_split": getattr(self, "min_keys_per_split", None),
        }

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
        self._check_ready(weights=past_key_values is None)  # (the ~0.1 ms weight-fingerprint check runs where a sequence starts, not on every decode step of a forward() loop)
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
        if B == 1 or self._lp_ws is not None:
            # (batched decode and the packed prefill also contain in-kernel hand-overs -- dl_linear_packed with k ranges, reporting through
            # _lp_err -- so a reducer timeout must be seen for B > 1 too; generate() is about to synchronise for its result anyway)
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
