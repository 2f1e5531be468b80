"""KV slab with in-place eviction -- the MI355X-native counterpart of the reference's DynamicCachePlus
(llava/model/language_model/cache_utils.py:63-320).

Reference behaviour being replaced: per layer per decoded token a whole-cache `torch.cat` re-allocation
(cache_utils.py:155-160, 243-248), a host sync on `cache_decision[0, 0]` (cache_utils.py:153-154) and, for B > 1, a
Python loop that slices / zero-pads / stacks rows (cache_utils.py:165-241).

Here: one pre-allocated slab per layer, K and V each [B, n_kv_heads, T_cap, head_dim] (sized for the whole
generation up front: 288 GB of HBM makes this the cheap option), and two device-resident int32 length vectors --
only two distinct length vectors exist in the reference's L x [B] `true_cache_length`: layers < sparse_layer (never
evicted) and layers >= sparse_layer (image tokens dropped at prefill, generated tokens kept only if the
output-text predictor says so).  A decode token is always written at slot len[b]; keeping it = `len[b] += 1`,
evicting it = leaving len[b] alone (the slot is overwritten by the next token).  No copies, no host syncs, and
the whole decode step is hipGraph-capturable.

Legacy view (what the harness indexes, dynamic_llava_long_text_mem.py:337-338):
    pkv[0][layer][0].shape[-2]  -> (padded) KV length of that layer  == max_b true length
    pkv[1]                      -> list of L CPU int64 tensors [B]   == true_cache_length
Both are materialised lazily (they cost a device->host copy, exactly what the caller asked for).
"""
from __future__ import annotations

import os

from typing import List

import torch


class _LayerViews:
    def __init__(self, cache: "KVSlabCache"):
        self._c = cache

    def __len__(self):
        return self._c.n_layers

    def __getitem__(self, i):
        c = self._c
        if i < 0:
            i += c.n_layers
        if not 0 <= i < c.n_layers:
            raise KeyError(f"Cache only has {c.n_layers} layers, attempted to access layer with index {i}")
        T = c.padded_length(i)
        return (c.k[i][:, :, :T, :], c.v[i][:, :, :T, :])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


# rows of at most this many keys run as ONE workgroup per (row, head): four prefetched trips of 64 keys with the speculative first request beat
# four splits + the in-kernel merge (two fabric round trips) -- 9.6 vs 9.9-10.5 us per launch, decode 2.648 -> 2.633 ms/token (A/B on one box)
_SINGLE_SPLIT_MAX_KEYS = int(os.environ.get("DL_SINGLE_SPLIT_MAX_KEYS", "256"))


class KVSlabCache:
    def __init__(self, n_layers, sparse_layer, batch, n_kv_heads, head_dim, t_cap, dtype, device):
        self.n_layers = n_layers
        self.sparse_layer = sparse_layer
        self.batch = batch
        self.n_kv_heads = n_kv_heads
        self.head_dim = head_dim
        self.t_cap = int(t_cap)
        self.dtype = dtype
        self.device = device
        # one allocation for all layers: [L, 2, B, nKV, T_cap, d]
        self.slab = torch.empty((n_layers, 2, batch, n_kv_heads, self.t_cap, head_dim), dtype=dtype, device=device)
        self.k = [self.slab[i, 0] for i in range(n_layers)]
        self.v = [self.slab[i, 1] for i in range(n_layers)]
        # lens[0] = layers < sparse_layer, lens[1] = layers >= sparse_layer
        self.lens = torch.zeros((2, batch), dtype=torch.int32, device=device)
        # exact host mirror of lens[0] (advances by one per token for every row) -> capacity checks without a sync
        self.full_len_host: List[int] = [0] * batch
        self.seen_tokens = 0
        # `logical_cap`: the capacity the REQUEST asked for (prompt + new tokens + 1).  A pooled slab may be larger; everything that
        # shapes the computation (split-KV factor) is derived from the logical capacity, so results do not depend on pooling history
        self.logical_cap = self.t_cap
        self.sparse_cap = self.t_cap  # host-known upper bound of lens[1] (set by the prefill: logical_cap minus the dropped image tokens)
        # Round 4: what the decode attention is SCHEDULED for (split-KV factor, and with it the fused q|k|v + attention launch) is the number of
        # keys the steps about to be enqueued can actually attend, not the capacity the request reserved: `full_bound` is exact on the host
        # (every row grows by one per token), `sparse_bound` = the evicted group's lengths as last OBSERVED on the device (generate(): a
        # non-blocking copy per chunk of steps, read one chunk late so the queue never drains) + the steps enqueued since.  Reference
        # semantics being served: the cache only grows where `cache_decision` says so (DML:2377-2391, cache_utils.py:153-164) -- BASELINE
        # configs[4] reserves 1588 slots in layers >= 2 and fills 259.  None = no tighter bound known (capacities apply).
        self.full_bound = None
        self.sparse_bound = None
        # rows of at most this many keys run ONE workgroup per (row, head).  The model raises it for batch-1 decoding on the fused
        # q|k|v + attention launch, where the slab part of the attention hides under the weight stream (tools/bench_qkv_attn.py)
        self._sch = None  # decode schedule state (sched_*)
        self.single_split_max_keys = _SINGLE_SPLIT_MAX_KEYS
        self.fused_single_keys = 256  # see fused_attn_splits
        self.min_keys_per_split = 64  # a split workgroup is given at least this many keys (tests lower it to force split launches on tiny rows)

    def n_splits(self, layer_idx: int, rows_times_heads: int, max_splits: int = 32) -> int:
        """Split-KV factor of the decode attention (tools/bench_attn_decode.py sweep): enough workgroups to cover the 256 CUs
        (rows x heads x splits >= 256), never fewer than ~64 keys per workgroup, judged on the host-known length bound."""
        cap = self.key_bound(self.group(layer_idx))
        want = max(1, 256 // max(1, rows_times_heads))
        if cap <= self.single_split_max_keys:
            return 1
        return max(1, min(max_splits, want, -(-cap // self.min_keys_per_split)))

    def key_bound(self, group: int) -> int:
        """Host-known upper bound of the keys a decode step enqueued now attends in this length group (new token included)."""
        if group == 0:
            return self.logical_cap if self.full_bound is None else min(self.logical_cap, self.full_bound)
        cap = min(self.sparse_cap, self.logical_cap)
        return cap if self.sparse_bound is None else min(cap, self.sparse_bound)

    # ---- the decode schedule: a deterministic function of (decode step index, lengths observed at chunk boundaries) ----
    # Steps are enqueued in chunks (4, 4, then `sync_every`); after every chunk the evicted group's longest row is OBSERVED, and chunk i is
    # scheduled from the observation made after chunk i-2 (one chunk late: generate() reads it from a non-blocking copy without draining the
    # launch queue) plus the steps enqueued since.  forward()-driven loops (the reference's own driver, BLTM:310-337) follow the very same rule
    # step by step, so generate() and a forward() loop replay the same kernels on the same data -- bit-identical, as before round 4.
    def sched_begin(self, full0: int, sparse0: int, sync_every: int = 8):
        """After a prefill (or on a cache of unknown history): `full0` / `sparse0` = longest row of the two length groups, one token produced."""
        self._sch = dict(S=max(1, int(sync_every)), full0=int(full0), obs=[(1, int(sparse0))], produced=1, chunk_end=1, chunks=0)

    def sched_active(self) -> bool:
        return getattr(self, "_sch", None) is not None

    def sched_at_boundary(self) -> bool:
        return self._sch["produced"] == self._sch["chunk_end"]

    def sched_observe(self, produced: int, sparse_max: int):
        """The evicted group's longest row when `produced` tokens had been produced (recorded at a chunk boundary)."""
        self._sch["obs"].append((int(produced), int(sparse_max)))
        del self._sch["obs"][:-3]

    def sched_chunk(self) -> int:
        """Start the next chunk at the current position: sets the bounds its steps are scheduled with, returns its nominal length."""
        sc = self._sch
        n = min(sc["S"], 4) if sc["chunks"] < 2 else sc["S"]
        p = sc["produced"]
        # the newest observation the rule may use was made when the chunk BEFORE the previous one ended
        usable = [o for o in sc["obs"] if o[0] <= sc.get("prev_start", 1)]
        at, val = usable[-1] if usable else sc["obs"][0]
        self.set_bounds(sc["full0"] + p - 1 + n, val + (p - at) + n)
        sc["prev_start"], sc["chunk_end"], sc["chunks"] = p, p + n, sc["chunks"] + 1
        return n

    def sched_advance(self, n: int):
        self._sch["produced"] += int(n)
        if self._sch["produced"] > self._sch["chunk_end"]:
            raise RuntimeError("decode schedule: advanced past the end of the current chunk")
        # a chunk cut short by max_new_tokens simply ends the generation; nothing to fix up

    def sched_drop(self):
        self._sch = None
        self.set_bounds(None, None)

    def fused_attn_splits(self, layer_idx: int, max_splits: int = 4) -> int:
        """Attention workgroups per head INSIDE the fused q|k|v + attention launch (dl_gemv_qkv_attn).  One while the row is short enough for its later
        K/V trips to hide under the weight stream (`fused_single_keys`, set by the model); beyond that one per 128 keys of the bound -- a workgroup
        holds two 64-key trips in registers while it waits for q, so the whole row is on chip before q arrives (tools/bench_qkv_attn.py)."""
        bound = self.key_bound(self.group(layer_idx))
        if bound <= self.fused_single_keys:
            return 1
        return max(1, min(max_splits, 4, -(-bound // 128)))  # (4 = the kernel's kQaMaxSplits, whatever the caller asks for)

    def set_bounds(self, full_bound, sparse_bound):
        self.full_bound = None if full_bound is None else int(full_bound)
        self.sparse_bound = None if sparse_bound is None else int(sparse_bound)

    @staticmethod
    def spec_chunk(n_splits: int) -> int:
        """`chunk_keys` of dl_attn_decode_rope.  With ONE split per (row, head) the key range is the whole row whatever the chunk says, so the
        speculative form (first K/V rows requested before kv_len has arrived: any slot < t_cap is readable) is bit-identical to the plain one
        and saves the dependent round trip: 24.9 -> 23.2 us on the B=32 ragged batch (tools/bench_attn_decode.py).  With more splits the chunk
        boundaries would move, so those launches keep the length-derived ranges."""
        return 256 if n_splits == 1 else 0

    eight_wave_single_split = True  # tests switch it off to compare the stand-alone attention with dl_gemv_qkv_attn (four waves) bit for bit

    @staticmethod
    def keys_in_flight(n_splits: int, rows_times_heads: int) -> int:
        """`keys_in_flight` of dl_attn_decode_rope: single-split launches of few (row, head) pairs (decode batches <= 4) run eight waves per
        workgroup -- 128 keys per trip, two trips in flight: 8.9 vs 9.7 us at 226 keys; with more rows the four-wave form keeps more
        workgroups resident per CU (B=32 ragged: 22.6 vs 28.5 us)."""
        return 128 if n_splits == 1 and rows_times_heads <= 128 and KVSlabCache.eight_wave_single_split else 64

    # ---- which length vector a layer uses ----
    def group(self, layer_idx: int) -> int:
        return 0 if layer_idx < self.sparse_layer else 1

    def len_of_layer(self, layer_idx: int) -> torch.Tensor:
        return self.lens[self.group(layer_idx)]

    @property
    def len_full(self):
        return self.lens[0]

    @property
    def len_sparse(self):
        return self.lens[1]

    def ensure_capacity(self, extra_tokens: int):
        """Grow the slab (copy) if max_b full length + extra_tokens would not fit.  Returns True if it grew."""
        need = max(self.full_len_host) + extra_tokens
        if need <= self.t_cap:
            return False
        new_cap = max(need, int(self.t_cap * 1.5) + 16)
        new = torch.empty((self.n_layers, 2, self.batch, self.n_kv_heads, new_cap, self.head_dim), dtype=self.dtype, device=self.device)
        new[:, :, :, :, : self.t_cap, :] = self.slab
        self.slab = new
        self.sparse_cap += new_cap - self.t_cap
        self.logical_cap += new_cap - self.t_cap
        self.t_cap = new_cap
        self.k = [self.slab[i, 0] for i in range(self.n_layers)]
        self.v = [self.slab[i, 1] for i in range(self.n_layers)]
        return True

    # ---- reference-compatible surface ----
    def __len__(self):
        return self.n_layers

    def get_seq_length(self, layer_idx: int = 0) -> int:  # cache_utils.py:272-276 (padded length)
        return self.padded_length(layer_idx)

    def padded_length(self, layer_idx: int) -> int:
        if self.group(layer_idx) == 0:
            return max(self.full_len_host) if self.batch else 0
        return int(self.lens[1].max().item())  # device -> host copy, only when the caller asks

    @property
    def true_cache_length(self):  # cache_utils.py:79, L x [B], CPU int64 like the reference (cache_utils.py:148)
        host = self.lens.to("cpu", torch.int64)
        return [host[self.group(i)].clone() for i in range(self.n_layers)]

    def to_legacy_cache(self):  # cache_utils.py:295-302
        return self

    def __getitem__(self, idx):
        if idx == 0:
            return _LayerViews(self)
        if idx == 1:
            return self.true_cache_length
        raise IndexError("legacy cache tuple has two entries: (per-layer (K, V), true_cache_length)")

    def __iter__(self):
        yield self[0]
        yield self[1]

    # ---- import of a genuine legacy tuple (e.g. produced by the reference / oracle) ----
    @classmethod
    def from_legacy_cache(cls, pkv, sparse_layer, t_cap_extra=256, device=None):  # cache_utils.py:304-318
        layers, lens = pkv[0], pkv[1]
        L = len(layers)
        k0 = layers[0][0]
        B, nKV, _, d = k0.shape
        device = device or k0.device
        tmax = max(int(layers[i][0].shape[2]) for i in range(L))
        c = cls(L, sparse_layer, B, nKV, d, tmax + t_cap_extra, k0.dtype, device)
        for i in range(L):
            k, v = layers[i]
            c.k[i][:, :, : k.shape[2], :] = k.to(device)
            c.v[i][:, :, : v.shape[2], :] = v.to(device)
        c.lens[0] = torch.as_tensor(lens[0]).to(device=device, dtype=torch.int32)
        c.lens[1] = torch.as_tensor(lens[L - 1]).to(device=device, dtype=torch.int32)
        c.full_len_host = [int(x) for x in torch.as_tensor(lens[0]).tolist()]
        c.seen_tokens = max(c.full_len_host)
        return c
