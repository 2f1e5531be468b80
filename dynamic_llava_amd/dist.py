"""Data-parallel sharding of independent requests across the GPUs of one node.

Reference: share-nothing processes, one per GPU, each taking a contiguous chunk of the question file
(`get_chunk`, llava/dynamic_eval/model_vqa_loader.py:30-38, launched by run/dynamic_eval/eval_for_vqav2.sh:11-21)
and results merged by `cat` of JSONL files (eval_for_vqav2.sh:25-33).  There is no collective on the reference's
inference path.

Here: one process per GPU under torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU
tests), weights replicated (13.5 GB bf16 << 288 GB), the same contiguous-chunk rule, and ONE small collective per
request batch: an all-gather of the last-token logits [B_local, V] fp32 (4.1 MB per rank at B_local=32) and of
the generated ids [B_local, T_new] int64.  No collective sits inside the decode loop.  Messages this small are
latency-bound, so they are issued as a single all_gather (one message per xGMI peer link) rather than chunked.
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "DL_FORCE_DEVICE" in os.environ:  # test hook: several ranks on ONE GPU (only possible with the gloo backend)
        local = int(os.environ["DL_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("DL_DIST_BACKEND")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def split_list(lst: Sequence, n: int) -> List[Sequence]:
    """model_vqa_loader.py:30-33 -- contiguous chunks of ceil(len/n)."""
    chunk = math.ceil(len(lst) / n) if len(lst) else 0
    return [lst[i : i + chunk] for i in range(0, len(lst), chunk)] if chunk else []


def get_chunk(lst: Sequence, n: int, k: int) -> Sequence:
    """model_vqa_loader.py:36-38 (rank k of n; ranks past the end get nothing)."""
    chunks = split_list(lst, n)
    return chunks[k] if k < len(chunks) else lst[:0]


def all_gather_rows(x: torch.Tensor, pad_value=0) -> torch.Tensor:
    """All-gather along dim 0 of a [B_local, ...] tensor whose B_local (and trailing dim 1, e.g. T_new) may differ
    between ranks; returns the concatenation in rank order (== the single-process order under get_chunk)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    world = dist.get_world_size()
    shape = torch.tensor(list(x.shape) + [0] * (4 - x.dim()), dtype=torch.int64, device=x.device)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    shapes = [s.tolist()[: x.dim()] for s in shapes]
    mx = [max(s[i] for s in shapes) for i in range(x.dim())]
    buf = torch.full(mx, pad_value, dtype=x.dtype, device=x.device)  # padded to the max shape, pad_value elsewhere
    buf[tuple(slice(0, n) for n in x.shape)] = x
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return torch.cat([o[: s[0]] for o, s in zip(outs, shapes)], dim=0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
