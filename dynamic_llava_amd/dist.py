"""Data-parallel sharding of independent requests across the GPUs of one node.

Reference: share-nothing processes, one per GPU, each taking a contiguous chunk of the question file
(`get_chunk`, llava/dynamic_eval/model_vqa_loader.py:30-38, launched by run/dynamic_eval/eval_for_vqav2.sh:11-21)
and results merged by `cat` of JSONL files (eval_for_vqav2.sh:25-33).  There is no collective on the reference's
inference path.

Here: one process per GPU under torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU
tests), weights replicated (13.5 GB bf16 << 288 GB), the same contiguous-chunk rule, and ONE collective per
request batch: `gather_results` packs the last-token logits [B_local, V] fp32 (4.1 MB per rank at B_local=32), the
generated ids [B_local, T_new] int64 and a 64-byte header with the true shapes into one byte message per rank and
issues a single `all_gather_into_tensor` (one message per xGMI peer link; messages this small are latency-bound,
so they are never chunked).  The message size is host-known without communication: B_local <= ceil(n / world) by
the chunk rule and T_new <= max_new_tokens.  No collective sits inside the decode loop.
"""
from __future__ import annotations

import math
import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_HDR_WORDS = 8  # int64 header words per packed tensor: ndim, then up to 7 dims
_ALIGN = 16


def init_distributed(backend: str | None = None, force: bool = False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torchrun).  Returns (rank, world, local_rank).
    `force`: create the process group even at world size 1 (the GPU test suite pushes a device tensor through RCCL that way)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "DL_FORCE_DEVICE" in os.environ:  # test hook: several ranks on ONE GPU (only possible with the gloo backend)
        local = int(os.environ["DL_FORCE_DEVICE"])
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force and world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = backend or os.environ.get("DL_DIST_BACKEND")
        if backend is None:
            backend = "nccl" if (torch.cuda.is_available() and "DL_FORCE_DEVICE" not in os.environ) else "gloo"
        if backend == "nccl":
            if "DL_FORCE_DEVICE" in os.environ and world > 1:
                raise RuntimeError("DL_FORCE_DEVICE puts several ranks on one GPU, which RCCL refuses: use DL_DIST_BACKEND=gloo with it")
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


class LaunchError(RuntimeError):
    """self_launch cannot start the ranks it was asked for (fewer visible GPUs than ranks and no test hook)."""


def describe(device=None) -> dict:
    """What the process group looks like FROM the collective library's side, for the bench line's `config`: backend, world size
    and, per rank, the device it drives -- gathered with one all_gather_object outside any timed region.  World size 1 without a group: a stub."""
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(),
          "device": (str(device) if device is not None else None),
          "device_name": (torch.cuda.get_device_name(device) if device is not None and torch.cuda.is_available() else None)}
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "ranks": [me]}
    ranks = [None] * dist.get_world_size()
    dist.all_gather_object(ranks, me)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks,
            "distinct_devices": len({(r["device"]) for r in ranks}), "collective": "one all_gather_into_tensor per request batch (gather_results)"}


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(script: str, argv: Sequence[str], n_procs: int) -> int:
    """`python script --gpus N` without a launcher: start N ranks of `script` on this node the way the reference's shell loop forks
    one process per GPU (run/dynamic_eval/eval_for_vqav2.sh:11-21), here through `torch.distributed.run` so that RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* are set.  With fewer visible GPUs than ranks the test hook DL_FORCE_DEVICE (all ranks on that device, gloo
    backend) must be set by the caller; otherwise this raises instead of silently oversubscribing a device.  Returns the exit code."""
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = dict(os.environ)
    if n_dev < n_procs:
        if "DL_FORCE_DEVICE" not in env:
            raise LaunchError(f"--gpus {n_procs} but only {n_dev} visible GPU(s): one rank per GPU is the contract (set DL_FORCE_DEVICE=<dev> to put "
                              f"every rank on one device over gloo -- a functional test hook, not a measurement)")
        env.setdefault("DL_DIST_BACKEND", "gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_procs) // n_procs)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_procs}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, *argv]
    return subprocess.run(cmd, env=env).returncode


def split_list(lst: Sequence, n: int) -> List[Sequence]:
    """model_vqa_loader.py:30-33 -- contiguous chunks of ceil(len/n)."""
    chunk = math.ceil(len(lst) / n) if len(lst) else 0
    return [lst[i : i + chunk] for i in range(0, len(lst), chunk)] if chunk else []


def get_chunk(lst: Sequence, n: int, k: int) -> Sequence:
    """model_vqa_loader.py:36-38 (rank k of n; ranks past the end get nothing)."""
    chunks = split_list(lst, n)
    return chunks[k] if k < len(chunks) else lst[:0]


def _round_up(n: int, a: int = _ALIGN) -> int:
    return (n + a - 1) // a * a


def _collective_active(force: bool) -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or force)


def all_gather_packed(tensors: Sequence[torch.Tensor], max_shapes: Sequence[Sequence[int]], pad_values: Optional[Sequence] = None,
                      force: bool = False, timing: Optional[dict] = None) -> List[torch.Tensor]:
    """ONE `all_gather_into_tensor` for several [B_local, ...] tensors whose B_local (and trailing dims, e.g. T_new) may differ between
    ranks but are bounded by the host-known `max_shapes`.  Per rank one byte message: a header of true shapes, then every tensor's bytes
    in a section sized for its bound.  Returns, per tensor, the concatenation along dim 0 in rank order (== the single-process order
    under get_chunk), trailing dims padded with `pad_value` to the largest TRUE extent over the ranks (not to the bound)."""
    tensors = [t.contiguous() for t in tensors]
    pad_values = list(pad_values) if pad_values is not None else [0] * len(tensors)
    if not _collective_active(force):
        if timing is not None:
            timing.update(collective_us=None, bytes_per_rank=0, world_size=1, note="no process group: nothing was sent")
        return list(tensors)
    for t, ms in zip(tensors, max_shapes):
        if t.dim() != len(ms) or t.dim() >= _HDR_WORDS or any(a > b for a, b in zip(t.shape, ms)):
            raise ValueError(f"tensor of shape {tuple(t.shape)} does not fit the announced bound {tuple(ms)}")
    world, dev = dist.get_world_size(), tensors[0].device
    hdr_bytes = _round_up(8 * _HDR_WORDS * len(tensors))
    sec = [_round_up(int(math.prod(ms)) * t.element_size()) for t, ms in zip(tensors, max_shapes)]
    off = [hdr_bytes]
    for s in sec:
        off.append(off[-1] + s)
    msg = torch.zeros(off[-1], dtype=torch.uint8, device=dev)
    hdr = torch.tensor([w for t in tensors for w in ([t.dim()] + list(t.shape) + [0] * (_HDR_WORDS - 1 - t.dim()))], dtype=torch.int64)
    msg[: hdr.numel() * 8] = hdr.view(torch.uint8).to(dev)
    for t, o in zip(tensors, off):
        nb = t.numel() * t.element_size()
        if nb:
            msg[o : o + nb] = t.reshape(-1).view(torch.uint8)
    out = torch.empty(world * off[-1], dtype=torch.uint8, device=dev)  # flat: the layout every backend's all_gather_into_tensor accepts
    if timing is None:
        dist.all_gather_into_tensor(out, msg)
    else:
        # the collective BY ITSELF (packing and unpacking excluded): device events on the caller's stream around the call (the process group's own stream is
        # joined to it before the call returns control of `out`), host clock for a host-side backend; to be read against the per-link estimate of
        # SURVEY 8e (4.1 MB per rank: ~27 us direct over xGMI, ~190 us for a ring bound by one link)
        import time as _time

        on_dev = dev.type == "cuda" and dist.get_backend() == "nccl"
        if on_dev:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = _time.perf_counter()
        dist.all_gather_into_tensor(out, msg)
        if on_dev:
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3
        else:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            us = (_time.perf_counter() - t0) * 1e6
        timing.update(collective_us=round(us, 1), bytes_per_rank=int(off[-1]), gathered_bytes=int(world * off[-1]), world_size=world, backend=dist.get_backend(),
                      clock="device events on the caller's stream" if on_dev else "host clock around the call")
    out = out.view(world, off[-1])
    hdrs = out[:, : hdr.numel() * 8].cpu().contiguous().view(torch.int64).view(world, len(tensors), _HDR_WORDS)  # the one device->host copy
    res = []
    for j, (t, o, pv) in enumerate(zip(tensors, off, pad_values)):
        shapes = [hdrs[r, j, 1 : 1 + t.dim()].tolist() for r in range(world)]
        tail = [max(s[i] for s in shapes) for i in range(1, t.dim())]
        parts = []
        for r, s in enumerate(shapes):
            n = int(math.prod(s))
            x = out[r, o : o + n * t.element_size()].view(t.dtype).view(s)
            if list(s[1:]) != tail:
                buf = torch.full([s[0]] + tail, pv, dtype=t.dtype, device=dev)
                buf[tuple(slice(0, k) for k in s)] = x
                x = buf
            parts.append(x)
        res.append(torch.cat(parts, dim=0))
    return res


def gather_results(logits: torch.Tensor, ids: torch.Tensor, max_rows: int, max_new_tokens: int, pad_token_id: int = 0,
                   force: bool = False, timing: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The per-request-batch collective of the DP path: last-token logits [B_local, V] and generated ids [B_local, T_new] of every rank in
    ONE message (see module docstring).  `max_rows` = ceil(n_requests / world) (the chunk rule), `max_new_tokens` bounds T_new."""
    lg, tk = all_gather_packed([logits, ids], [(max_rows, logits.shape[1]), (max_rows, max_new_tokens)], [0, pad_token_id], force=force, timing=timing)
    return lg, tk


def all_gather_rows(x: torch.Tensor, pad_value=0, max_shape: Optional[Sequence[int]] = None, force: bool = False) -> torch.Tensor:
    """All-gather along dim 0 of a [B_local, ...] tensor whose B_local (and trailing dims) may differ between ranks; returns the
    concatenation in rank order.  With a host-known `max_shape` bound: one collective (all_gather_packed).  Without: the bound is agreed
    first with one small all_gather_into_tensor of the shapes (two collectives)."""
    if not _collective_active(force):
        return x
    if max_shape is None:
        world = dist.get_world_size()
        shape = torch.tensor(list(x.shape), dtype=torch.int64, device=x.device)
        shapes = torch.empty(world * x.dim(), dtype=torch.int64, device=x.device)
        dist.all_gather_into_tensor(shapes, shape)
        max_shape = shapes.view(world, x.dim()).max(dim=0).values.tolist()
    return all_gather_packed([x], [max_shape], [pad_value], force=force)[0]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def min_max_over_ranks(value: float, device) -> Tuple[float, float]:
    """(min, max) of a per-rank scalar with ONE all-reduce (MAX over [-v, v]): the spread a slow rank leaves in a weak-scaling run."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value, value
    t = torch.tensor([-value, value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(-t[0].item()), float(t[1].item())


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
