"""CPU, world_size 2 over gloo: the data-parallel wrapper (dynamic_llava_amd/dist.py).
DP result must equal the single-process result: contiguous chunks (model_vqa_loader.py:30-38) gathered in rank order."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_generate(item):
    """Stand-in for per-request work: ragged number of new tokens + a 'last logits' row, deterministic in the item."""
    g = torch.Generator().manual_seed(int(item))
    n_new = 1 + int(item) % 5
    return torch.randint(0, 100, (n_new,), generator=g), torch.randn(16, generator=g)


def _worker(rank, world, port, items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dynamic_llava_amd import dist as dd

    r, w, _ = dd.init_distributed("gloo")
    mine = dd.get_chunk(items, w, r)
    outs = [_fake_generate(i) for i in mine]
    T = max([o[0].numel() for o in outs], default=0)
    ids = torch.full((len(outs), T), -1, dtype=torch.int64)
    for i, o in enumerate(outs):
        ids[i, : o[0].numel()] = o[0]
    logits = torch.stack([o[1] for o in outs]) if outs else torch.zeros(0, 16)
    all_ids = dd.all_gather_rows(ids, pad_value=-1)
    all_logits = dd.all_gather_rows(logits)
    # the product's per-batch collective: logits + ids + true shapes in ONE message, sized by host-known bounds (chunk rule, max_new_tokens)
    per = -(-len(items) // w)
    one_lg, one_ids = dd.gather_results(logits, ids, max_rows=per, max_new_tokens=5, pad_token_id=-1)
    # the same message with the collective timed by itself (bench.py's configs[3] leg): same result, a duration, the per-rank message size
    coll = {}
    timed_lg, timed_ids = dd.gather_results(logits, ids, max_rows=per, max_new_tokens=5, pad_token_id=-1, timing=coll)
    assert torch.equal(timed_lg, one_lg) and torch.equal(timed_ids, one_ids)
    assert coll["world_size"] == w and coll["collective_us"] > 0 and coll["backend"] == "gloo"
    assert coll["bytes_per_rank"] >= per * 16 * 4 + per * 5 * 8 and coll["gathered_bytes"] == w * coll["bytes_per_rank"]
    lo, hi = dd.min_max_over_ranks(float(rank + 1), "cpu")
    assert (lo, hi) == (1.0, float(w))
    bounded = dd.all_gather_rows(ids, pad_value=-1, max_shape=(per, 5))
    t = dd.max_over_ranks(float(rank + 1), "cpu")
    view = dd.describe()
    dd.barrier()
    if rank == 0:
        assert view["backend"] == "gloo" and view["world_size"] == w and [r_["rank"] for r_ in view["ranks"]] == list(range(w)), view
        q.put((all_ids.numpy(), all_logits.numpy(), t, bool(  # by value: a shared-memory tensor handle dies with this process
            torch.equal(one_lg, all_logits) and torch.equal(one_ids, all_ids) and torch.equal(bounded, all_ids))))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_items,world", [(7, 2), (4, 2), (1, 2), (29, 8), (256, 8), (5, 8)])
def test_dp_equals_single_process(n_items, world):
    """world 8 = the reference's eight forked processes (run/dynamic_eval/eval_for_vqav2.sh:11-21): 256 requests -> 32 per rank (configs[3]); 29 -> a
    ragged last chunk; 5 -> ranks past the end get nothing and still take part in the collective."""
    items = list(range(10, 10 + n_items))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, items, q)) for r in range(world)]
    for p in procs:
        p.start()
    all_ids, all_logits, t, one_message_same = q.get(timeout=300)
    all_ids, all_logits = torch.from_numpy(all_ids), torch.from_numpy(all_logits)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = [_fake_generate(i) for i in items]
    assert all_ids.shape[0] == n_items and all_logits.shape[0] == n_items
    for i, (tok, lg) in enumerate(single):
        assert torch.equal(all_ids[i, : tok.numel()], tok)
        assert (all_ids[i, tok.numel() :] == -1).all()
        assert torch.equal(all_logits[i], lg)
    assert t == float(world)
    assert one_message_same, "gather_results / bounded all_gather_rows (one collective) must equal the two-collective gather"


def _count_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from dynamic_llava_amd import dist as dd

    dd.init_distributed("gloo")
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    lg, ids = dd.gather_results(torch.randn(3, 8), torch.arange(6).view(3, 2) + 10 * rank, max_rows=3, max_new_tokens=4)
    if rank == 0:
        q.put((len(calls), tuple(lg.shape), ids.tolist()))
    dd.barrier()
    dist.destroy_process_group()


def test_gather_results_is_one_collective():
    """one all-gather per request batch -- logits, ids and their true shapes travel in a single all_gather_into_tensor."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_count_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n_calls, shape, ids = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n_calls == 1 and shape == (6, 8)
    assert ids == [[0, 1], [2, 3], [4, 5], [10, 11], [12, 13], [14, 15]]


def test_forced_world_size_one_collective_runs():
    """`force=True` really issues the collective at world size 1 (the GPU suite uses it to push a device tensor through RCCL on a 1-GPU box)."""
    import subprocess

    code = ("import sys; sys.path.insert(0, %r); import torch; from dynamic_llava_amd import dist as dd; import torch.distributed as dist\n"
            "dd.init_distributed('gloo', force=True); assert dist.is_initialized() and dist.get_world_size() == 1\n"
            "n = []; real = dist.all_gather_into_tensor; dist.all_gather_into_tensor = lambda *a, **k: (n.append(1), real(*a, **k))[1]\n"
            "lg, ids = dd.gather_results(torch.randn(2, 4), torch.arange(6).view(2, 3), 2, 5, force=True)\n"
            "assert len(n) == 1 and lg.shape == (2, 4) and ids.tolist() == [[0, 1, 2], [3, 4, 5]]\n"
            "x = torch.randn(2, 4); assert dd.all_gather_rows(x) is x  # no force, world 1: no collective\n"
            "dist.destroy_process_group(); print('ok')" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_get_chunk_matches_reference_rule():
    from dynamic_llava_amd.dist import get_chunk, split_list

    lst = list(range(10))
    assert split_list(lst, 4) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert get_chunk(lst, 4, 3) == [9] and get_chunk(lst, 8, 7) == []
    assert sum((list(get_chunk(lst, 3, k)) for k in range(3)), []) == lst


def test_bench_abort_is_one_json_line():
    """a multi-rank run that must not report a number still prints ONE parseable line: the metric, value null, the reason, the
    per-rank detail and the partial timings."""
    import json

    sys.path.insert(0, ROOT)
    import bench

    line = bench.abort_line("data-parallel ranks produced different results for identical requests", {"per_rank": [{"rank": 0}, {"rank": 1, "ids_equal_rank0": False}]},
                            {"n_gpus": 2, "ms_per_step": 161.3, "dist": {"backend": "nccl", "world_size": 2}})
    assert "\n" not in line
    d = json.loads(line)
    assert d["metric"] == bench.METRIC and d["value"] is None and "different results" in d["error"]
    assert d["detail"]["per_rank"][1]["ids_equal_rank0"] is False and d["partial"]["dist"]["backend"] == "nccl"


def test_bench_refuses_more_ranks_than_gpus_with_a_json_line():
    """`python bench.py --gpus 2` on a box without GPUs: exit code 2 and an "error" line instead of a bare SystemExit string."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DL_FORCE_DEVICE")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-1500:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["value"] is None and "visible GPU" in d["error"] and d["detail"]["gpus"] == 2 and d["partial"]["n_gpus"] == 2
