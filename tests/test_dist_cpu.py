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
    t = dd.max_over_ranks(float(rank + 1), "cpu")
    dd.barrier()
    if rank == 0:
        q.put((all_ids, all_logits, t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 4, 1])
def test_dp_equals_single_process(n_items):
    items = list(range(10, 10 + n_items))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, items, q)) for r in range(2)]
    for p in procs:
        p.start()
    all_ids, all_logits, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = [_fake_generate(i) for i in items]
    assert all_ids.shape[0] == n_items and all_logits.shape[0] == n_items
    for i, (tok, lg) in enumerate(single):
        assert torch.equal(all_ids[i, : tok.numel()], tok)
        assert (all_ids[i, tok.numel() :] == -1).all()
        assert torch.equal(all_logits[i], lg)
    assert t == 2.0


def test_get_chunk_matches_reference_rule():
    from dynamic_llava_amd.dist import get_chunk, split_list

    lst = list(range(10))
    assert split_list(lst, 4) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert get_chunk(lst, 4, 3) == [9] and get_chunk(lst, 8, 7) == []
    assert sum((list(get_chunk(lst, 3, k)) for k in range(3)), []) == lst
