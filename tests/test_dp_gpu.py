"""GPU: the data-parallel path with the REAL model (BASELINE configs[3]'s per-rank shape).  Two ranks share the one
GPU of the test box (gloo backend + the DL_FORCE_DEVICE hook; on the 8-GPU node the same code runs one rank per GPU over RCCL):
64 ragged requests are split by the reference's contiguous-chunk rule (model_vqa_loader.py:30-38) into 32 per rank, each rank
runs sparsified prefill + greedy decode with eviction on its chunk, and ONE all-gather returns last-token logits + generated ids.
The gathered result must equal, bit for bit, the same chunks run one after the other by a single process."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_REQ, PER_RANK, STEPS = 64, 32, 10


def _requests(cfg, dtype):
    from oracle import fixtures as fx

    g = torch.Generator().manual_seed(1)
    n_q = torch.randint(8, 65, (N_REQ,), generator=g).tolist()  # question lengths ~U[8,64]
    prompts = [fx.make_prompt(cfg, 35, n_q[b], seed=b) for b in range(N_REQ)]
    feats = torch.randn(N_REQ, 576, cfg.hidden_size, generator=g).to(dtype)
    return prompts, feats


def _run_chunk(model, prompts, feats, idx):
    W = max(prompts[i].shape[0] for i in idx)
    ids = torch.zeros(len(idx), W, dtype=torch.long)
    am = torch.zeros(len(idx), W, dtype=torch.long)
    for r, i in enumerate(idx):
        ids[r, : prompts[i].shape[0]] = prompts[i]
        am[r, : prompts[i].shape[0]] = 1
    out = model.generate(ids.cuda(), attention_mask=am.cuda(), image_features=feats[list(idx)].cuda(), max_new_tokens=STEPS, eos_token_id=None)
    return out.cpu(), model.last_prefill_logits.float().cpu().clone(), model.last_cache[1][-1].clone()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      DL_FORCE_DEVICE="0", DL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from dynamic_llava_amd import dist as dd
    from dynamic_llava_amd.builder import build_from_state_dict, build_random_model
    from dynamic_llava_amd.config import DynamicLlavaConfig
    from oracle import fixtures as fx

    r, w, local = dd.init_distributed()
    torch.cuda.set_device(local)
    dtype = torch.bfloat16
    cfg = fx.llava7b_config(num_hidden_layers=3)
    cfg.vocab_size = 4096
    sd = fx.make_state_dict(cfg, seed=11, predictor_gain=50.0)
    model = build_from_state_dict(DynamicLlavaConfig.from_namespace(cfg), sd, None, dtype=dtype, device="cuda")
    prompts, feats = _requests(cfg, dtype)
    mine = dd.get_chunk(list(range(N_REQ)), w, r)
    assert len(mine) == PER_RANK
    # the two ranks share ONE GPU here and run AT THE SAME TIME (round 2 let them take turns: with both running, rows drifted apart.  Root
    # cause, round 3: not a race in this package and not the library GEMMs -- on gfx950 a packed fp32 instruction whose low half reads
    # src1's high register (hipcc's SLP vectoriser emits it; the text predictor's two-neuron dot products used it) is mis-executed while
    # a wave of ANOTHER kernel runs MFMA on the same SIMD, which only happens when two processes / streams share the device.  The
    # library is built without that instruction form now: DESIGN.md section 5, tools/pkfma_probe.hip, tools/tp_race_probe.py)
    dd.barrier()
    ids, logits, lens = _run_chunk(model, prompts, feats, mine)
    torch.cuda.synchronize()
    dd.barrier()
    all_logits, all_ids = dd.gather_results(logits, ids, max_rows=PER_RANK, max_new_tokens=STEPS)  # the product's ONE collective per batch
    all_lens = dd.all_gather_rows(lens)
    # bench.py's builder: every rank must hold identical random-init weights (seeded device generator)
    rm = build_random_model(DynamicLlavaConfig(num_hidden_layers=1, vocab_size=512), dtype=dtype, device="cuda", seed=0, predictor_gain=50.0)
    digest = torch.stack([p.detach().double().sum().cpu() for p in rm.parameters()])
    digests = dd.all_gather_rows(digest[None])
    dd.barrier()
    if rank == 0:
        ok_w = bool(torch.equal(digests[0], digests[1]))
        # single process, same chunks one after the other (the reference's share-nothing processes, concatenated)
        ref = [_run_chunk(model, prompts, feats, dd.get_chunk(list(range(N_REQ)), w, k)) for k in range(w)]
        ref_ids = torch.cat([x[0] for x in ref])
        ref_logits = torch.cat([x[1] for x in ref])
        ref_lens = torch.cat([x[2] for x in ref])
        kept = (ref_lens - torch.tensor([35 + 115 + p.shape[0] - 36 for p in prompts])).tolist()
        q.put(dict(ids=bool(torch.equal(all_ids, ref_ids)), logits=bool(torch.equal(all_logits, ref_logits)), lens=bool(torch.equal(all_lens, ref_lens)),
                   shape=tuple(all_ids.shape), weights=ok_w, lens_diff=(all_lens - ref_lens).tolist(), evicted_some=bool(0 < sum(kept) < N_REQ * (STEPS - 1))))
    dd.barrier()
    torch.distributed.destroy_process_group()


def test_dp2_real_model_equals_single_process_bit_exact():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert res["shape"] == (N_REQ, STEPS)
    assert res["weights"], "build_random_model must give every rank identical weights"
    assert res["ids"] and res["logits"] and res["lens"], res
    assert res["evicted_some"], "the per-row eviction must be exercised both ways"


_NCCL_WS1 = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from dynamic_llava_amd import dist as dd
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DL_FORCE_DEVICE", "DL_DIST_BACKEND"):
    os.environ.pop(k, None)
rank, world, local = dd.init_distributed("nccl", force=True)   # RCCL process group of ONE rank on cuda:0
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", local)
n = []
real = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (n.append(1), real(*a, **k))[1]
g = torch.Generator().manual_seed(0)
logits = torch.randn(32, 32000, generator=g).to(dev)              # configs[3]'s per-rank payload: 32 x 32000 fp32 = 4.1 MB
ids = torch.randint(0, 32000, (32, 17), generator=g).to(dev)
lg, tk = dd.gather_results(logits, ids, max_rows=32, max_new_tokens=32, force=True)
torch.cuda.synchronize()
assert len(n) == 1, n                                             # one collective carried both tensors
assert lg.is_cuda and torch.equal(lg, logits) and torch.equal(tk, ids)
x = torch.arange(12, device=dev, dtype=torch.float32).view(3, 4)
assert torch.equal(dd.all_gather_rows(x, force=True), x)          # shape exchange + payload, device tensors
assert dd.max_over_ranks(1.5, dev) == 1.5
t = torch.tensor([2.0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 2.0
dist.barrier(); dist.destroy_process_group()
print("nccl-ws1-ok")
"""


def test_nccl_backend_world_size_one_pushes_device_tensors_through_rccl():
    """`backend="nccl"` (RCCL) had never executed.  A 1-GPU box cannot host two RCCL ranks, but a process group of ONE
    rank loads librccl, creates the communicator on cuda:0 and runs the very collective of the DP path (`gather_results`: one
    all_gather_into_tensor of logits + ids + shapes) and the timing all-reduce on DEVICE tensors."""
    import subprocess

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_WS1 % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "nccl-ws1-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_self_launches_two_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no launcher (VERDICT r3 missing #2): bench.py starts its ranks itself.  On this 1-GPU box the two ranks
    share the device through the DL_FORCE_DEVICE hook (gloo); on the 8-GPU node the same call runs one rank per GPU over RCCL.  A shallow
    model keeps it short: the JSON line is marked INVALID by bench.py itself, which is fine -- this test checks the launch and the DP legs."""
    import json
    import subprocess

    env = dict(os.environ, DL_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "4", "--new-tokens", "8",
                        "--no-cpu-baseline", "--no-ref-gpu"], capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["dp_rows_identical"] is True
    assert res["configs3"]["gathered_rows"] == 64 and res["configs3"]["dp_equals_rerun_of_last_rank"] is True


def test_bench_self_launches_eight_ranks_on_one_gpu():
    """the shape of the 8-GPU run the driver will make -- `python bench.py --gpus 8` -- with all eight ranks on the one GPU of the test
    box (DL_FORCE_DEVICE hook, gloo; on the 8-GPU node the same call is one rank per GPU over RCCL).  Two decoder layers keep it short (bench.py marks
    the line INVALID itself).  Checked: the line's view of the process group (backend, world size, eight ranks, one device), identical results on all
    ranks, configs[3] with 256 requests split 32 per rank by the reference's chunk rule, ONE collective for the batch, gathered rows == a re-run."""
    import json
    import subprocess

    env = dict(os.environ, DL_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--layers", "2", "--new-tokens", "4",
                        "--no-cpu-baseline", "--no-ref-gpu"], capture_output=True, text=True, env=env, timeout=2400, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 8 and res["config"]["parallelism"] == "dp8" and "error" not in res
    d = res["config"]["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 8 and [x["rank"] for x in d["ranks"]] == list(range(8)) and d["distinct_devices"] == 1
    assert res["config"]["dp_rows_identical"] is True
    c3 = res["configs3"]
    assert c3["gathered_rows"] == 256 and c3["rows_per_rank"] == [32] * 8 and c3["collectives_per_batch"] == 1
    assert c3["dp_equals_rerun_of_last_rank"] is True, c3
