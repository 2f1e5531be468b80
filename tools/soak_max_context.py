"""One request to the maximum context (LLaVA-1.5: 4096 positions): 631-token prompt + 3460 new tokens at B=1, hipGraph replay against eager
launches token for token, the schedule's split factors on the way, final KV lengths.
    python tools/soak_max_context.py"""
import sys, time
sys.path.insert(0, ".")
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig
import bench
cfg = DynamicLlavaConfig()
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
prompt, images = bench.make_inputs(cfg, torch.device("cuda"), torch.bfloat16)
n = cfg.max_position_embeddings - prompt.shape[1] - 576 + 1 - 5 if hasattr(cfg, "max_position_embeddings") else 3460
n = min(n, 3460)
outs = []
for graph in (True, False):
    model.use_hip_graph = graph
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = model.generate(prompt, images=images, max_new_tokens=n, eos_token_id=None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lens = [int(t[0]) for t in (model.last_cache[1][0], model.last_cache[1][-1])]
    print(f"graph={graph}: {n} new tokens in {(t1 - t0) * 1e3:.0f} ms ({(t1 - t0) / n * 1e3:.3f} ms/token incl. prefill), kv_len full/sparse {lens[0]}/{lens[1]}, decode graphs {len(model._dstate.graphs) if model._dstate is not None else 0}", flush=True)
    outs.append(o)
model.check_device_errors()
same = torch.equal(outs[0], outs[1])
print("graph == eager token for token:", same, "| peak memory GB:", round(torch.cuda.max_memory_allocated() / 2**30, 1))
assert same and outs[0].shape[1] == n
