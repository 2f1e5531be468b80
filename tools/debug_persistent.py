"""Debug driver for the persistent decode step: generate() under the four (persistent, graph) settings, dumping the control words."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fixtures as fx
from dynamic_llava_amd.builder import build_from_state_dict
from dynamic_llava_amd.config import DynamicLlavaConfig

dtype = torch.bfloat16
cfg = fx.llava7b_config(num_hidden_layers=4)
cfg.vocab_size = 4096
sd = fx.make_state_dict(cfg, seed=23, predictor_gain=50.0)
model = build_from_state_dict(DynamicLlavaConfig.from_namespace(cfg), sd, None, dtype=dtype, device="cuda")
g = torch.Generator().manual_seed(9)
ids = fx.make_prompt(cfg, 35, 20, seed=5)[None]
feats = torch.randn(1, 576, cfg.hidden_size, generator=g).to(dtype)
ref = None
for persistent in (False, True):
    for graph in (False, True):
        model.use_persistent_decode, model.use_hip_graph = persistent, graph
        try:
            out = model.generate(ids.cuda(), image_features=feats.cuda(), max_new_tokens=40, eos_token_id=None).cpu()
        except Exception as e:
            print("FAILED", persistent, graph, repr(e)[:200])
            st = model._dstate
            torch.cuda.synchronize()
            print("psync[:40]", [hex(x & 0xffffffff) for x in st.psync[:40].cpu().tolist()])
            continue
        if ref is None:
            ref = out
        print(persistent, graph, "equal to ref:", torch.equal(out, ref), out[0, :8].tolist(), [int(t[0]) for t in (model.last_cache[1][0], model.last_cache[1][-1])])
        st = model._dstate
        if st.psync is not None:
            print("   psync[:8]", [hex(x & 0xffffffff) for x in st.psync[:8].cpu().tolist()])
