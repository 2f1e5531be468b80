"""Timeline of one persistent decode step (7B, bench prompt): per-phase wall-clock stamps of every workgroup's streaming wave 0
(slots 0-3: phase start, first batches issued, x ready, rows done) and poller 0 (4: start, 5: delta complete, 6: sweep done, 7: x written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.builder import build_random_model
from dynamic_llava_amd.config import DynamicLlavaConfig

L = int(os.environ.get("LAYERS", "8"))
cfg = DynamicLlavaConfig(num_hidden_layers=L)
model = build_random_model(cfg, dtype=torch.bfloat16, device="cuda", seed=0, predictor_gain=50.0)
g = torch.Generator().manual_seed(0)
ids = torch.randint(3, cfg.vocab_size, (55,), generator=g)
prompt = torch.cat([torch.tensor([1]), ids[:34], torch.tensor([-200]), ids[35:]]).long()[None].cuda()
images = torch.randn((1, 3, 336, 336), generator=g).to(torch.bfloat16).cuda()
model.use_hip_graph = False
model.use_persistent_decode = True
n_ph = 2 + 5 * L
G = torch.cuda.get_device_properties(0).multi_processor_count
model._pstamps = torch.zeros((G, n_ph, 8), dtype=torch.int64, device="cuda")
model.generate(prompt, images=images, max_new_tokens=6, eos_token_id=None)
torch.cuda.synchronize()
st = model._pstamps.cpu().double()
st[st == 0] = float("nan")
t0 = float(st[~st.isnan()].min())
st = (st - t0) / 100.0
names = ["embed"] + ["qkv", "attn", "o", "gu", "down"] * L + ["lm_head"]
print("us since step start.  per phase: min/median/max over workgroups of [streamer0: start, issued, x ready, done | poller0: start, delta ok, sweep ok, x written]")
for ph in range(min(n_ph, int(os.environ.get("SHOW", "17")))):
    row = st[:, ph, :]
    if row.isnan().all():
        continue
    def f(c):
        v = row[:, c][~row[:, c].isnan()]
        return "      --        " if v.numel() == 0 else f"{float(v.min()):6.1f}/{float(v.median()):6.1f}/{float(v.max()):6.1f}"
    print(f"{ph:3d} {names[ph]:8s} " + " ".join(f(c) for c in range(8)))
    if names[ph] in ("gu", "down"):
        done = row[:, 3]
        slow = torch.argsort(done, descending=True)[:4].tolist()
        print("      slowest workgroups (done):", [(w, round(float(done[w]), 1), "xready", round(float(row[w, 2]), 1)) for w in slow])
last = st[:, n_ph - 1, :]
print("step span (us):", float(last[~last.isnan()].max()))
