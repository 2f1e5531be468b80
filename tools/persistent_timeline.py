"""Timeline of one persistent decode step (7B, bench prompt): per-phase wall-clock stamps of one workgroup's streaming wave 0 and poller 0."""
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
n_ph = 2 + 5 * L
for wg in (0, 100, 255):
    model._pstamps = torch.zeros((n_ph, 8), dtype=torch.int64, device="cuda")
    model._pstamp_wg = wg
    model.generate(prompt, images=images, max_new_tokens=6, eos_token_id=None)
    torch.cuda.synchronize()
    st = model._pstamps.cpu()
    t0 = int(st[st > 0].min())
    names = ["embed"] + ["qkv", "attn", "o", "gu", "down"] * L + ["lm_head"]
    print(f"== workgroup {wg}: us since step start; streamer0: start issue xready done | poller0: start hint sweep ready")
    for ph in range(min(n_ph, 18)):
        r = [(int(x) - t0) / 100.0 if int(x) > 0 else float("nan") for x in st[ph]]
        print(f"{ph:3d} {names[ph]:8s} " + " ".join(f"{x:8.2f}" for x in r))
    last = [(int(x) - t0) / 100.0 for x in st[n_ph - 1] if int(x) > 0]
    print("   step span (us):", max(last) if last else None)
