"""Full-size CLIP ViT-L/14-336 tower (random weights, bf16) replayed from a hipGraph at 1 / 2 / 4 images: every projection on dl_linear_tiles
(`tiles_max_batch` raised) against the library-GEMM path.  Decides CLIPVisionTower.tiles_max_batch (modules.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.config import DynamicLlavaConfig
from dynamic_llava_amd.modules import CLIPVisionTower
from oracle import fixtures as fx

cfg = DynamicLlavaConfig.from_namespace(fx.llava7b_config(num_hidden_layers=1))
torch.manual_seed(5)
t = CLIPVisionTower(cfg).to("cuda", torch.bfloat16).pack()


def timed(x, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            t(x)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            t(x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


for B in (1, 2, 3, 4, 8, 32):
    x = torch.randn(B, 3, 336, 336, device="cuda", dtype=torch.bfloat16)
    t.tiles_max_batch = 0
    lib = timed(x)
    t.tiles_max_batch = 8
    til = timed(x)
    print(f"images {B}: library path {lib:.3f} ms, dl_linear_tiles path {til:.3f} ms  ({til / lib:.3f})")
