import sys, os; sys.path.insert(0, "/root/repo")
import torch
from dynamic_llava_amd import hip_ops as ops
def timed(fn, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3
dt = torch.bfloat16
for M in (1360, 5048, 631):
    gu = torch.randn(M, 22016, device="cuda", dtype=dt); out = torch.empty(M, 11008, device="cuda", dtype=dt)
    print(f"silu_mul M={M}: {timed(lambda: ops.silu_mul(gu, out=out)):.2f} us  (bytes at 5 TB/s: {M*11008*6/5e6:.1f} us)  DL_EXACT_ACT={os.environ.get('DL_EXACT_ACT','0')}")
for B in (32, 8, 1):
    x = torch.randn(577 * B, 4096, device="cuda", dtype=dt); o = torch.empty_like(x)
    print(f"quick_gelu {B} images: {timed(lambda: ops.quick_gelu(x, out=o)):.2f} us  (bytes at 5 TB/s: {577*B*4096*4/5e6:.1f} us)  DL_EXACT_ACT={os.environ.get('DL_EXACT_ACT','0')}")
