"""Launch floor of a dependent kernel chain inside a hipGraph: empty kernels of several grid sizes.
    python tools/bench_launch_floor.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamic_llava_amd import hip_ops as ops  # noqa: E402


def chain_time_us(grid, block, n=400, reps=20):
    def fn():
        for _ in range(n):
            ops.launch_probe(grid, block)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


if __name__ == "__main__":
    ops.require_gpu()
    for grid, block in [(1, 64), (256, 64), (256, 256), (512, 256), (1024, 256), (512, 1024), (4096, 256)]:
        print(f"empty kernel grid={grid:5d} block={block:4d}: {chain_time_us(grid, block):6.2f} us per dependent launch (in-graph)")
