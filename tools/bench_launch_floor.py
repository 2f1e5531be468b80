"""Launch floor of a dependent kernel chain inside a hipGraph: empty kernels of several grid sizes.
    python tools/bench_launch_floor.py
"""
import os
import sys

import torch

import ctypes
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from dynamic_llava_amd import hip_ops as ops  # noqa: E402


def _probe_lib():
    """The empty kernel lives in tools/ (diagnostics are not part of the shipped C ABI): built on demand with hipcc."""
    so, src = os.path.join(HERE, "_launch_probe.so"), os.path.join(HERE, "launch_probe.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.launch_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


_LIB = None


def launch_probe(grid, block):
    global _LIB
    _LIB = _LIB or _probe_lib()
    assert _LIB.launch_probe(int(grid), int(block), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0


def chain_time_us(grid, block, n=400, reps=20):
    def fn():
        for _ in range(n):
            launch_probe(grid, block)

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
