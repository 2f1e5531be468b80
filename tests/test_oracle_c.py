"""CPU: the plain-C restatement of the integer part (oracle/int_path.c) against the torch oracle."""
import ctypes
import os
import subprocess

import pytest
import torch

from oracle.ref_cpu import topk_keep_index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_int.so"))
    lib.orc_topk_keep.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.orc_compact_map.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
    lib.orc_cache_advance.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int]
    lib.orc_get_chunk.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("n,k", [(576, 115), (36, 7), (64, 64), (65, 1), (10, 0)])
def test_topk_matches_torch_oracle(clib, n, k):
    g = torch.Generator().manual_seed(n * 1000 + k)
    s = ((torch.randn(4, n, generator=g) * 3).round() / 3).to(torch.bfloat16).float().contiguous()  # heavy ties
    ref = topk_keep_index(s, k, "stable")
    for b in range(4):
        keep = torch.empty(max(k, 1), dtype=torch.int64)
        assert clib.orc_topk_keep(s[b].data_ptr(), n, k, keep.data_ptr()) == 0
        assert keep[:k].tolist() == ref[b].tolist()


def test_compact_map_and_bookkeeping(clib):
    n_in, s, n_img, k = 48, 5, 36, 7
    keep = torch.tensor([1, 12, 14, 15, 22, 25, 26], dtype=torch.int64)
    src = torch.empty(n_in, dtype=torch.int32)
    pos = torch.empty(n_in, dtype=torch.int32)
    n_out = clib.orc_compact_map(n_in, s, n_img, k, keep.data_ptr(), src.data_ptr(), pos.data_ptr())
    ref = torch.cat([torch.arange(0, s), keep + s, torch.arange(s + n_img, n_in)])  # DML:1963-1983
    assert n_out == n_in - (n_img - k) and pos[:n_out].tolist() == ref.tolist() == src[:n_out].tolist()
    lf, ls = torch.tensor([48, 50]), torch.tensor([19, 21])
    dec = torch.tensor([1, 0], dtype=torch.int32)
    clib.orc_cache_advance(lf.data_ptr(), ls.data_ptr(), dec.data_ptr(), 2)
    assert lf.tolist() == [49, 51] and ls.tolist() == [20, 21]
    b, e = ctypes.c_int64(), ctypes.c_int64()
    from dynamic_llava_amd.dist import get_chunk

    for n, c in [(10, 4), (7, 2), (1, 2), (0, 3)]:
        for k_ in range(c):
            clib.orc_get_chunk(n, c, k_, ctypes.byref(b), ctypes.byref(e))
            assert list(range(n))[b.value : e.value] == list(get_chunk(list(range(n)), c, k_))
