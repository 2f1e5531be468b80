"""ctypes binding of libdynllava_hip.so (C ABI in include/dynllava.h) for torch tensors.

PyTorch is plumbing here: it owns device memory and the stream; every op below enqueues hand-written
HIP kernels on `torch.cuda.current_stream()` through the C ABI with raw `data_ptr()`s.  There is NO
fallback: if the library is missing or the device is not gfx950 the ops raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdynllava_hip.so")

DL_F32, DL_F16, DL_BF16 = 0, 1, 2
ABI_VERSION = 4  # include/dynllava.h DL_ABI_VERSION: a library built from another header is refused at load
EPI_GELU, EPI_RESIDUAL = 1, 2
_DTYPES = {torch.float32: DL_F32, torch.float16: DL_F16, torch.bfloat16: DL_BF16}


class HipOpsError(RuntimeError):
    pass


class VpBlock(Structure):
    _fields_ = [(n, c_void_p) for n in ("norm1_w", "norm1_b", "qkv_w", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VpWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_w", "ln_b", "down_w", "down_b", "out0_w", "out0_b", "out2_w", "out2_b", "out4_w", "out4_b")] + [
        ("num_layers", c_int),
        ("blocks", VpBlock * 4),
    ]


class TpWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_w", "ln_b", "l1_w", "l1_b", "l3_w", "l3_b", "l5_w", "l5_b", "l7_w", "l7_b")]


# symbol -> (restype, argtypes); must list every function declared in include/dynllava.h
SIGNATURES = {
    "dl_version": (c_int, []),
    "dl_last_error": (c_char_p, []),
    "dl_device_check": (c_int, []),
    "dl_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_silu_mul": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "dl_rope_kv_write": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "dl_rope_kv_write_parts": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p],
    ),
    "dl_attn_prefill": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "dl_attn_prefill_cached": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "dl_attn_decode_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "dl_attn_decode": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "dl_attn_decode_rope": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "dl_attn_decode_rope_parts": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "dl_topk_select": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_compact_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "dl_linear": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_vision_predictor_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dl_vision_predictor": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(VpWeights), c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    ),
    "dl_text_predictor_decide": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, POINTER(TpWeights), c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dl_text_predictor_workspace_bytes": (c_int64, [c_int, c_int]),
    "dl_gemv_max_batch": (c_int, [c_int, c_int]),
    "dl_gemv": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "dl_linear_splitk": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_gemv_qkv_attn_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "dl_gemv_gu_tp_workspace_bytes": (c_int64, [c_int]),
    "dl_gemv_gu_tp": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, POINTER(TpWeights), c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "dl_gemv_qkv_attn": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_attn_policy_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "dl_attn_policy_fwd": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "dl_attn_policy_bwd": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_int64, c_int64,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "dl_gumbel_hard_keep_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_int, c_void_p]),
    "dl_gumbel_hard_keep_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_int, c_void_p]),
    "dl_kv_pack_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_prompt_layout": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dl_compact_rows_by_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dl_gemm_smallm_max_m": (c_int, []),
    "dl_gemm_smallm_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "dl_gemm_smallm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_gemm_smallm_slices": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "dl_add_rmsnorm_parts": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_silu_mul_parts": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "dl_add_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_quick_gelu": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "dl_rmsnorm_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_add_rmsnorm_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_add_rmsnorm_parts_packed": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "dl_packed_weight_bytes": (c_int64, [c_int, c_int, c_int]),
    "dl_pack_weight_tiles": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_packed_x_bytes": (c_int64, [c_int, c_int]),
    "dl_pack_x_tiles": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dl_linear_packed_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "dl_linear_packed": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dl_linear_packed_stamped": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dl_tiles_x_bytes": (c_int64, [c_int, c_int]),
    "dl_layernorm_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_void_p]),
    "dl_add_layernorm_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_void_p]),
    "dl_add_layernorm_parts": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_void_p]),
    "dl_pack_x_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dl_linear_tiles": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dl_linear_tiles_stamped": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "dl_decode_advance": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    ),
}

_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen + bind every symbol of the ABI.  Raises HipOpsError when the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise HipOpsError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback for the hot path."
        )
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.dl_version() != ABI_VERSION:  # argument lists changed between versions: a stale .so would read pointers as integers
        raise HipOpsError(f"{path} reports ABI version {lib.dl_version()}, this binding is written for {ABI_VERSION}: rebuild (`__graft_entry__.build()`)")
    _lib = lib
    return lib


def lib():
    return load_library()


def require_gpu():
    if not torch.cuda.is_available():
        raise HipOpsError("no HIP device visible: dynamic_llava_amd's hot path only runs on an MI355X (gfx950)")
    if lib().dl_device_check() != 0:
        raise HipOpsError(lib().dl_last_error().decode())


def _check(rc: int, what: str):
    if rc != 0:
        raise HipOpsError(f"{what} failed (rc={rc}): {lib().dl_last_error().decode()}")


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPES[dt]
    except KeyError:
        raise HipOpsError(f"unsupported dtype {dt}")


def _dev(t, *more):
    if not t.is_cuda:
        raise HipOpsError("tensor is not on the GPU: the HIP hot path has no CPU fallback")
    for m in more:
        if m is not None and not m.is_cuda:
            raise HipOpsError("tensor is not on the GPU: the HIP hot path has no CPU fallback")


# ------------------------------------------------------------------------------------------------
# thin op wrappers (shapes documented in include/dynllava.h)
# ------------------------------------------------------------------------------------------------
def _packed_out(rows, H, like, out):
    """Flat buffer for a [rows, H] activation in dl_linear_packed's fragment order (pack_x_tiles / the *_packed norm launches)."""
    need = int(lib().dl_packed_x_bytes(rows, H))
    if need < 0:
        raise HipOpsError(f"packed activations: unsupported shape [{rows},{H}] (rows <= 256, H % 64 == 0)")
    if out is None:
        out = torch.empty(need // like.element_size(), dtype=like.dtype, device=like.device)
    assert out.is_contiguous() and out.dtype == like.dtype and out.numel() * out.element_size() >= need
    return out


def rmsnorm(x, w, eps, out=None, packed=False):
    """packed: the result goes out in dl_linear_packed's activation order (a flat tensor; feed it with x_packed_mk=(rows, H))."""
    _dev(x, w)
    assert x.is_contiguous() and x.dtype == w.dtype
    H = x.shape[-1]
    if packed:
        out = _packed_out(x.numel() // H, H, x, out)
        _check(lib().dl_rmsnorm_packed(_p(x), _p(w), _p(out), x.numel() // H, H, eps, dtype_code(x.dtype), _stream()), "dl_rmsnorm_packed")
        return out
    out = torch.empty_like(x) if out is None else out
    _check(lib().dl_rmsnorm(_p(x), _p(w), _p(out), x.numel() // H, H, eps, dtype_code(x.dtype), _stream()), "dl_rmsnorm")
    return out


def add_rmsnorm(h, delta, w, eps, out=None, packed=False):
    """h += delta (in place, rounded); returns rmsnorm(h) (or None when w is None).  packed: see rmsnorm."""
    _dev(h, delta, w)
    assert h.is_contiguous() and delta.is_contiguous() and h.shape == delta.shape and h.dtype == delta.dtype
    H = h.shape[-1]
    if packed:
        assert w is not None
        out = _packed_out(h.numel() // H, H, h, out)
        _check(lib().dl_add_rmsnorm_packed(_p(h), _p(delta), _p(w), _p(out), h.numel() // H, H, eps, dtype_code(h.dtype), _stream()), "dl_add_rmsnorm_packed")
        return out
    if w is not None and out is None:
        out = torch.empty_like(h)
    _check(lib().dl_add_rmsnorm(_p(h), _p(delta), _p(w), _p(out) if w is not None else None, h.numel() // H, H, eps, dtype_code(h.dtype), _stream()), "dl_add_rmsnorm")
    return out if w is not None else None


def silu_mul(gate_up, out=None):
    _dev(gate_up)
    assert gate_up.is_contiguous()
    I = gate_up.shape[-1] // 2
    rows = gate_up.numel() // (2 * I)
    out = torch.empty(*gate_up.shape[:-1], I, dtype=gate_up.dtype, device=gate_up.device) if out is None else out
    _check(lib().dl_silu_mul(_p(gate_up), _p(out), rows, I, dtype_code(gate_up.dtype), _stream()), "dl_silu_mul")
    return out


def rope_kv_write(qkv, cos, sin, cu_seqlens, pos, pos_base, kv_base, k_slab, v_slab, n_heads, n_kv_heads, head_dim, parts=None):
    """qkv [total, (nH+2nKV)*d] rotated in place; k_slab/v_slab [B, nKV, T_cap, d].  parts (fp32 [n_parts, total, (nH+2nKV)*d], dl_linear_packed's
    partial sums): qkv is written from their rounded sum instead (q, k rotated, v as is)."""
    _dev(qkv, cos, sin, cu_seqlens, k_slab, v_slab)
    assert qkv.is_contiguous() and cos.is_contiguous() and sin.is_contiguous()
    assert k_slab.stride(3) == 1 and k_slab.stride(2) == head_dim and k_slab.stride() == v_slab.stride()
    B = cu_seqlens.numel() - 1
    total = qkv.shape[0]
    if parts is not None:
        _dev(parts)
        assert parts.dtype == torch.float32 and parts.is_contiguous() and parts.dim() == 3 and tuple(parts.shape[1:]) == tuple(qkv.shape)
        _check(
            lib().dl_rope_kv_write_parts(
                _p(qkv), _p(parts), parts.shape[0], _p(cos), _p(sin), cos.shape[0], _p(cu_seqlens), _p(pos), _p(pos_base), _p(kv_base), _p(k_slab), _p(v_slab),
                k_slab.stride(0), k_slab.stride(1), k_slab.shape[2], B, total, n_heads, n_kv_heads, head_dim, dtype_code(qkv.dtype), _stream(),
            ),
            "dl_rope_kv_write_parts",
        )
        return
    _check(
        lib().dl_rope_kv_write(
            _p(qkv), _p(cos), _p(sin), cos.shape[0], _p(cu_seqlens), _p(pos), _p(pos_base), _p(kv_base), _p(k_slab), _p(v_slab),
            k_slab.stride(0), k_slab.stride(1), k_slab.shape[2], B, total, n_heads, n_kv_heads, head_dim, dtype_code(qkv.dtype), _stream(),
        ),
        "dl_rope_kv_write",
    )


def attn_prefill(q, k, v, out, cu_seqlens, max_seqlen, n_heads, n_kv_heads, head_dim, causal):
    """q/k/v: 2-D strided views [total, *] whose row stride is .stride(0); out [total, n_heads*head_dim]."""
    _dev(q, k, v, out, cu_seqlens)
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1 and out.stride(1) == 1
    assert k.stride(0) == v.stride(0)
    B = cu_seqlens.numel() - 1
    _check(
        lib().dl_attn_prefill(
            _p(q), _p(k), _p(v), q.stride(0), k.stride(0), _p(out), out.stride(0), _p(cu_seqlens), B, int(max_seqlen), n_heads, n_kv_heads,
            head_dim, 1 if causal else 0, dtype_code(q.dtype), _stream(),
        ),
        "dl_attn_prefill",
    )
    return out


def attn_prefill_cached(q, k_slab, v_slab, kv_len, out, cu_seqlens, max_seqlen, max_kv_len, n_heads, n_kv_heads, head_dim):
    """Causal attention of a packed query chunk against the KV slab (chunk keys already appended at [kv_len[b], ...))."""
    _dev(q, k_slab, v_slab, kv_len, out, cu_seqlens)
    assert q.stride(1) == 1 and out.stride(1) == 1 and kv_len.dtype == torch.int32
    assert k_slab.stride(3) == 1 and k_slab.stride(2) == head_dim and k_slab.stride() == v_slab.stride()
    B = cu_seqlens.numel() - 1
    _check(
        lib().dl_attn_prefill_cached(
            _p(q), q.stride(0), _p(k_slab), _p(v_slab), k_slab.stride(0), k_slab.stride(1), _p(kv_len), _p(out), out.stride(0), _p(cu_seqlens), B,
            int(max_seqlen), int(max_kv_len), n_heads, n_kv_heads, head_dim, dtype_code(q.dtype), _stream(),
        ),
        "dl_attn_prefill_cached",
    )
    return out


def attn_decode_workspace(B, n_heads, head_dim, n_splits, device):
    nbytes = lib().dl_attn_decode_workspace_bytes(B, n_heads, head_dim, n_splits)
    return torch.zeros(max(int(nbytes) // 4, 1), dtype=torch.float32, device=device)


def attn_decode(q, k_slab, v_slab, kv_len, extra, out, workspace, n_splits, n_heads, n_kv_heads, head_dim):
    """q: [B, *] strided view (row stride q.stride(0)); k_slab/v_slab [B, nKV, T_cap, d]; kv_len int32 [B]."""
    _dev(q, k_slab, v_slab, kv_len, out)
    assert q.stride(1) == 1 and out.stride(1) == 1 and kv_len.dtype == torch.int32
    B = q.shape[0]
    _check(
        lib().dl_attn_decode(
            _p(q), q.stride(0), _p(k_slab), _p(v_slab), k_slab.stride(0), k_slab.stride(1), _p(kv_len), int(extra), _p(out), out.stride(0),
            _p(workspace), int(n_splits), B, n_heads, n_kv_heads, head_dim, dtype_code(q.dtype), _stream(),
        ),
        "dl_attn_decode",
    )
    return out


def attn_decode_rope(qkv, cos, sin, pos_base, kv_len, k_slab, v_slab, out, workspace, n_splits, n_heads, n_kv_heads, head_dim, keys_in_flight=64, chunk_keys=0, call_tag=-1):
    """Fused RoPE + KV append + ragged decode attention.  qkv [B, (nH+2nKV)*d] un-rotated (not modified)."""
    _dev(qkv, cos, sin, pos_base, kv_len, k_slab, v_slab, out)
    assert qkv.stride(1) == 1 and out.stride(1) == 1 and kv_len.dtype == torch.int32 and pos_base.dtype == torch.int32
    assert k_slab.stride(3) == 1 and k_slab.stride(2) == head_dim and k_slab.stride() == v_slab.stride()
    B = qkv.shape[0]
    _check(
        lib().dl_attn_decode_rope(
            _p(qkv), qkv.stride(0), _p(cos), _p(sin), cos.shape[0], _p(pos_base), _p(kv_len), _p(k_slab), _p(v_slab), k_slab.stride(0), k_slab.stride(1),
            k_slab.shape[2], _p(out), out.stride(0), _p(workspace), int(n_splits), int(keys_in_flight), int(chunk_keys), int(call_tag), B, n_heads, n_kv_heads, head_dim, dtype_code(qkv.dtype), _stream(),
        ),
        "dl_attn_decode_rope",
    )
    return out


def attn_decode_rope_parts(qkv_parts, cos, sin, pos_base, kv_len, k_slab, v_slab, out, workspace, n_splits, n_heads, n_kv_heads, head_dim, chunk_keys=0, call_tag=-1):
    """attn_decode_rope on the projection's fp32 partial sums qkv_parts [n_parts, B, (nH+2nKV)*d] (linear_packed(..., epilogue=LP_PARTS)): added in range order and
    rounded to out.dtype inside the launch."""
    _dev(qkv_parts, cos, sin, pos_base, kv_len, k_slab, v_slab, out)
    assert qkv_parts.dtype == torch.float32 and qkv_parts.dim() == 3 and qkv_parts.stride(2) == 1 and out.stride(1) == 1
    assert kv_len.dtype == torch.int32 and pos_base.dtype == torch.int32 and cos.dtype == out.dtype == k_slab.dtype
    assert k_slab.stride(3) == 1 and k_slab.stride(2) == head_dim and k_slab.stride() == v_slab.stride()
    n_parts, B = qkv_parts.shape[0], qkv_parts.shape[1]
    _check(
        lib().dl_attn_decode_rope_parts(
            _p(qkv_parts), n_parts, qkv_parts.stride(0), qkv_parts.stride(1), _p(cos), _p(sin), cos.shape[0], _p(pos_base), _p(kv_len), _p(k_slab), _p(v_slab), k_slab.stride(0),
            k_slab.stride(1), k_slab.shape[2], _p(out), out.stride(0), _p(workspace), int(n_splits), int(chunk_keys), int(call_tag), B, n_heads, n_kv_heads, head_dim, dtype_code(out.dtype), _stream(),
        ),
        "dl_attn_decode_rope_parts",
    )
    return out


def topk_select(score, k):
    _dev(score)
    assert score.is_contiguous() and score.dim() == 2
    B, n = score.shape
    keep = torch.empty((B, k), dtype=torch.int64, device=score.device)
    _check(lib().dl_topk_select(_p(score), _p(keep), B, n, k, dtype_code(score.dtype), _stream()), "dl_topk_select")
    return keep


def compact_tokens(h_in, keep_idx, cu_in, cu_out, img_start, n_img, k, total_out, norm_w=None, eps=0.0):
    """-> (h_out, pos) or, with norm_w, (h_out, pos, rmsnorm(h_out) * norm_w) from the same launch."""
    _dev(h_in, keep_idx, cu_in, cu_out, img_start, norm_w)
    assert h_in.is_contiguous()
    H = h_in.shape[-1]
    B = cu_in.numel() - 1
    h_out = torch.empty((total_out, H), dtype=h_in.dtype, device=h_in.device)
    pos = torch.empty((total_out,), dtype=torch.int32, device=h_in.device)
    x_out = torch.empty_like(h_out) if norm_w is not None else None
    _check(
        lib().dl_compact_tokens(_p(h_in), _p(h_out), _p(keep_idx), _p(cu_in), _p(cu_out), _p(img_start), _p(pos), B, n_img, k, total_out, H, _p(norm_w), float(eps), _p(x_out),
                                dtype_code(h_in.dtype), _stream()),
        "dl_compact_tokens",
    )
    return (h_out, pos) if norm_w is None else (h_out, pos, x_out)


def compact_rows_by_mask(h_in, pos_in, decision, span0, n_span):
    """One packed sequence: rows [span0, span0 + n_span) kept where decision != 0 -> (h_out [total,H] zero past the kept rows, pos int32[total],
    cu int32[2] = {0, kept}, counts int64[2] = {kept, kept - 1}); nothing is read back."""
    _dev(h_in, pos_in, decision)
    assert h_in.is_contiguous() and decision.dtype == torch.int32 and (pos_in is None or pos_in.dtype == torch.int32)
    total, H = h_in.shape
    dev = h_in.device
    h_out = torch.zeros_like(h_in)
    pos = torch.zeros(total, dtype=torch.int32, device=dev)
    cu = torch.empty(2, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    _check(lib().dl_compact_rows_by_mask(_p(h_in), _p(pos_in), _p(decision), int(span0), int(n_span), total, H, _p(h_out), _p(pos), _p(cu), _p(counts), dtype_code(h_in.dtype), _stream()),
           "dl_compact_rows_by_mask")
    return h_out, pos, cu, counts


def linear(a, w, bias=None, flags=0, residual=None, out=None):
    _dev(a, w, bias, residual)
    assert a.dim() == 2 and a.stride(1) == 1 and w.is_contiguous()
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=a.dtype, device=a.device) if out is None else out
    _check(
        lib().dl_linear(_p(a), a.stride(0), _p(w), _p(bias), _p(out), out.stride(0), _p(residual), residual.stride(0) if residual is not None else 0, M, N, K, flags, dtype_code(a.dtype), _stream()),
        "dl_linear",
    )
    return out


def layernorm(x, w, b, eps=1e-5, row_index=None, rows=None):
    _dev(x, w, b, row_index)
    assert x.is_contiguous()
    H = x.shape[-1]
    rows = (x.numel() // H) if rows is None else rows
    out = torch.empty((rows, H), dtype=x.dtype, device=x.device)
    _check(lib().dl_layernorm(_p(x), _p(row_index), _p(w), _p(b), _p(out), rows, H, eps, dtype_code(x.dtype), _stream()), "dl_layernorm")
    return out


def add_layernorm(h, delta, w=None, b=None, eps=1e-5, out=None):
    """h += delta in place (rounded to the dtype); returns LN(h) * w + b (or None when w is None: residual add only)."""
    _dev(h, delta, w, b, out)
    assert h.is_contiguous() and delta.is_contiguous() and h.shape == delta.shape
    H = h.shape[-1]
    rows = h.numel() // H
    if w is not None and out is None:
        out = torch.empty_like(h)
    _check(lib().dl_add_layernorm(_p(h), _p(delta), _p(w), _p(b), _p(out) if w is not None else None, rows, H, eps, dtype_code(h.dtype), _stream()), "dl_add_layernorm")
    return out if w is not None else None


def _ln_rows_out(rows, H, like, out, packed):
    if packed:
        n = tiles_x_numel(rows, H)
        if out is None:
            out = torch.empty(n, dtype=like.dtype, device=like.device)
        assert out.is_contiguous() and out.dtype == like.dtype and out.numel() >= n
    elif out is None:
        out = torch.empty((rows, H), dtype=like.dtype, device=like.device)
    return out


def layernorm_rows(x, w, b, eps=1e-5, out=None, packed=False):
    """LN(x) * w + b, a wave per row (the CLIP tower's launches around linear_tiles); packed: fragment-order output for linear_tiles(x_packed_mk=...)."""
    _dev(x, w, b, out)
    assert x.is_contiguous()
    H = x.shape[-1]
    rows = x.numel() // H
    out = _ln_rows_out(rows, H, x, out, packed)
    _check(lib().dl_layernorm_rows(_p(x), _p(w), _p(b), _p(out), rows, H, eps, int(bool(packed)), dtype_code(x.dtype), _stream()), "dl_layernorm_rows")
    return out


def add_layernorm_rows(h, delta, w=None, b=None, eps=1e-5, out=None, packed=False):
    """h += delta in place (rounded); returns LN(h) * w + b (None when w is None), a wave per row; packed: fragment-order output."""
    _dev(h, delta, w, b, out)
    assert h.is_contiguous() and delta.is_contiguous() and h.shape == delta.shape
    H = h.shape[-1]
    rows = h.numel() // H
    if w is not None:
        out = _ln_rows_out(rows, H, h, out, packed)
    _check(lib().dl_add_layernorm_rows(_p(h), _p(delta), _p(w), _p(b), _p(out) if w is not None else None, rows, H, eps, int(bool(packed)), dtype_code(h.dtype), _stream()),
           "dl_add_layernorm_rows")
    return out if w is not None else None


def add_layernorm_parts(h, parts, bias=None, w=None, b=None, eps=1e-5, out=None, packed=False):
    """h += cast(sum of the fp32 k-range partial sums `parts` [n, rows, H] (linear_tiles LT_PARTS) + bias) in place; returns LN(h) * w + b (None when w is None)."""
    _dev(h, parts, bias, w, b, out)
    assert h.is_contiguous() and parts.is_contiguous() and parts.dtype == torch.float32 and parts.dim() == 3 and tuple(parts.shape[1:]) == tuple(h.shape)
    H = h.shape[-1]
    rows = h.numel() // H
    if w is not None:
        out = _ln_rows_out(rows, H, h, out, packed)
    _check(lib().dl_add_layernorm_parts(_p(h), _p(parts), parts.shape[0], _p(bias), _p(w), _p(b), _p(out) if w is not None else None, rows, H, eps, int(bool(packed)),
                                        dtype_code(h.dtype), _stream()), "dl_add_layernorm_parts")
    return out if w is not None else None


def quick_gelu(x, out=None):
    _dev(x, out)
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    _check(lib().dl_quick_gelu(_p(x), _p(out), x.numel(), dtype_code(x.dtype), _stream()), "dl_quick_gelu")
    return out


def vision_predictor(hidden, cu_seqlens, img_start, n_img, weights: VpWeights, d_model, nhead, dim_ff, workspace=None):
    """hidden packed [total,H] -> (logits [B,n_img,2], score [B,n_img]) in the model dtype."""
    _dev(hidden, cu_seqlens, img_start)
    assert hidden.is_contiguous()
    B = cu_seqlens.numel() - 1
    H = hidden.shape[-1]
    dc = dtype_code(hidden.dtype)
    if workspace is None:
        nbytes = lib().dl_vision_predictor_workspace_bytes(B, n_img, H, d_model, dim_ff, dc)
        workspace = torch.empty(int(nbytes), dtype=torch.uint8, device=hidden.device)
    logits = torch.empty((B, n_img, 2), dtype=hidden.dtype, device=hidden.device)
    score = torch.empty((B, n_img), dtype=hidden.dtype, device=hidden.device)
    _check(
        lib().dl_vision_predictor(_p(hidden), _p(cu_seqlens), _p(img_start), B, n_img, H, d_model, nhead, dim_ff, ctypes.byref(weights), _p(workspace), _p(logits), _p(score), dc, _stream()),
        "dl_vision_predictor",
    )
    return logits, score


def text_predictor_workspace(B, d_model, device):
    return torch.empty(max(int(lib().dl_text_predictor_workspace_bytes(int(B), int(d_model))) // 4, 1), dtype=torch.float32, device=device)


def text_predictor_decide(x, weights: TpWeights, d_model, workspace, logits_out, decision):
    """x [B,H] (row stride x.stride(0)) -> decision int32 [B] (1 = keep this token's KV), logits fp32 [B,2]."""
    _dev(x, workspace, decision)
    assert x.stride(1) == 1 and decision.dtype == torch.int32
    B, H = x.shape
    assert workspace.dtype == torch.float32 and workspace.numel() * 4 >= lib().dl_text_predictor_workspace_bytes(B, d_model), "workspace too small (text_predictor_workspace)"
    _check(
        lib().dl_text_predictor_decide(_p(x), x.stride(0), B, H, d_model, ctypes.byref(weights), _p(workspace), _p(logits_out), _p(decision), dtype_code(x.dtype), _stream()),
        "dl_text_predictor_decide",
    )
    return decision


GEMV_PLAIN, GEMV_ADDNORM, GEMV_SILUMUL, GEMV_OUT_SILU_PAIR = 0, 1, 2, 16


def gemv_max_batch(K, dtype):
    return int(lib().dl_gemv_max_batch(int(K), dtype_code(dtype)))


def gemv(w, y, x=None, mode=GEMV_PLAIN, h_in=None, h_out=None, delta=None, norm_w=None, eps=0.0, grid_cap=0):
    """y[b,:] = W @ prologue(x)[b,:] (see include/dynllava.h).  w [N,K]; y [B,N] (row stride y.stride(0))."""
    _dev(w, y, x, h_in, h_out, delta, norm_w)
    assert w.is_contiguous() and y.stride(1) == 1
    N, K = w.shape
    B = y.shape[0]
    if (mode & 3) == GEMV_ADDNORM:
        assert h_in.is_contiguous() and h_in.shape == (B, K) and (delta is None or (delta.is_contiguous() and h_out.is_contiguous()))
        xs = 0
    else:
        assert x.stride(1) == 1 and x.shape[0] == B
        xs = x.stride(0)
    _check(
        lib().dl_gemv(mode, _p(w), N, K, _p(x), xs, _p(h_in), _p(h_out), _p(delta), _p(norm_w), eps, _p(y), y.stride(0), B, dtype_code(w.dtype), int(grid_cap), _stream()),
        "dl_gemv",
    )
    return y


def gemv_qkv_attn_workspace(n_heads, n_kv_heads, head_dim, device):
    """Granule buffer of gemv_qkv_attn (zeroed: no tag is 0); zero it again at the start of every request."""
    n = int(lib().dl_gemv_qkv_attn_workspace_bytes(int(n_heads), int(n_kv_heads), int(head_dim)))
    return torch.zeros(n // 8, dtype=torch.int64, device=device)


def gemv_qkv_attn(w, qkv, h_in, h_out, delta, norm_w, eps, cos, sin, pos_base, kv_len, k_slab, v_slab, out, granules, call_tag, n_heads, n_kv_heads, head_dim,
                  err=None, grid_cap=0, n_splits=1):
    """One launch = gemv(w, qkv, mode=GEMV_ADDNORM, ...) + attn_decode_rope(qkv, ..., n_splits=1) for ONE row (see include/dynllava.h).
    n_splits: attention workgroups per head (1..4; > 1: 128 slab keys each, partials merged by the head's first workgroup)."""
    _dev(w, qkv, h_in, h_out, delta, norm_w, cos, sin, pos_base, kv_len, k_slab, v_slab, out, granules, err)
    N, K = w.shape
    assert w.is_contiguous() and qkv.shape == (1, N) and qkv.is_contiguous() and h_in.is_contiguous() and h_in.shape == (1, K)
    assert delta is None or (delta.is_contiguous() and h_out.is_contiguous())
    assert out.shape == (1, n_heads * head_dim) and out.is_contiguous() and N == (n_heads + 2 * n_kv_heads) * head_dim
    assert pos_base.dtype == torch.int32 and kv_len.dtype == torch.int32 and granules.numel() * granules.element_size() >= lib().dl_gemv_qkv_attn_workspace_bytes(int(n_heads), int(n_kv_heads), int(head_dim))
    assert k_slab.stride(3) == 1 and k_slab.stride(2) == head_dim and k_slab.stride() == v_slab.stride()
    sb, sh = k_slab.stride(0), k_slab.stride(1)
    _check(
        lib().dl_gemv_qkv_attn(
            _p(w), K, _p(h_in), _p(h_out), _p(delta), _p(norm_w), eps, _p(qkv), _p(cos), _p(sin), cos.shape[0], _p(pos_base), _p(kv_len), _p(k_slab), _p(v_slab), sb, sh,
            k_slab.shape[2], _p(out), _p(granules), int(call_tag), _p(err), int(n_splits), int(n_heads), int(n_kv_heads), int(head_dim), dtype_code(w.dtype), int(grid_cap), _stream(),
        ),
        "dl_gemv_qkv_attn",
    )
    return out


def gemv_gu_tp_workspace(d_model, device):
    """Granule buffer of gemv_gu_tp (zeroed: no tag is 0); zero it again at the start of every request."""
    return torch.zeros(int(lib().dl_gemv_gu_tp_workspace_bytes(int(d_model))) // 8, dtype=torch.int64, device=device)


def gemv_gu_tp(w, y, h_in, h_out, delta, norm_w, eps, tp_weights: TpWeights, d_model, tp_workspace, logits_out, decision, pos_base, granules, call_tag, err=None, grid_cap=0):
    """One launch = gemv(w, y, mode=GEMV_ADDNORM | GEMV_OUT_SILU_PAIR, ...) + text_predictor_decide(h_in, ...) for ONE row (see include/dynllava.h)."""
    _dev(w, y, h_in, h_out, delta, norm_w, tp_workspace, logits_out, decision, pos_base, granules, err)
    N, K = w.shape
    assert w.is_contiguous() and y.shape == (1, N // 2) and y.is_contiguous() and h_in.is_contiguous() and h_in.shape == (1, K)
    assert delta is None or (delta.is_contiguous() and h_out.is_contiguous())
    assert decision.dtype == torch.int32 and pos_base.dtype == torch.int32 and tp_workspace.dtype == torch.float32
    assert tp_workspace.numel() * 4 >= lib().dl_text_predictor_workspace_bytes(1, d_model) and granules.numel() * granules.element_size() >= lib().dl_gemv_gu_tp_workspace_bytes(d_model)
    _check(
        lib().dl_gemv_gu_tp(_p(w), N, K, _p(h_in), _p(h_out), _p(delta), _p(norm_w), eps, _p(y), ctypes.byref(tp_weights), int(d_model), _p(tp_workspace), _p(logits_out),
                            _p(decision), _p(pos_base), _p(granules), int(call_tag), _p(err), dtype_code(w.dtype), int(grid_cap), _stream()),
        "dl_gemv_gu_tp",
    )
    return y


def decode_advance(logits, next_ids, out_ids=None, step=None, finished=None, eos_id=-1, pad_id=0, kv_len_full=None, kv_len_sparse=None, decision=None, min_new_tokens=0):
    """eos_id: -1 (none), one id, or a sequence of up to three ids (the EOS set)."""
    _dev(logits, next_ids)
    eos = [int(e) for e in eos_id] if isinstance(eos_id, (list, tuple)) else [int(eos_id)]
    if len(eos) > 3:
        raise HipOpsError("dl_decode_advance compares at most three eos ids on the device")
    eos = (eos + [-1, -1, -1])[:3]
    assert logits.dim() == 2 and logits.stride(1) == 1 and next_ids.dtype == torch.int64
    B, V = logits.shape
    out_cap = out_ids.shape[1] if out_ids is not None else 0
    _check(
        lib().dl_decode_advance(
            _p(logits), dtype_code(logits.dtype), logits.stride(0), V, B, _p(next_ids), _p(out_ids), out_cap, _p(step), _p(finished), eos[0], eos[1], eos[2], int(pad_id),
            _p(kv_len_full), _p(kv_len_sparse), _p(decision), int(min_new_tokens), _stream(),
        ),
        "dl_decode_advance",
    )
    return next_ids


def _strides3(t):
    assert t.dim() == 4 and t.stride(3) == 1
    return (c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def attn_policy_workspace(B, H, L, d, device):
    return torch.empty(int(lib().dl_attn_policy_workspace_floats(B, H, L, d)), device=device, dtype=torch.float32)


def _bias_args(bias, B, L):
    if bias is None:
        return None, 0, 0
    assert bias.dim() == 4 and bias.shape[1] == 1 and bias.shape[2] == L and bias.shape[3] == L and bias.stride(3) == 1 and bias.shape[0] in (1, B)
    return bias, (bias.stride(0) if bias.shape[0] == B else 0), bias.stride(2)


def attn_policy_fwd(q, k, v, out, policy, bias, row_max, row_denom, workspace, causal, scale, eps, n_for_eps):
    """q/k/v/out: [B,H,L,d] views (d contiguous; q, k, v with equal strides); policy fp32 [B,L]; see include/dynllava.h."""
    _dev(q, k, v, out, policy, bias, row_max, row_denom, workspace)
    B, H, L, d = q.shape
    assert k.stride() == q.stride() and v.stride() == q.stride() and policy.dtype == torch.float32 and policy.is_contiguous()
    bias, bsb, bsl = _bias_args(bias, B, L)
    _check(
        lib().dl_attn_policy_fwd(_p(q), _p(k), _p(v), _strides3(q), _p(out), _strides3(out), _p(policy), _p(bias), bsb, bsl, _p(row_max), _p(row_denom), _p(workspace),
                                 B, H, L, d, int(bool(causal)), float(scale), float(eps), int(n_for_eps), dtype_code(q.dtype), _stream()),
        "dl_attn_policy_fwd",
    )
    return out


def attn_policy_bwd(q, k, v, out, d_out, dq, dk, dv, policy, bias, row_max, row_denom, dpolicy_heads, workspace, causal, scale, eps, n_for_eps):
    _dev(q, k, v, out, d_out, dq, dk, dv, policy, bias, row_max, row_denom, dpolicy_heads, workspace)
    B, H, L, d = q.shape
    assert k.stride() == q.stride() and v.stride() == q.stride()
    assert d_out.stride() == out.stride() and dq.stride() == out.stride() and dk.stride() == out.stride() and dv.stride() == out.stride()
    bias, bsb, bsl = _bias_args(bias, B, L)
    _check(
        lib().dl_attn_policy_bwd(_p(q), _p(k), _p(v), _strides3(q), _p(out), _p(d_out), _p(dq), _p(dk), _p(dv), _strides3(out), _p(policy), _p(bias), bsb, bsl,
                                 _p(row_max), _p(row_denom), _p(dpolicy_heads), _p(workspace), B, H, L, d, int(bool(causal)), float(scale), float(eps), int(n_for_eps),
                                 dtype_code(q.dtype), _stream()),
        "dl_attn_policy_bwd",
    )


def gumbel_hard_keep_fwd(log_probs, gumbels, prev, keep, y_soft, tau):
    _dev(log_probs, gumbels, prev, keep, y_soft)
    assert log_probs.is_contiguous() and gumbels.is_contiguous() and prev.is_contiguous() and log_probs.shape[-1] == 2
    n = log_probs.numel() // 2
    _check(lib().dl_gumbel_hard_keep_fwd(_p(log_probs), _p(gumbels), _p(prev), _p(keep), _p(y_soft), n, float(tau), dtype_code(log_probs.dtype), _stream()), "dl_gumbel_hard_keep_fwd")


def gumbel_hard_keep_bwd(d_keep, prev, y_soft, d_log_probs, tau, d_prev=None):
    _dev(d_keep, prev, y_soft, d_log_probs, d_prev)
    assert d_keep.is_contiguous() and prev.is_contiguous() and y_soft.is_contiguous()
    n = y_soft.numel() // 2
    _check(lib().dl_gumbel_hard_keep_bwd(_p(d_keep), _p(prev), _p(y_soft), _p(d_log_probs), _p(d_prev), n, float(tau), dtype_code(y_soft.dtype), _stream()), "dl_gumbel_hard_keep_bwd")


def linear_splitk(a, w, parts, n_slices):
    """parts[s] = a @ w[:, slice s]^T in fp32 (no bias); a [M,K] (row stride a.stride(0)), w [N,K]; parts: fp32, >= n_slices*M*N elements.
    Returns the [n_slices, M, N] view for add_rmsnorm_parts."""
    _dev(a, w, parts)
    assert a.dim() == 2 and a.stride(1) == 1 and w.is_contiguous() and parts.dtype == torch.float32
    M, K = a.shape
    N = w.shape[0]
    assert parts.numel() >= n_slices * M * N
    _check(lib().dl_linear_splitk(_p(a), a.stride(0), _p(w), _p(parts), M, N, K, int(n_slices), dtype_code(a.dtype), _stream()), "dl_linear_splitk")
    return parts[: n_slices * M * N].view(n_slices, M, N)


LP_STORE, LP_SILU_PAIR, LP_RESID, LP_PARTS = 0, 1, 2, 3
LP_Y_PACKED = 16
LP_MAX_ROWS = 256


def linear_packed_ok(M, N, K, dtype):
    """Shapes dl_linear_packed takes."""
    return dtype in (torch.bfloat16, torch.float16) and 0 < M <= LP_MAX_ROWS and N % 16 == 0 and K % 64 == 0


def pack_weight_tiles(w, gate_up_pairs=False, out=None):
    """Copy of w [N,K] (nn.Linear layout) in matrix-core operand order for linear_packed (include/dynllava.h).  gate_up_pairs: w = [gate; up],
    gate / up tiles interleaved for the SiLU * up epilogue."""
    _dev(w, out)
    assert w.dim() == 2 and w.is_contiguous()
    N, K = w.shape
    need = int(lib().dl_packed_weight_bytes(N, K, dtype_code(w.dtype)))
    if need < 0:
        raise HipOpsError(f"dl_pack_weight_tiles: unsupported shape / dtype [{N},{K}] {w.dtype}")
    if out is None:
        out = torch.empty(N * K, dtype=w.dtype, device=w.device)
    assert out.numel() * out.element_size() >= need and out.is_contiguous()
    _check(lib().dl_pack_weight_tiles(_p(w), _p(out), N, K, int(bool(gate_up_pairs)), dtype_code(w.dtype), _stream()), "dl_pack_weight_tiles")
    return out


def pack_x_tiles(x, out=None):
    """x [M,K] in dl_linear_packed's fragment order (include/dynllava.h); returns a flat tensor."""
    _dev(x, out)
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    need = int(lib().dl_packed_x_bytes(M, K))
    if need < 0:
        raise HipOpsError(f"dl_pack_x_tiles: unsupported shape [{M},{K}]")
    if out is None:
        out = torch.empty(need // x.element_size(), dtype=x.dtype, device=x.device)
    assert out.numel() * out.element_size() >= need and out.is_contiguous()
    _check(lib().dl_pack_x_tiles(_p(x), x.stride(0), _p(out), M, K, dtype_code(x.dtype), _stream()), "dl_pack_x_tiles")
    return out


def linear_packed_workspace(M, N, K, device, epilogue=LP_STORE, units_per_workgroup=0, k_split=1):
    """Zeroed hand-over workspace of a k_split > 1 call (flag words + fp32 tiles); None for k_split == 1.  One workspace serves every call that needs
    at most its size (calls in one stream run one after the other and leave the flag words zero)."""
    need = int(lib().dl_linear_packed_workspace_bytes(M, N, K, int(epilogue), int(units_per_workgroup), int(k_split)))
    if need < 0:
        raise HipOpsError(f"dl_linear_packed_workspace_bytes: unsupported shape / split M={M} N={N} K={K} k_split={k_split}")
    return torch.zeros(max(need, 256), dtype=torch.uint8, device=device) if need else None


def linear_packed(x, wp, N, out=None, epilogue=LP_STORE, resid=None, units_per_workgroup=0, k_split=1, workspace=None, err=None, x_packed_mk=None, y_packed=False, _ablate=0, stamps=None, _trim256=None):
    """out = x [M,K] @ W^T on the packed copy wp of W [N,K] (pack_weight_tiles).  epilogue LP_SILU_PAIR: out [M, N/2] = silu(gate) * up
    (wp packed with gate_up_pairs); LP_RESID: out = resid + x W^T (resid may be out).  x_packed_mk=(M, K): x is pack_x_tiles' output.
    k_split > 1: that many workgroups share each unit set's k range (linear_packed_workspace).
    epilogue LP_PARTS: out is an fp32 tensor [k_split, M, N] of partial sums (no workspace).  y_packed (LP_STORE / LP_SILU_PAIR): out is a flat tensor in
    fragment order, the x_packed_mk input of the next linear_packed."""
    _dev(x, wp, out, resid, workspace, err)
    if x_packed_mk is None:
        assert x.dim() == 2 and x.stride(1) == 1
        M, K = x.shape
        ldx = x.stride(0)
    else:
        M, K = x_packed_mk
        ldx = K
        assert x.is_contiguous() and x.numel() * x.element_size() >= int(lib().dl_packed_x_bytes(M, K))
    assert wp.numel() == N * K and wp.dtype == x.dtype
    n_out = N // 2 if epilogue == LP_SILU_PAIR else N
    if epilogue == LP_PARTS:
        if out is None:
            out = torch.empty((k_split, M, N), dtype=torch.float32, device=x.device)
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= k_split * M * N
        ldy = N
    elif y_packed:
        out = _packed_out(M, n_out, x, out)
        ldy = n_out
    else:
        if out is None:
            out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
        assert out.dim() == 2 and out.shape == (M, n_out) and out.stride(1) == 1 and out.dtype == x.dtype
        ldy = out.stride(0)
    if epilogue == LP_RESID:
        assert resid is not None and resid.shape == (M, N) and resid.stride(1) == 1 and resid.dtype == x.dtype
    if k_split > 1 and epilogue != LP_PARTS:
        need = int(lib().dl_linear_packed_workspace_bytes(M, N, K, int(epilogue), int(units_per_workgroup), int(k_split)))
        assert workspace is not None and workspace.numel() * workspace.element_size() >= need, "linear_packed: workspace missing / too small"
    args_ = (_p(x), ldx, int(x_packed_mk is not None), _p(wp), _p(out), ldy, _p(resid), 0 if resid is None else resid.stride(0), M, N, K,
             int(epilogue) | (LP_Y_PACKED if y_packed else 0) | (int(_ablate) << 8) | (0 if _trim256 is None else (int(_trim256) + 1) << 16), int(units_per_workgroup), int(k_split), _p(workspace), _p(err))
    if stamps is not None:  # measurement: per-wave timeline (tools/lp_timeline.py)
        assert stamps.dtype == torch.int64 and stamps.is_cuda and stamps.is_contiguous()
        _check(lib().dl_linear_packed_stamped(*args_, _p(stamps), dtype_code(x.dtype), _stream()), "dl_linear_packed_stamped")
    else:
        _check(lib().dl_linear_packed(*args_, dtype_code(x.dtype), _stream()), "dl_linear_packed")
    return out[: k_split * M * N].view(k_split, M, N) if epilogue == LP_PARTS else out


LT_BIAS, LT_QGELU, LT_GELU, LT_PARTS = 0, 1, 2, 3


def linear_tiles_ok(M, N, K, dtype):
    """Shapes dl_linear_tiles takes (any row count: the grid grows with it)."""
    return dtype in (torch.bfloat16, torch.float16) and M > 0 and N % 16 == 0 and K % 64 == 0


def tiles_x_numel(M, K):
    need = int(lib().dl_tiles_x_bytes(M, K))
    if need < 0:
        raise HipOpsError(f"dl_tiles_x_bytes: unsupported shape [{M},{K}]")
    return need // 2


def pack_x_rows(x, out=None):
    """x [M,K] in dl_linear_tiles' fragment order (ceil(M / 16) tiles; include/dynllava.h); returns a flat tensor."""
    _dev(x, out)
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    n = tiles_x_numel(M, K)
    if out is None:
        out = torch.empty(n, dtype=x.dtype, device=x.device)
    assert out.numel() >= n and out.is_contiguous() and out.dtype == x.dtype
    _check(lib().dl_pack_x_rows(_p(x), x.stride(0), _p(out), M, K, dtype_code(x.dtype), _stream()), "dl_pack_x_rows")
    return out


def linear_tiles(x, wp, N, bias=None, out=None, epilogue=LT_BIAS, x_packed_mk=None, y_packed=False, tile_shape=0, k_split=1, stamps=None, _wrap=0):
    """out = x [M,K] @ W^T (+ bias, activation) on the packed copy wp of W [N,K] (pack_weight_tiles): the tiled MFMA GEMM of the vision side
    (CLIP tower, projector, vision predictor).  x_packed_mk=(M, K): x is in fragment order (pack_x_rows / layernorm(rows_packed=True) / a previous call's
    y_packed output).  epilogue LT_QGELU / LT_GELU: activation on the rounded Linear output; LT_PARTS: out is fp32 [k_split, M, N] partial sums (no bias)."""
    _dev(x, wp, out, bias)
    if x_packed_mk is None:
        assert x.dim() == 2 and x.stride(1) == 1
        M, K = x.shape
        ldx = x.stride(0)
    else:
        M, K = x_packed_mk
        ldx = K
        assert x.is_contiguous() and (_wrap or x.numel() >= tiles_x_numel(M, K))
    assert (_wrap or wp.numel() == N * K) and wp.dtype == x.dtype
    if epilogue == LT_PARTS:
        assert bias is None and not y_packed
        if out is None:
            out = torch.empty((k_split, M, N), dtype=torch.float32, device=x.device)
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= k_split * M * N
        ldy = N
    elif y_packed:
        n = tiles_x_numel(M, N)
        if out is None:
            out = torch.empty(n, dtype=x.dtype, device=x.device)
        assert out.is_contiguous() and out.dtype == x.dtype and out.numel() >= n
        ldy = N
    else:
        if out is None:
            out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        assert out.dim() == 2 and out.shape == (M, N) and out.stride(1) == 1 and out.dtype == x.dtype
        ldy = out.stride(0)
    if bias is not None:
        assert bias.dtype == x.dtype and bias.numel() == N and bias.is_contiguous()
    args_ = (_p(x), ldx, int(x_packed_mk is not None), _p(wp), _p(bias), _p(out), ldy, int(bool(y_packed)), M, N, K, int(epilogue), int(tile_shape), int(k_split))
    if stamps is not None:
        assert stamps.dtype == torch.int64 and stamps.is_cuda and stamps.is_contiguous()
        args_ = args_[:11] + (int(epilogue) | (int(_wrap) << 8),) + args_[12:]
        _check(lib().dl_linear_tiles_stamped(*args_, _p(stamps), dtype_code(x.dtype), _stream()), "dl_linear_tiles_stamped")
    else:
        _check(lib().dl_linear_tiles(*args_, dtype_code(x.dtype), _stream()), "dl_linear_tiles")
    return out[: k_split * M * N].view(k_split, M, N) if epilogue == LT_PARTS else out


def gemm_smallm_ok(M, N, K, dtype):
    """Shapes dl_gemm_smallm takes (callers use the library GEMM otherwise)."""
    return dtype in (torch.bfloat16, torch.float16) and 0 < M <= 32 and K % 256 == 0 and N % 4 == 0


def gemm_smallm(x, w, out=None, workspace=None, n_slices=0, wg_waves=0, variant=0):
    """out[M,N] = x[M,K] @ w[N,K]^T (nn.Linear, no bias) for M <= 32.  workspace: fp32 scratch for split-K partials (allocated here
    if missing / too small -- pass a persistent one under hipGraph capture)."""
    _dev(x, w, out, workspace)
    assert x.dim() == 2 and x.stride(1) == 1 and w.is_contiguous()
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    assert out.stride(1) == 1
    need = int(lib().dl_gemm_smallm_workspace_bytes(M, N, K, int(n_slices), int(variant)))
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    _check(
        lib().dl_gemm_smallm(_p(x), x.stride(0), _p(w), _p(out), out.stride(0), _p(workspace), M, N, K, int(n_slices), int(wg_waves), int(variant), 0, dtype_code(x.dtype), _stream()),
        "dl_gemm_smallm",
    )
    return out


def gemm_smallm_parts(x, w, workspace, n_slices=0, variant=0):
    """x @ w^T left as fp32 split-K partials in `workspace` (viewed [slices, M, N]); returns (parts, slices) for add_rmsnorm_parts /
    silu_mul_parts, which add the slices themselves (no reduce launch)."""
    _dev(x, w, workspace)
    assert x.dim() == 2 and x.stride(1) == 1 and w.is_contiguous() and workspace.dtype == torch.float32
    M, K = x.shape
    N = w.shape[0]
    s = int(lib().dl_gemm_smallm_slices(M, N, K, int(n_slices), int(variant)))
    assert s >= 1 and workspace.numel() >= s * M * N, "workspace too small (dl_gemm_smallm_workspace_bytes)"
    _check(
        lib().dl_gemm_smallm(_p(x), x.stride(0), _p(w), None, 0, _p(workspace), M, N, K, int(n_slices), 0, int(variant), 1, dtype_code(x.dtype), _stream()),
        "dl_gemm_smallm",
    )
    return workspace[: s * M * N].view(s, M, N), s


def add_rmsnorm_parts(h, parts, w=None, eps=1e-6, out=None, packed=False):
    """h += cast(sum_s parts[s]) in place, then RMSNorm(h) * w -> out (w None: the add only).  parts: [slices, rows, H] fp32.  packed: see rmsnorm."""
    _dev(h, parts, w, out)
    assert h.is_contiguous() and parts.is_contiguous() and parts.dtype == torch.float32 and parts.shape[1:] == h.shape
    rows, H = h.shape
    if packed:
        assert w is not None
        out = _packed_out(rows, H, h, out)
        _check(lib().dl_add_rmsnorm_parts_packed(_p(h), _p(parts), parts.shape[0], _p(w), _p(out), rows, H, eps, dtype_code(h.dtype), _stream()), "dl_add_rmsnorm_parts_packed")
        return out
    if w is not None and out is None:
        out = torch.empty_like(h)
    _check(lib().dl_add_rmsnorm_parts(_p(h), _p(parts), parts.shape[0], _p(w), _p(out) if w is not None else None, rows, H, eps, dtype_code(h.dtype), _stream()), "dl_add_rmsnorm_parts")
    return out if w is not None else None


def silu_mul_parts(parts, out):
    """out[r, :] = silu(gate) * up with gate|up = cast(sum_s parts[s, r, :]) ([slices, rows, 2I] fp32)."""
    _dev(parts, out)
    assert parts.is_contiguous() and parts.dtype == torch.float32 and out.is_contiguous()
    rows, I = out.shape
    assert parts.shape[1] == rows and parts.shape[2] == 2 * I
    _check(lib().dl_silu_mul_parts(_p(parts), parts.shape[0], _p(out), rows, I, dtype_code(out.dtype), _stream()), "dl_silu_mul_parts")
    return out


def kv_pack_rows(k_slab0, v_slab0, layer_stride, n_layers, keep, kv_len, T_cap):
    """Pack the kept rows of a just-appended T-token chunk in place, for n_layers consecutive layer slabs (see include/dynllava.h).
    k_slab0 / v_slab0: [B, nKV, T_cap, d] views of the first layer; keep int32 [B, T]; kv_len int32 [B] (not modified)."""
    _dev(k_slab0, v_slab0, keep, kv_len)
    assert keep.dtype == torch.int32 and keep.is_contiguous() and kv_len.dtype == torch.int32 and k_slab0.stride() == v_slab0.stride() and k_slab0.stride(3) == 1
    B, nKV, _, d = k_slab0.shape
    _check(
        lib().dl_kv_pack_rows(_p(k_slab0), _p(v_slab0), int(layer_stride), int(n_layers), k_slab0.stride(0), k_slab0.stride(1), int(T_cap), _p(keep), _p(kv_len), B, nKV,
                              keep.shape[1], d, dtype_code(k_slab0.dtype), _stream()),
        "dl_kv_pack_rows",
    )


def prompt_layout(input_ids, n_feat, image_token, user_ids):
    """Device-side layout of un-padded one-image-per-row prompts (include/dynllava.h).  Returns dict of device tensors."""
    _dev(input_ids)
    assert input_ids.dtype == torch.int64 and input_ids.is_contiguous() and input_ids.dim() == 2
    B, W = input_ids.shape
    dev = input_ids.device
    out = dict(
        seg=torch.zeros((B, 8), dtype=torch.int32, device=dev), text_src=torch.empty(B * (W - 1), dtype=torch.int64, device=dev),
        text_dst=torch.empty(B * (W - 1), dtype=torch.int64, device=dev), img_dst=torch.empty(B * n_feat, dtype=torch.int64, device=dev),
        img_start=torch.empty(B, dtype=torch.int32, device=dev), err=torch.zeros(1, dtype=torch.int32, device=dev),
    )
    prompt_layout_into(input_ids, n_feat, image_token, user_ids, out)
    return out


def prompt_layout_into(input_ids, n_feat, image_token, user_ids, out, w_true=None, n_drop=0, cu=None, cu2=None, lens=None, last_rows=None):
    """`w_true` (int32 device scalar): the row buffer is a width bucket, only its first w_true columns are the prompt; `cu` / `cu2` int32 [B+1],
    `lens` int32 [2, B], `last_rows` int64 [B]: the prefill plan's device metadata, written at the true lengths (include/dynllava.h)."""
    B, W = input_ids.shape
    for t, dt, n in ((w_true, torch.int32, 1), (cu, torch.int32, B + 1), (cu2, torch.int32, B + 1), (lens, torch.int32, 2 * B), (last_rows, torch.int64, B)):
        if t is not None:
            _dev(t)
            assert t.dtype == dt and t.is_contiguous() and t.numel() == n, (t.dtype, t.shape, dt, n)
    _check(
        lib().dl_prompt_layout(_p(input_ids), B, W, int(n_feat), int(image_token), int(user_ids[0]), int(user_ids[1]), _p(out["seg"]), _p(out["text_src"]), _p(out["text_dst"]),
                               _p(out["img_dst"]), _p(out["img_start"]), _p(out["err"]), _p(w_true), int(n_drop), _p(cu), _p(cu2), _p(lens), _p(last_rows), _stream()),
        "dl_prompt_layout",
    )
