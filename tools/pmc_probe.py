#!/usr/bin/env python
"""Launches the HBM-bound kernels on known shapes so that a `rocprofv3 --kernel-trace --pmc <counter>` pass can attribute
FETCH_SIZE / WRITE_SIZE to them (one counter group per pass, no other trace domains).  Prints the algorithmic bytes per
launch as JSON so that tools/pmc_report.py can put measured traffic next to them."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd import hip_ops as ops
from oracle.ref_cpu import rope_table

dev, dt = "cuda", torch.bfloat16
nH, d = 32, 128
H = nH * d
cos, sin = (t.to(dev) for t in rope_table(d, 4096, 10000.0, dt))
plan = []


def attn(tag, B, Ts, ns):
    T_cap = max(Ts) + 1
    lens = torch.tensor([t - 1 for t in Ts], dtype=torch.int32, device=dev)
    qkv = torch.randn(B, 3 * H, device=dev, dtype=dt)
    out = torch.empty(B, H, device=dev, dtype=dt)
    ws = ops.attn_decode_workspace(B, nH, d, 32, dev)
    for rep in range(3):  # fresh slabs every launch: cold in L2 / Infinity Cache
        k = torch.randn(B, nH, T_cap, d, device=dev, dtype=dt)
        v = torch.randn_like(k)
        torch.cuda.synchronize()
        ops.attn_decode_rope(qkv, cos, sin, lens, lens, k, v, out, ws, ns, nH, nH, d)
        torch.cuda.synchronize()
    plan.append({"tag": tag, "kernel": "attn_decode_split_kernel", "grid_x": ns, "B": B, "algorithmic_bytes": sum(2 * t * H * 2 + 2 * H * 2 for t in Ts)})


attn("decode_attn B=1 T=226", 1, [226], 8)
attn("decode_attn B=1 T=695", 1, [695], 8)
attn("decode_attn B=32 ragged", 32, [200 + (i * 701) % 700 for i in range(32)], 1)
attn("decode_attn B=32 T=2048", 32, [2048] * 32, 1)

for tag, N, K, mode in [("gemv qkv", 12288, 4096, ops.GEMV_ADDNORM), ("gemv o", 4096, 4096, ops.GEMV_PLAIN), ("gemv gate_up", 22016, 4096, ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR), ("gemv down", 4096, 11008, ops.GEMV_PLAIN)]:
    pair = bool(mode & ops.GEMV_OUT_SILU_PAIR)
    y = torch.empty(1, N // 2 if pair else N, device=dev, dtype=dt)
    x = torch.randn(1, K, device=dev, dtype=dt)
    h, h2, dl = (torch.randn(1, K, device=dev, dtype=dt) for _ in range(3))
    nw = torch.ones(K, device=dev, dtype=dt)
    for rep in range(3):
        w = torch.randn(N, K, device=dev, dtype=dt)
        torch.cuda.synchronize()
        if (mode & 3) == ops.GEMV_ADDNORM:
            ops.gemv(w, y, mode=mode, h_in=h, h_out=h2, delta=dl, norm_w=nw, eps=1e-5)
        else:
            ops.gemv(w, y, x=x)
        torch.cuda.synchronize()
    plan.append({"tag": tag, "kernel": "gemv_kernel", "N": N, "K": K, "algorithmic_bytes": N * K * 2})

# round 4: the two FUSED launches of the product's batch-1 decode step (the PMC passes only covered the separate kernels)
from dynamic_llava_amd.model import TextPredictor  # noqa: E402

for T in (226, 380):  # dl_gemv_qkv_attn: q|k|v projection + single-workgroup attention over T keys (the new token included)
    N, K = 3 * H, H
    T_cap = T + 8
    h, h2, dl = (torch.randn(1, K, device=dev, dtype=dt) for _ in range(3))
    nw = torch.ones(K, device=dev, dtype=dt)
    qkv, out = torch.empty(1, N, device=dev, dtype=dt), torch.empty(1, H, device=dev, dtype=dt)
    lens = torch.tensor([T - 1], dtype=torch.int32, device=dev)
    gran = ops.gemv_qkv_attn_workspace(nH, nH, d, dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep in range(3):
        w = torch.randn(N, K, device=dev, dtype=dt) * 0.02
        k = torch.randn(1, nH, T_cap, d, device=dev, dtype=dt)
        v = torch.randn_like(k)
        torch.cuda.synchronize()
        ops.gemv_qkv_attn(w, qkv, h, h2, dl, nw, 1e-5, cos, sin, lens, lens, k, v, out, gran, rep + 1, nH, nH, d, err=err)
        torch.cuda.synchronize()
    assert int(err.item()) == 0
    plan.append({"tag": f"gemv_qkv_attn T={T}", "kernel": "gemv_qkv_attn_kernel", "N": N, "K": K, "algorithmic_bytes": N * K * 2 + 2 * T * H * 2 + 2 * H * 2})

tp = TextPredictor(H, 512).to(device=dev, dtype=dt)
N, K = 22016, H
h, h2, dl = (torch.randn(1, K, device=dev, dtype=dt) for _ in range(3))
nw = torch.ones(K, device=dev, dtype=dt)
y = torch.empty(1, N // 2, device=dev, dtype=dt)
tp_ws, tp_lg, dec = ops.text_predictor_workspace(1, tp.d_model, dev), torch.zeros(1, 2, device=dev), torch.ones(1, dtype=torch.int32, device=dev)
pos = torch.tensor([300], dtype=torch.int32, device=dev)
gran = ops.gemv_gu_tp_workspace(tp.d_model, dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(3):
    w = torch.randn(N, K, device=dev, dtype=dt) * 0.02
    torch.cuda.synchronize()
    ops.gemv_gu_tp(w, y, h, h2, dl, nw, 1e-5, tp._weights(), tp.d_model, tp_ws, tp_lg, dec, pos, gran, rep + 1, err=err)
    torch.cuda.synchronize()
assert int(err.item()) == 0
tp_bytes = sum(p_.numel() * 2 for p_ in tp.parameters())
plan.append({"tag": "gemv_gu_tp", "kernel": "gemv_gu_tp_kernel", "N": N, "K": K, "algorithmic_bytes": N * K * 2 + tp_bytes + K * 2})

# round 5: dl_linear_packed on the two prefill shapes the product runs on it (M = 170 packed rows; activations in fragment order): algorithmic bytes =
# the weights once + X once + Y once (what a perfect kernel moves through HBM; X is read by every workgroup, but out of L2 after its first touch)
for tag, N, K, epi, nu, ks in [("linear_packed qkv M=170", 12288, 4096, ops.LP_STORE, 6, 2), ("linear_packed gate_up+silu M=170", 22016, 4096, ops.LP_SILU_PAIR, 6, 1)]:
    M = 170
    x = torch.randn(M, K, device=dev, dtype=dt)
    xp = ops.pack_x_tiles(x)
    wsb = ops.linear_packed_workspace(M, N, K, dev, epi, nu, ks)
    n_out = N // 2 if epi == ops.LP_SILU_PAIR else N
    out = torch.empty(M, n_out, device=dev, dtype=dt)
    for rep in range(3):
        wp = ops.pack_weight_tiles(torch.randn(N, K, device=dev, dtype=dt) * 0.02, gate_up_pairs=(epi == ops.LP_SILU_PAIR))
        torch.cuda.synchronize()
        ops.linear_packed(xp, wp, N, out=out, epilogue=epi, units_per_workgroup=nu, k_split=ks, workspace=wsb, x_packed_mk=(M, K))
        torch.cuda.synchronize()
    plan.append({"tag": tag, "kernel": "linear_packed_kernel", "N": N, "K": K, "algorithmic_bytes": N * K * 2 + M * K * 2 + M * n_out * 2})

for rows in (631, 170):
    x = torch.randn(rows, H, device=dev, dtype=dt)
    w = torch.ones(H, device=dev, dtype=dt)
    for rep in range(3):
        x = torch.randn(rows, H, device=dev, dtype=dt)
        torch.cuda.synchronize()
        ops.rmsnorm(x, w, 1e-5)
        torch.cuda.synchronize()
    plan.append({"tag": f"rmsnorm [{rows},{H}]", "kernel": "rmsnorm_kernel", "rows": rows, "algorithmic_bytes": 2 * rows * H * 2 + H * 2})
print("PLAN " + json.dumps(plan))
