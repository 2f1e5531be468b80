#!/usr/bin/env python
"""N5: fused policy attention (dl_attn_policy_fwd/_bwd) vs the eager op sequence of DML:913-970 on the same GPU, forward + backward,
at the training shape of LLaVA-1.5-7B (32 heads x 128, model_max_length 2048).  Prints time, TFLOP/s (causal flops) and peak memory."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_llava_amd.train_ops import scaled_dot_product_attention_with_policy

dev, dt = "cuda", torch.bfloat16


def eager(q, k, v, policy, eps=1e-6):  # what the reference's training attention does, op for op (masks as is_causal=True)
    B, H, L, d = q.shape
    bias = torch.zeros(B, 1, L, L, dtype=q.dtype, device=q.device)
    bias.masked_fill_(torch.ones(B, 1, L, L, dtype=torch.bool, device=q.device).tril(diagonal=0).logical_not(), float("-inf"))
    w = q @ k.transpose(-2, -1) * (1 / math.sqrt(d))
    w += bias
    pol = policy.reshape(B, 1, 1, L)
    pol = pol + (1.0 - pol) * torch.eye(L, dtype=pol.dtype, device=pol.device).view(1, 1, L, L)
    mx = torch.max(w, dim=-1, keepdim=True)[0]
    w = (w - mx).to(torch.float32).exp_() * pol.to(torch.float32)
    w = (w + eps / L) / (w.sum(dim=-1, keepdim=True) + eps)
    return w.type_as(mx) @ v


def run(fn, q, k, v, pol, do, reps):
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    for _ in range(2):
        for t in (q, k, v, pol): t.grad = None
        fn(q, k, v, pol).backward(do)
    peak = torch.cuda.max_memory_allocated() - base
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize(); tf = tb = 0.0
    for _ in range(reps):
        for t in (q, k, v, pol): t.grad = None
        a.record(); o = fn(q, k, v, pol); b.record(); o.backward(do); c.record(); torch.cuda.synchronize()
        tf += a.elapsed_time(b); tb += b.elapsed_time(c)
    return tf / reps, tb / reps, peak


for B, H, L, d in [(1, 32, 2048, 128), (4, 32, 2048, 128), (1, 32, 631, 128), (8, 16, 577, 64)]:
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v, do = (torch.randn(B, L, H, d, device=dev, dtype=dt, generator=g).transpose(1, 2) for _ in range(4))
    q, k, v = (t.detach().requires_grad_(True) for t in (q, k, v))
    pol = (torch.rand(B, L, 1, device=dev, generator=g) > 0.5).to(dt).requires_grad_(True)
    flops_f = 4 * B * H * L * L * d / 2
    fused = lambda q, k, v, p: scaled_dot_product_attention_with_policy(q, k, v, is_causal=True, policy=p)
    f_f, f_b, f_m = run(fused, q, k, v, pol, do, 10)
    line = f"B={B} H={H} L={L} d={d}: fused fwd {f_f*1e3:8.1f} us ({flops_f/f_f/1e9:6.1f} TFLOP/s)  bwd {f_b*1e3:8.1f} us ({3.5*flops_f/f_b/1e9:6.1f} TFLOP/s, 7 matmuls)  peak {f_m/2**20:8.1f} MiB"
    try:
        e_f, e_b, e_m = run(eager, q, k, v, pol, do, 3)
        line += f" | eager fwd {e_f*1e3:8.1f} us  bwd {e_b*1e3:8.1f} us  peak {e_m/2**20:8.1f} MiB | speedup fwd {e_f/f_f:.1f}x bwd {e_b/f_b:.1f}x"
    except torch.OutOfMemoryError:
        line += " | eager: out of memory"
    print(line)
