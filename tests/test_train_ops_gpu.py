"""N5 on the GPU: the fused policy attention (forward + backward) and the Gumbel hard keep mask, through the C ABI, against the oracle
restatements of DML:913-970 / DML:1868-1876 (oracle/ref_cpu.py, pinned to the reference by tests/golden/train_ops.npz).

Floating point: the truth is the oracle evaluated in fp32 (autograd for the gradients) on the same bf16/f16-representable inputs.  The
eager reference rounds its [B,H,N,N] intermediates to the model dtype several times; the bar is its own noise class:
    max|hip - truth| <= 2 * max|oracle(model dtype) - truth| + 1e-3 * max|truth|."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _inputs(seed, B, H, L, d, dtype):
    g = torch.Generator().manual_seed(seed)
    q, k, v, do = (torch.randn(B, L, H, d, generator=g).to(dtype).transpose(1, 2) for _ in range(4))  # the module's layout: [B,L,H,d] viewed [B,H,L,d]
    policy = (torch.rand(B, L, 1, generator=g) > 0.4).float()
    policy[:, : max(1, L // 4)] = 1.0
    policy[:, L // 2] = 0.37
    return q, k, v, do, policy.to(dtype)


def _mask(kind, B, L, dtype):
    if kind == "causal" or kind == "none":
        return None
    neg = torch.finfo(dtype).min
    m = torch.zeros(B, 1, L, L, dtype=dtype)
    m.masked_fill_(torch.ones(L, L, dtype=torch.bool).tril().logical_not()[None, None], neg)
    if B > 1:
        m[1:, :, :, L - 5 :] = neg  # right padding of the later rows
    if kind == "bool":
        return m == 0
    return m


def _oracle(q, k, v, do, pol, mask, causal, dtype):
    q, k, v, pol = (t.detach().to(dtype).clone().requires_grad_(True) for t in (q, k, v, pol))
    if mask is not None and mask.dtype != torch.bool:
        mask = mask.to(dtype)
    o = O.sdpa_with_policy(q, k, v, attn_mask=mask, is_causal=causal, policy=pol)
    o.backward(do.to(dtype))
    return [t.detach().float() for t in (o, q.grad, k.grad, v.grad, pol.grad)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,L,d,kind", [(2, 4, 200, 128, "causal"), (1, 2, 64, 128, "causal"), (2, 2, 131, 64, "causal"), (2, 3, 150, 128, "additive"),
                                          (2, 2, 70, 64, "bool"), (1, 2, 96, 128, "none"), (1, 8, 333, 128, "causal"),
                                          (2, 32, 520, 128, "causal"), (2, 32, 449, 128, "additive")])  # the last two: >= 256 workgroups -> the 8-wave backward kernels
def test_sdpa_with_policy_forward_backward(dtype, B, H, L, d, kind):
    from dynamic_llava_amd.train_ops import scaled_dot_product_attention_with_policy

    q, k, v, do, pol = _inputs(100 + L, B, H, L, d, dtype)
    mask = _mask(kind, B, L, dtype)
    causal = kind == "causal"
    truth = _oracle(q, k, v, do, pol, mask if (mask is None or mask.dtype == torch.bool) else mask.float(), causal, torch.float32)
    noisy = _oracle(q, k, v, do, pol, mask, causal, dtype)
    dev = "cuda"
    qd, kd, vd, pd = (t.to(dev).requires_grad_(True) for t in (q, k, v, pol))
    o = scaled_dot_product_attention_with_policy(qd, kd, vd, attn_mask=None if mask is None else mask.to(dev), is_causal=causal, policy=pd)
    assert o.shape == (B, H, L, d) and o.transpose(1, 2).is_contiguous()
    o.backward(do.to(dev))
    got = [t.detach().float().cpu() for t in (o, qd.grad, kd.grad, vd.grad, pd.grad)]
    for name, g, t, n in zip(("out", "dq", "dk", "dv", "dpolicy"), got, truth, noisy):
        assert g.shape == t.shape, name
        err, ref_err, mag = float((g - t).abs().max()), float((n - t).abs().max()), float(t.abs().max())
        assert err <= 2 * ref_err + 1e-3 * mag, f"{name}: |hip - fp32 truth| = {err:.3e}, the eager reference's own error = {ref_err:.3e}, magnitude {mag:.3e}"
    # a dropped key only reaches its own row: the output of row i must not depend on v_j for a dropped j != i (up to eps / N)
    j = int((pol[0, :, 0] == 0).nonzero()[0])
    v2 = v.clone()
    v2[0, :, j] += 4.0
    o2 = scaled_dot_product_attention_with_policy(q.to(dev), k.to(dev), v2.to(dev), attn_mask=None if mask is None else mask.to(dev), is_causal=causal, policy=pol.to(dev))
    changed = (o2.float().cpu() != got[0])[0].float().mean(dim=(0, 2))  # per query row: fraction of output elements that moved at all
    others = torch.cat([changed[:j], changed[j + 1 :]])
    assert float(changed[j]) > 0.5 and float(others.mean()) < 1e-3  # elsewhere only the eps / N leak: a rare one-ulp rounding flip


def test_sdpa_with_policy_all_kept_equals_plain_sdpa():
    """policy == 1 everywhere: softmax_with_policy degenerates to softmax (up to eps) -- against torch's own SDPA on the GPU."""
    from dynamic_llava_amd.train_ops import scaled_dot_product_attention_with_policy

    q, k, v, _, _ = _inputs(3, 2, 4, 257, 128, torch.bfloat16)
    q, k, v = (t.cuda() for t in (q, k, v))
    o = scaled_dot_product_attention_with_policy(q, k, v, is_causal=True, policy=torch.ones(2, 257, 1, device="cuda", dtype=torch.bfloat16))
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=True)
    assert float((o.float() - ref).abs().max()) < 2e-2


def test_sdpa_with_policy_refuses_what_it_does_not_implement():
    from dynamic_llava_amd import hip_ops as ops
    from dynamic_llava_amd.train_ops import scaled_dot_product_attention_with_policy

    q = torch.randn(1, 2, 16, 32, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ops.HipOpsError):  # head_dim 32
        scaled_dot_product_attention_with_policy(q, q, q, is_causal=True, policy=torch.ones(1, 16, 1, device="cuda", dtype=torch.bfloat16))
    q = torch.randn(1, 2, 16, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        scaled_dot_product_attention_with_policy(q, q, q, dropout_p=0.1, policy=torch.ones(1, 16, 1, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(ops.HipOpsError):  # no CPU fallback
        scaled_dot_product_attention_with_policy(q.cpu(), q.cpu(), q.cpu(), is_causal=True, policy=torch.ones(1, 16, 1, dtype=torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gumbel_hard_keep(dtype):
    from dynamic_llava_amd.train_ops import gumbel_hard_keep

    g = torch.Generator().manual_seed(9)
    B, N = 3, 576
    lp = torch.log_softmax(torch.randn(B, N, 2, generator=g), -1).to(dtype)
    prev = (torch.rand(B, N, 1, generator=g) > 0.2).to(dtype)
    noise = (-torch.empty(B, N, 2).exponential_(generator=g).log()).to(dtype)
    w = torch.randn(B, N, 1, generator=g).to(dtype)
    lo = lp.clone().requires_grad_(True)
    po = prev.clone().requires_grad_(True)
    keep_o = O.gumbel_hard_keep(lo, noise, 0.7, po)
    keep_o.backward(w)
    ld, pd = lp.cuda().requires_grad_(True), prev.cuda().requires_grad_(True)
    keep = gumbel_hard_keep(ld, 0.7, pd, gumbels=noise.cuda())
    keep.backward(w.cuda())
    ulp = {torch.float32: 2e-6, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}[dtype]
    a, b = keep.detach().float().cpu(), keep_o.detach().float()
    # the hard decision is the same everywhere (exact ties y0 == y1 do occur in bf16: both sides take class 0, torch.max's first index)
    assert torch.equal(a > 0.5, b > 0.5)
    assert float((a - b).abs().max()) <= ulp
    for got, ref in ((ld.grad, lo.grad), (pd.grad, po.grad)):
        got, ref = got.float().cpu(), ref.float()
        assert float((got - ref).abs().max()) <= 2 * ulp * max(1.0, float(ref.abs().max()))
    # drawing the noise inside: the same torch generator call as F.gumbel_softmax (DML:1870)
    torch.manual_seed(5)
    k1 = gumbel_hard_keep(lp.cuda(), 0.7, prev.cuda())
    torch.manual_seed(5)
    k2 = torch.nn.functional.gumbel_softmax(lp.cuda(), tau=0.7, hard=True)[:, :, 0:1] * prev.cuda()
    assert float((k1.float() - k2.float()).abs().max()) <= ulp
