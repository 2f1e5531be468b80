"""TEST INFRASTRUCTURE ONLY -- golden vectors for the N5 training-time ops, produced by RUNNING THE REFERENCE's own functions
(`softmax_with_policy`, `scaled_dot_product_attention_with_policy`, DML:913-970) and torch's `F.gumbel_softmax` (as called at
DML:1868-1876) with autograd, in the build container:

    python -m oracle.make_golden_train        ->  tests/golden/train_ops.npz

tests/test_oracle_golden.py::test_train_ops_* pins oracle/ref_cpu.py's restatements to these values (forward AND gradients)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle._ref_import import import_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "train_ops.npz")


def case_inputs(seed, B, H, L, d, dtype):
    g = torch.Generator().manual_seed(seed)
    q, k, v, do = (torch.randn(B, H, L, d, generator=g).to(dtype) for _ in range(4))
    policy = (torch.rand(B, L, 1, generator=g) > 0.4).float()
    policy[:, : L // 4] = 1.0  # system-prompt tokens are always kept (DML:1879-1896)
    policy[:, L // 2] = 0.37   # a fractional value: the gradient w.r.t. the policy must not rely on it being 0 / 1
    return q, k, v, do, policy.to(dtype)


def hf_mask(B, L, n_pad, dtype):
    """causal + right-padding additive mask as transformers' _prepare_4d_causal_attention_mask builds it (finfo.min, not -inf)."""
    m = torch.zeros(B, 1, L, L, dtype=dtype)
    neg = torch.finfo(dtype).min
    m.masked_fill_(torch.ones(L, L, dtype=torch.bool).tril().logical_not()[None, None], neg)
    m[1:, :, :, L - n_pad :] = neg
    return m


def main():
    import_reference()
    dml = sys.modules["llava.model.language_model.dynamic_modeling_llama"]
    out = {}
    for name, dtype, causal, n_pad in [("f32_causal", torch.float32, True, 0), ("bf16_causal", torch.bfloat16, True, 0),
                                       ("f32_mask", torch.float32, False, 7), ("bf16_mask", torch.bfloat16, False, 7)]:
        B, H, L, d = 2, 2, 40, 32
        q, k, v, do, pol = case_inputs(11, B, H, L, d, dtype)
        q, k, v, pol = (t.clone().requires_grad_(True) for t in (q, k, v, pol))
        mask = None if causal else hf_mask(B, L, n_pad, dtype)
        o = dml.scaled_dot_product_attention_with_policy(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=causal, policy=pol)
        o.backward(do)
        for key, t in [("o", o), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad), ("dpolicy", pol.grad)]:
            out[f"sdpa_{name}_{key}"] = t.detach().float().numpy()
    # softmax_with_policy alone on a small score tensor
    g = torch.Generator().manual_seed(5)
    attn = torch.randn(2, 3, 24, 24, generator=g)
    pol = (torch.rand(2, 24, 1, generator=g) > 0.5).float()
    out["softmax_in_attn"], out["softmax_in_policy"] = attn.numpy(), pol.numpy()
    out["softmax_out"] = dml.softmax_with_policy(attn, pol).numpy()
    # gumbel hard keep (DML:1868-1876): torch draws the noise; it is replayed here through the same generator state
    for name, dtype in [("f32", torch.float32), ("bf16", torch.bfloat16)]:
        g = torch.Generator().manual_seed(3)
        logit = torch.randn(2, 36, 2, generator=g).to(dtype)
        prev = (torch.rand(2, 36, 1, generator=g) > 0.2).to(dtype)
        lp = F.log_softmax(logit, dim=-1).requires_grad_(True)
        torch.manual_seed(123)
        keep = F.gumbel_softmax(lp, tau=0.7, hard=True)[:, :, 0:1] * prev
        torch.manual_seed(123)
        gumbels = -torch.empty_like(lp).exponential_().log()  # the draw F.gumbel_softmax made (same global generator state)
        w = torch.randn(2, 36, 1, generator=g).to(dtype)
        keep.backward(w)
        out[f"gumbel_{name}_logp"], out[f"gumbel_{name}_noise"] = lp.detach().float().numpy(), gumbels.float().numpy()
        out[f"gumbel_{name}_prev"], out[f"gumbel_{name}_w"] = prev.float().numpy(), w.float().numpy()
        out[f"gumbel_{name}_keep"], out[f"gumbel_{name}_dlogp"] = keep.detach().float().numpy(), lp.grad.float().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
