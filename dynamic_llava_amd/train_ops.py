"""N5: the training-time ops of Dynamic-LLaVA behind the reference's own function names.

    scaled_dot_product_attention_with_policy(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, policy=None)
        -- llava/model/language_model/dynamic_modeling_llama.py:933-970 (+ softmax_with_policy, :913-930), called at :1103
    gumbel_hard_keep(log_probs, tau, prev_decision)
        -- `F.gumbel_softmax(image_pred_score, tau=self.gumbel_tau, hard=True)[:, :, 0:1] * image_prev_decision`, :1868-1876
           (the same expression at :2075 and :2204 for the text predictors)

Both are `torch.autograd.Function`s over the C ABI (`dl_attn_policy_fwd/_bwd`, `dl_gumbel_hard_keep_fwd/_bwd`): fused HIP kernels for
gfx950, no [B,H,N,N] tensor is ever materialised (the reference builds five fp32/bf16 ones per layer in the forward and autograd keeps
them for the backward).  There is no fallback: CPU tensors or a missing extension raise.
"""
from __future__ import annotations

import math

import torch

from . import hip_ops as ops


class _SdpaWithPolicy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, value, policy, bias, is_causal, scale, eps):
        B, H, L, d = query.shape
        # one layout for q / k / v ([B,L,H,d] storage viewed as [B,H,L,d], which is what the attention module's transpose(1, 2) yields)
        if not (query.stride(3) == 1 and key.stride() == query.stride() and value.stride() == query.stride() and all(s % 8 == 0 for s in query.stride()[:3])):
            query, key, value = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (query, key, value))
        pol = policy.reshape(B, L).float().contiguous()
        out = torch.empty(B, L, H, d, device=query.device, dtype=query.dtype).transpose(1, 2)  # the caller's transpose(1, 2).contiguous() is then free
        row_max = torch.empty(B, H, L, device=query.device, dtype=torch.float32)
        row_denom = torch.empty_like(row_max)
        ws = ops.attn_policy_workspace(B, H, L, d, query.device)
        ops.attn_policy_fwd(query, key, value, out, pol, bias, row_max, row_denom, ws, is_causal, scale, eps, L)
        ctx.save_for_backward(query, key, value, out, pol, bias if bias is not None else torch.empty(0), row_max, row_denom)
        ctx.has_bias, ctx.is_causal, ctx.scale, ctx.eps, ctx.policy_shape, ctx.policy_dtype = bias is not None, is_causal, scale, eps, policy.shape, policy.dtype
        return out

    @staticmethod
    def backward(ctx, d_out):
        query, key, value, out, pol, bias, row_max, row_denom = ctx.saved_tensors
        B, H, L, d = query.shape
        if d_out.stride() != out.stride():
            d_out = d_out.transpose(1, 2).contiguous().transpose(1, 2)
        dq, dk, dv = (torch.empty(B, L, H, d, device=query.device, dtype=query.dtype).transpose(1, 2) for _ in range(3))
        dpol_heads = torch.empty(B, H, L, device=query.device, dtype=torch.float32)
        ws = ops.attn_policy_workspace(B, H, L, d, query.device)
        ops.attn_policy_bwd(query, key, value, out, d_out, dq, dk, dv, pol, bias if ctx.has_bias else None, row_max, row_denom, dpol_heads, ws,
                            ctx.is_causal, ctx.scale, ctx.eps, L)
        d_policy = dpol_heads.sum(dim=1).reshape(ctx.policy_shape).to(ctx.policy_dtype)
        return dq, dk, dv, d_policy, None, None, None, None


def scaled_dot_product_attention_with_policy(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, policy=None, eps=1e-6):
    """Drop-in for DML:933-970.  query / key / value: [B,H,L,d] (bf16 / f16, d in {64, 128}, L == S); attn_mask: None, a bool mask or an
    additive mask broadcastable to [B,1,L,L]; policy: [B,L,1] keep decisions (differentiable).  Returns [B,H,L,d]."""
    if policy is None:
        return torch.nn.functional.scaled_dot_product_attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)
    if dropout_p != 0.0:
        raise NotImplementedError("attention dropout is not supported by the fused policy attention (attention_dropout is 0.0 in every shipped config)")
    B, H, L, d = query.shape
    if key.shape != query.shape or value.shape != query.shape:
        raise ValueError("self-attention shapes expected: query, key and value must all be [B,H,L,d] (repeat_kv already applied, DML:1086-1087)")
    bias = None
    if attn_mask is not None:
        if is_causal:
            raise AssertionError("is_causal and attn_mask are exclusive (DML:944)")
        if attn_mask.dtype == torch.bool:
            bias = torch.zeros(attn_mask.shape, dtype=query.dtype, device=query.device).masked_fill_(attn_mask.logical_not(), float("-inf"))
        else:
            bias = attn_mask.to(query.dtype)
        while bias.dim() < 4:
            bias = bias[None]
        if bias.shape[1] != 1:
            raise NotImplementedError("per-head masks are not supported (the reference passes [B,1,L,S], DML:1089-1093)")
        bias = bias.expand(bias.shape[0], 1, L, L)
        if bias.stride(3) != 1:
            bias = bias.contiguous()
    scale_factor = 1.0 / math.sqrt(d) if scale is None else float(scale)
    return _SdpaWithPolicy.apply(query, key, value, policy, bias, bool(is_causal), scale_factor, float(eps))


class _GumbelHardKeep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, gumbels, prev_decision, tau):
        lp, g, prev = log_probs.contiguous(), gumbels.contiguous(), prev_decision.contiguous()
        keep = torch.empty_like(prev)
        y_soft = torch.empty_like(lp)
        ops.gumbel_hard_keep_fwd(lp, g, prev, keep, y_soft, tau)
        ctx.save_for_backward(prev, y_soft)
        ctx.tau = tau
        return keep

    @staticmethod
    def backward(ctx, d_keep):
        prev, y_soft = ctx.saved_tensors
        d_lp = torch.empty_like(y_soft)
        d_prev = torch.empty_like(prev) if ctx.needs_input_grad[2] else None
        ops.gumbel_hard_keep_bwd(d_keep.contiguous(), prev, y_soft, d_lp, ctx.tau, d_prev)
        return d_lp, None, d_prev, None


def gumbel_hard_keep(log_probs, tau, prev_decision, gumbels=None):
    """`F.gumbel_softmax(log_probs, tau=tau, hard=True)[:, :, 0:1] * prev_decision` (DML:1868-1876).  log_probs: [B,N,2]; prev_decision:
    [B,N,1].  The Gumbel noise is drawn exactly as torch.nn.functional.gumbel_softmax draws it (same generator, same call), unless given.
    The straight-through gradient flows to log_probs (and to prev_decision when it requires one: stacked sparsification layers)."""
    if gumbels is None:
        gumbels = -torch.empty_like(log_probs, memory_format=torch.legacy_contiguous_format).exponential_().log()
    return _GumbelHardKeep.apply(log_probs, gumbels, prev_decision, float(tau))
