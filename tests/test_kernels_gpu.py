"""GPU parity tests, kernel level: every C-ABI entry point (through ctypes, on a real MI355X) against the
oracle's primitive of the same name or a plain fp32 PyTorch evaluation of the same formula.

Tolerances: integer / index / copy results are bit-exact.  Element-wise kernels that reproduce the eager
reference's rounding points are required to match the oracle evaluated in the SAME dtype to within one
unit in the last place of that dtype (fp32 statistics may differ in the last bit).  Attention / GEMM
kernels are compared against an fp32 evaluation of the same rounded inputs with an absolute+relative bound
stated at each test."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx  # noqa: E402
from oracle import ref_cpu as orc  # noqa: E402

DTYPES = [torch.float32, torch.float16, torch.bfloat16]
ULP = {torch.float32: 2.0**-23, torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}


@pytest.fixture(scope="module")
def ops():
    from dynamic_llava_amd import hip_ops

    hip_ops.require_gpu()
    return hip_ops


def _close_ulp(a, b, dtype, n_ulp=1.0, atol=0.0, mag=None):
    """|a-b| <= n_ulp * ulp(dtype) * max(|a|,|b|[,mag]) + atol.  `mag`: magnitude of the operands of a final add
    (after cancellation the result's own magnitude is not the relevant scale)."""
    if dtype == torch.float32:
        n_ulp = max(n_ulp, 8.0)  # fp32: rsqrt / summation-order differences of a few ulp; 16-bit results absorb them
    a, b = a.float().cpu(), b.float().cpu()
    scale = torch.maximum(a.abs(), b.abs())
    if mag is not None:
        scale = torch.maximum(scale, mag.float().cpu().abs())
    tol = n_ulp * ULP[dtype] * scale + atol
    bad = (a - b).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} / {a.numel()} elements differ by more than {n_ulp} ulp; max abs diff {float((a - b).abs().max())}"


def _frac_exact(a, b, dtype=None):
    if dtype == torch.float32:
        return 1.0  # exact-match fractions are only meaningful after rounding to a 16-bit dtype
    return float((a.float().cpu() == b.float().cpu()).float().mean())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,H", [(1, 4096), (37, 4096), (5, 256), (3, 5120)])
def test_rmsnorm(ops, dtype, rows, H):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(rows, H, generator=g) * 3).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    ref = orc.rmsnorm(x, w, 1e-5)
    out = ops.rmsnorm(x.cuda(), w.cuda(), 1e-5)
    _close_ulp(out, ref, dtype, 1.0)
    assert _frac_exact(out, ref, dtype) > 0.99


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_rmsnorm(ops, dtype):
    g = torch.Generator().manual_seed(1)
    h = torch.randn(19, 4096, generator=g).to(dtype)
    d = torch.randn(19, 4096, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(4096, generator=g)).to(dtype)
    hh = h.cuda().clone()
    out = ops.add_rmsnorm(hh, d.cuda(), w.cuda(), 1e-5)
    ref_h = h + d
    assert torch.equal(hh.cpu(), ref_h), "residual add must be bit-exact"
    _close_ulp(out, orc.rmsnorm(ref_h, w, 1e-5), dtype, 1.0)
    hh2 = h.cuda().clone()
    assert ops.add_rmsnorm(hh2, d.cuda(), None, 1e-5) is None
    assert torch.equal(hh2.cpu(), ref_h)


@pytest.mark.parametrize("dtype", DTYPES)
def test_silu_mul(ops, dtype):
    g = torch.Generator().manual_seed(2)
    gu = (torch.randn(7, 2 * 11008, generator=g) * 2).to(dtype)
    ref = F.silu(gu[:, :11008]) * gu[:, 11008:]
    out = ops.silu_mul(gu.cuda())
    _close_ulp(out, ref, dtype, 1.0, atol=1e-30)
    assert _frac_exact(out, ref, dtype) > 0.98


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fast_activations_have_the_exact_expressions_bits(ops, dtype, monkeypatch):
    """dl_silu_mul / dl_quick_gelu evaluate silu / sigmoid with v_exp_f32 / v_rcp_f32 and take the exact expression (expf, IEEE divide) only near a rounding boundary of the
    16-bit type, for large arguments and for fp16-subnormal results (csrc/act_round.h, late round 6).  DL_EXACT_ACT=1 selects the all-exact instantiation of the same kernel:
    every element of 4 x 2.3 M values spread from 1e-3 to 150 in magnitude must agree bit for bit; and dl_silu_mul_parts (which keeps the exact form) on the same values as
    one fp32 slice is a second, independent anchor."""
    g = torch.Generator().manual_seed(12)
    for scale in (0.01, 1.0, 6.0, 40.0):
        gu = (torch.randn(104, 2 * 11008, generator=g) * scale).to(dtype).cuda()
        x = (torch.randn(577, 4096, generator=g) * scale).to(dtype).cuda()
        monkeypatch.delenv("DL_EXACT_ACT", raising=False)
        fast_s, fast_q = ops.silu_mul(gu), ops.quick_gelu(x)
        monkeypatch.setenv("DL_EXACT_ACT", "1")
        exact_s, exact_q = ops.silu_mul(gu), ops.quick_gelu(x)
        monkeypatch.delenv("DL_EXACT_ACT", raising=False)
        assert torch.equal(fast_s, exact_s), f"silu_mul, scale {scale}: {int((fast_s != exact_s).sum())} elements differ"
        assert torch.equal(fast_q, exact_q), f"quick_gelu, scale {scale}: {int((fast_q != exact_q).sum())} elements differ"
        parts = gu.float()[None].contiguous()
        assert torch.equal(ops.silu_mul_parts(parts, torch.empty_like(fast_s)), fast_s)
        assert torch.isfinite(fast_s.float()).all() and torch.isfinite(fast_q.float()).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_gather(ops, dtype):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, 4096, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(4096, generator=g)).to(dtype)
    b = (0.1 * torch.randn(4096, generator=g)).to(dtype)
    idx = torch.tensor([3, 49, 0, 17, 17], dtype=torch.int32)
    ref = F.layer_norm(x[idx.long()], (4096,), w, b, 1e-5)
    out = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5, row_index=idx.cuda(), rows=5)
    _close_ulp(out, ref, dtype, 1.0, atol=2e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,H", [(577, 1024), (37, 64), (3, 4096)])
def test_add_layernorm(ops, dtype, rows, H):
    """CLIP encoder glue: h = cast(h + delta) in place (bit-exact), out = LayerNorm(h) as torch rounds it."""
    g = torch.Generator().manual_seed(31)
    h = torch.randn(rows, H, generator=g).to(dtype)
    dl = torch.randn(rows, H, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    b = (0.1 * torch.randn(H, generator=g)).to(dtype)
    h_ref = h + dl
    ref = F.layer_norm(h_ref, (H,), w, b, 1e-5)
    hd = h.cuda().clone()
    out = ops.add_layernorm(hd, dl.cuda(), w.cuda(), b.cuda(), 1e-5)
    assert torch.equal(hd.cpu(), h_ref), "residual stream must be bit-exact"
    _close_ulp(out, ref, dtype, 1.0, atol=2e-6)
    hd2 = h.cuda().clone()
    assert ops.add_layernorm(hd2, dl.cuda()) is None and torch.equal(hd2.cpu(), h_ref), "add-only form"


@pytest.mark.parametrize("dtype", DTYPES)
def test_quick_gelu(ops, dtype):
    """x * sigmoid(1.702 x) with the eager op's three roundings: exact up to the last-bit difference of expf implementations."""
    g = torch.Generator().manual_seed(32)
    x = (3 * torch.randn(577, 4096, generator=g)).to(dtype)
    ref = x * torch.sigmoid(1.702 * x)
    out = ops.quick_gelu(x.cuda())
    _close_ulp(out, ref, dtype, 1.0, atol=1e-6)
    if dtype != torch.float32:
        assert _frac_exact(out, ref, dtype) > 0.999


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nH,nKV,d", [(32, 32, 128), (8, 2, 128), (2, 2, 64)])
def test_rope_kv_write_bit_exact(ops, dtype, nH, nKV, d):
    """RoPE reproduces the eager op's three roundings (DML:283-284) => bit-exact vs the oracle in every dtype."""
    g = torch.Generator().manual_seed(4)
    lens = [5, 1, 9]
    B, total, T_cap = len(lens), sum(lens), 16
    cu = torch.tensor([0, 5, 6, 15], dtype=torch.int32)
    qkv = torch.randn(total, (nH + 2 * nKV) * d, generator=g).to(dtype)
    pos = torch.randint(0, 700, (total,), generator=g, dtype=torch.int32)
    kv_base = torch.tensor([2, 0, 4], dtype=torch.int32)
    cos, sin = orc.rope_table(d, 1024, 10000.0, dtype)
    k_slab = torch.zeros(B, nKV, T_cap, d, dtype=dtype, device="cuda")
    v_slab = torch.zeros_like(k_slab)
    q_dev = qkv.cuda().clone()
    ops.rope_kv_write(q_dev, cos.cuda(), sin.cuda(), cu.cuda(), pos.cuda(), None, kv_base.cuda(), k_slab, v_slab, nH, nKV, d)
    out = q_dev.cpu()
    q = qkv[:, : nH * d].view(total, nH, d).transpose(0, 1)[None]
    k = qkv[:, nH * d : (nH + nKV) * d].view(total, nKV, d).transpose(0, 1)[None]
    v = qkv[:, (nH + nKV) * d :].view(total, nKV, d)
    qr, kr = orc.apply_rope(q, k, cos, sin, pos.long()[None])
    assert torch.equal(out[:, : nH * d], qr[0].transpose(0, 1).reshape(total, nH * d))
    assert torch.equal(out[:, nH * d : (nH + nKV) * d], kr[0].transpose(0, 1).reshape(total, nKV * d))
    assert torch.equal(out[:, (nH + nKV) * d :], qkv[:, (nH + nKV) * d :])
    ks, vs = k_slab.cpu(), v_slab.cpu()
    for b in range(B):
        for j in range(lens[b]):
            t = int(cu[b]) + j
            assert torch.equal(ks[b, :, int(kv_base[b]) + j], kr[0][:, t])
            assert torch.equal(vs[b, :, int(kv_base[b]) + j], v[t])
    # pos_base form (decode): position = pos_base[b] + j
    q2 = qkv.cuda().clone()
    base = torch.tensor([10, 20, 30], dtype=torch.int32)
    ops.rope_kv_write(q2, cos.cuda(), sin.cuda(), cu.cuda(), None, base.cuda(), kv_base.cuda(), k_slab, v_slab, nH, nKV, d)
    pos2 = torch.cat([base[b] + torch.arange(lens[b]) for b in range(B)]).long()
    qr2, _ = orc.apply_rope(q, k, cos, sin, pos2[None])
    assert torch.equal(q2.cpu()[:, : nH * d], qr2[0].transpose(0, 1).reshape(total, nH * d))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,n,k", [(1, 576, 115), (4, 576, 115), (3, 36, 7), (2, 1500, 300), (2, 64, 64), (1, 65, 1)])
def test_topk_select_bit_exact(ops, dtype, B, n, k):
    g = torch.Generator().manual_seed(5)
    s = torch.randn(B, n, generator=g)
    s = (s * 4).round() / 4  # few distinct values -> heavy ties, incl. at the k-th boundary
    s = F.log_softmax(torch.stack([s, -s], -1), -1)[..., 0].to(dtype)
    keep = ops.topk_select(s.cuda(), k).cpu()
    ref = orc.topk_keep_index(s, k, "stable")
    assert torch.equal(keep, ref)
    kth = torch.sort(s.float(), dim=1, descending=True).values[:, k - 1 : k]
    if 1 < k < n:
        assert int((s.float() == kth).sum()) > B, "test must exercise ties at the boundary"


def test_topk_select_no_ties_matches_reference_argsort(ops):
    g = torch.Generator().manual_seed(6)
    s = torch.randn(8, 576, generator=g)
    keep = ops.topk_select(s.cuda(), 115).cpu()
    assert torch.equal(keep, orc.topk_keep_index(s, 115, "torch"))  # the reference's own (non-stable) call


@pytest.mark.parametrize("dtype", DTYPES)
def test_compact_tokens_bit_exact(ops, dtype):
    g = torch.Generator().manual_seed(7)
    H, n_img, k = 512, 36, 7
    starts = [5, 1, 12]
    lens = [5 + 36 + 7, 1 + 36 + 1, 12 + 36 + 30]
    B = len(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    h = torch.randn(sum(lens), H, generator=g).to(dtype)
    keep = torch.stack([torch.sort(torch.randperm(n_img, generator=g)[:k]).values for _ in range(B)])
    new_lens = [n - (n_img - k) for n in lens]
    cu2 = torch.tensor([0] + list(torch.tensor(new_lens).cumsum(0)), dtype=torch.int32)
    out, pos = ops.compact_tokens(h.cuda(), keep.cuda(), cu.cuda(), cu2.cuda(), torch.tensor(starts, dtype=torch.int32).cuda(), n_img, k, sum(new_lens))
    for b in range(B):
        row = h[int(cu[b]) : int(cu[b + 1])]
        s = starts[b]
        ref = torch.cat([row[:s], row[s : s + n_img][keep[b]], row[s + n_img :]])
        ref_pos = torch.cat([torch.arange(0, s), keep[b] + s, torch.arange(s + n_img, lens[b])])  # DML:1963-1983
        assert torch.equal(out.cpu()[int(cu2[b]) : int(cu2[b + 1])], ref)
        assert torch.equal(pos.cpu()[int(cu2[b]) : int(cu2[b + 1])].long(), ref_pos)
    # fused RMSNorm of the compacted rows: bit-identical to compaction followed by dl_rmsnorm
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype).cuda()
    out2, pos2, x2 = ops.compact_tokens(h.cuda(), keep.cuda(), cu.cuda(), cu2.cuda(), torch.tensor(starts, dtype=torch.int32).cuda(), n_img, k, sum(new_lens), w, 1e-5)
    assert torch.equal(out2, out) and torch.equal(pos2, pos)
    assert torch.equal(x2, ops.rmsnorm(out, w, 1e-5))


def _sdpa_ref(q, k, v, causal):
    """q [T,nH,d], k/v [T,nKV,d] -> [T,nH,d] in fp32 from the (already rounded) inputs."""
    nH, nKV = q.shape[1], k.shape[1]
    qf, kf, vf = q.float().transpose(0, 1), k.float().transpose(0, 1), v.float().transpose(0, 1)
    if nKV != nH:
        kf = kf.repeat_interleave(nH // nKV, dim=0)
        vf = vf.repeat_interleave(nH // nKV, dim=0)
    return F.scaled_dot_product_attention(qf[None], kf[None], vf[None], is_causal=causal)[0].transpose(0, 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("nH,nKV,d,lens", [(4, 4, 128, [170, 1, 64, 65, 200]), (8, 8, 64, [576, 36]), (16, 16, 64, [577]), (4, 2, 64, [129, 300]), (4, 2, 128, [129]), (32, 32, 128, [170]), (4, 4, 128, [117, 65, 192, 3]), (4, 4, 128, [128, 66]), (4, 2, 128, [300, 631, 17, 257]), (2, 2, 32, [37, 150, 5]),
                                             # >= 256 (request, head) pairs, head_dim 128, causal, <= 256 rows: the whole-head kernel of a batched prefill's compacted layers (late round 6)
                                             (32, 32, 128, [170, 214, 158, 256, 65, 1, 16, 17, 33, 255]), (32, 8, 128, [200, 64, 129, 96, 31, 241, 2, 160]), (64, 64, 128, [97, 224, 5, 180]),
                                             # >= 256 (image, head) pairs, head_dim 64, non-causal, 257..608 rows: the 16-wave whole-head kernel of a batched CLIP tower (late round 6)
                                             (16, 16, 64, [577] * 14 + [300, 608, 257, 590]), (32, 8, 64, [577, 576, 290, 601, 333, 480, 259, 512])])
def test_attn_prefill(ops, dtype, causal, nH, nKV, d, lens):
    g = torch.Generator().manual_seed(8)
    total = sum(lens)
    W = (nH + 2 * nKV) * d
    qkv = torch.randn(total, W, generator=g).to(dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    dev = qkv.cuda()
    out = torch.full((total, nH * d), float("nan"), dtype=dtype, device="cuda")
    ops.attn_prefill(dev[:, : nH * d], dev[:, nH * d : (nH + nKV) * d], dev[:, (nH + nKV) * d :], out, cu.cuda(), max(lens), nH, nKV, d, causal)
    out = out.cpu().float().view(total, nH, d)
    assert torch.isfinite(out).all()
    for b in range(len(lens)):
        a, e = int(cu[b]), int(cu[b + 1])
        ref = _sdpa_ref(qkv[a:e, : nH * d].view(-1, nH, d), qkv[a:e, nH * d : (nH + nKV) * d].view(-1, nKV, d), qkv[a:e, (nH + nKV) * d :].view(-1, nKV, d), causal)
        # fp32 path: pure rounding noise.  16-bit paths: P is rounded to the 16-bit dtype before P@V (as flash /
        # the reference's SDPA backends do) and the output is rounded once: bound = 2 ulp of |max V| ~ 4.
        tol = 2e-5 if dtype == torch.float32 else 6 * ULP[dtype]
        err = float((out[a:e] - ref).abs().max())
        assert err < tol, f"row {b} len {lens[b]}: max err {err} > {tol}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_empty_inputs_are_no_ops(ops, dtype):
    """Empty inputs (zero rows / tokens; an empty torch tensor has a NULL data pointer): every row kernel, dl_linear, dl_rope_kv_write and
    dl_attn_prefill return DL_OK and touch nothing, like the eager ops they replace; a NEGATIVE count is still an argument error."""
    H, I, nH, d = 256, 512, 2, 128
    e = lambda *shape: torch.empty(*shape, dtype=dtype, device="cuda")
    w, b = torch.ones(H, dtype=dtype, device="cuda"), torch.zeros(H, dtype=dtype, device="cuda")
    assert ops.rmsnorm(e(0, H), w, 1e-5).shape == (0, H)
    assert ops.add_rmsnorm(e(0, H), e(0, H), w, 1e-5).shape == (0, H)
    assert ops.layernorm(e(0, H), w, b, 1e-5).shape == (0, H)
    assert ops.silu_mul(e(0, 2 * I)).shape == (0, I)
    assert ops.quick_gelu(e(0, H)).shape == (0, H)
    assert ops.linear(e(0, H), torch.ones(I, H, dtype=dtype, device="cuda")).shape == (0, I)
    cu = torch.zeros(2, dtype=torch.int32, device="cuda")
    zeros_b = torch.zeros(1, dtype=torch.int32, device="cuda")
    k_slab = torch.full((1, nH, 8, d), 7.0, dtype=dtype, device="cuda")
    v_slab = k_slab.clone()
    cos, sin = torch.ones(16, d, dtype=dtype, device="cuda"), torch.zeros(16, d, dtype=dtype, device="cuda")
    ops.rope_kv_write(e(0, 3 * nH * d), cos, sin, cu, None, zeros_b, zeros_b, k_slab, v_slab, nH, nH, d)
    assert bool((k_slab == 7.0).all()) and bool((v_slab == 7.0).all())
    qkv = e(0, 3 * nH * d)
    out = e(0, nH * d)
    ops.attn_prefill(qkv[:, : nH * d], qkv[:, nH * d : 2 * nH * d], qkv[:, 2 * nH * d :], out, cu, 0, nH, nH, d, True)
    torch.cuda.synchronize()
    rc = ops.lib().dl_rmsnorm(None, None, None, -1, H, 1e-5, ops.dtype_code(dtype), None)
    assert rc != 0 and b"bad shape" in ops.lib().dl_last_error()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_splits", [1, 8, 32])
def test_attn_at_max_context(ops, dtype, n_splits):
    """Maximum sizes: LLaVA-1.5's max_position_embeddings = 4096.  Decode attention over 4095 cached keys + the new one (next to a short and an
    EMPTY row of the same batch), and the causal prefill attention of one 4096-token row (the software-pipelined kernel's longest K/V loop),
    both against an fp32 SDPA of the same 16-bit inputs."""
    nH, nKV, d, T_cap = 4, 2, 128, 4096
    g = torch.Generator().manual_seed(23)
    kv_len = [4095, 0, 2049, 63]
    B = len(kv_len)
    k_slab = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    v_slab = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    q = torch.randn(B, nH * d, generator=g).to(dtype)
    out = torch.full((B, nH * d), float("nan"), dtype=dtype, device="cuda")
    ws = ops.attn_decode_workspace(B, nH, d, n_splits, "cuda")
    ops.attn_decode(q.cuda(), k_slab.cuda(), v_slab.cuda(), torch.tensor(kv_len, dtype=torch.int32).cuda(), 1, out, ws, n_splits, nH, nKV, d)
    out = out.cpu().float().view(B, nH, d)
    assert torch.isfinite(out).all()
    for b in range(B):
        T = kv_len[b] + 1
        ref = _sdpa_ref(q[b].view(1, nH, d), k_slab[b, :, :T].transpose(0, 1), v_slab[b, :, :T].transpose(0, 1), False)[0]
        err = float((out[b] - ref).abs().max())
        assert err < 3 * ULP[dtype], f"decode row {b} T={T}: max err {err}"
    if n_splits != 1:
        return  # the prefill half does not depend on the split factor
    L = 4096
    W = (nH + 2 * nKV) * d
    qkv = torch.randn(L, W, generator=g).to(dtype)
    cu = torch.tensor([0, L], dtype=torch.int32)
    dev = qkv.cuda()
    o = torch.full((L, nH * d), float("nan"), dtype=dtype, device="cuda")
    ops.attn_prefill(dev[:, : nH * d], dev[:, nH * d : (nH + nKV) * d], dev[:, (nH + nKV) * d :], o, cu.cuda(), L, nH, nKV, d, True)
    o = o.cpu().float().view(L, nH, d)
    ref = _sdpa_ref(qkv[:, : nH * d].view(-1, nH, d), qkv[:, nH * d : (nH + nKV) * d].view(-1, nKV, d), qkv[:, (nH + nKV) * d :].view(-1, nKV, d), True)
    assert torch.isfinite(o).all()
    err = float((o - ref).abs().max())
    assert err < 6 * ULP[dtype], f"prefill L={L}: max err {err}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nH,nKV,d", [(32, 32, 128), (8, 2, 128), (4, 4, 64)])
@pytest.mark.parametrize("n_splits", [1, 4, 32])
def test_attn_decode_ragged(ops, dtype, nH, nKV, d, n_splits):
    g = torch.Generator().manual_seed(9)
    kv_len = [0, 16, 170, 631, 65]  # + extra(1): the current token's slot
    B, T_cap = len(kv_len), 700
    k_slab = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    v_slab = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    q = torch.randn(B, nH * d, generator=g).to(dtype)
    out = torch.full((B, nH * d), float("nan"), dtype=dtype, device="cuda")
    ws = ops.attn_decode_workspace(B, nH, d, n_splits, "cuda")
    ops.attn_decode(q.cuda(), k_slab.cuda(), v_slab.cuda(), torch.tensor(kv_len, dtype=torch.int32).cuda(), 1, out, ws, n_splits, nH, nKV, d)
    out = out.cpu().float().view(B, nH, d)
    assert torch.isfinite(out).all()
    for b in range(B):
        T = kv_len[b] + 1
        ref = _sdpa_ref(q[b].view(1, nH, d), k_slab[b, :, :T].transpose(0, 1), v_slab[b, :, :T].transpose(0, 1), False)[0]
        tol = 2e-5 if dtype == torch.float32 else 3 * ULP[dtype]
        err = float((out[b] - ref).abs().max())
        assert err < tol, f"row {b} T={T}: max err {err} > {tol}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nH,nKV,d", [(32, 32, 128), (8, 2, 128), (4, 4, 64)])
@pytest.mark.parametrize("n_splits,kif,chunk", [(1, 64, 0), (3, 64, 0), (1, 256, 256), (3, 256, 256), (5, 256, 256), (2, 64, 100), (4, 256, 0)])
def test_attn_decode_rope_fused_equals_unfused(ops, dtype, nH, nKV, d, n_splits, kif, chunk):
    """dl_attn_decode_rope == dl_rope_kv_write followed by dl_attn_decode: identical slab contents (bit-exact RoPE / append),
    outputs equal up to the summation order of the online softmax.  Host-chunked splits (speculative K/V loads) must give the
    same answer whether the host's bound covers the row (5 x 256 >= 1024), undershoots it (1 or 3 x 256 < 1024: the last split
    takes the remainder) or leaves splits empty; slots past kv_len hold NaN bit patterns and must never leak."""
    g = torch.Generator().manual_seed(17)
    kv_len = [0, 16, 170, 631, 65, 1023]
    B, T_cap = len(kv_len), 1100
    pos = [5, 40, 631, 700, 65, 2000]  # RoPE position of the new token (the un-evicted count), != slot index after eviction
    k0 = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    v0 = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    for b, T in enumerate(kv_len):  # everything past the current length is garbage a correct kernel never uses
        k0[b, :, T:] = float("nan")
        v0[b, :, T:] = float("nan")
    qkv = torch.randn(B, (nH + 2 * nKV) * d, generator=g).to(dtype)
    cos, sin = orc.rope_table(d, 2048, 10000.0, dtype)
    lens = torch.tensor(kv_len, dtype=torch.int32).cuda()
    posd = torch.tensor(pos, dtype=torch.int32).cuda()
    # un-fused reference path
    ka, va, qa = k0.cuda().clone(), v0.cuda().clone(), qkv.cuda().clone()
    cu = torch.arange(0, B + 1, dtype=torch.int32).cuda()
    ops.rope_kv_write(qa, cos.cuda(), sin.cuda(), cu, None, posd, lens, ka, va, nH, nKV, d)
    out_a = torch.empty(B, nH * d, dtype=dtype, device="cuda")
    ws = ops.attn_decode_workspace(B, nH, d, 8, "cuda")
    ops.attn_decode(qa[:, : nH * d], ka, va, lens, 1, out_a, ws, 4, nH, nKV, d)
    # fused path
    kb, vb, qb = k0.cuda().clone(), v0.cuda().clone(), qkv.cuda().clone()
    out_b = torch.full((B, nH * d), float("nan"), dtype=dtype, device="cuda")
    ops.attn_decode_rope(qb, cos.cuda(), sin.cuda(), posd, lens, kb, vb, out_b, ws, n_splits, nH, nKV, d, keys_in_flight=kif, chunk_keys=chunk)
    assert torch.equal(qb.cpu(), qkv), "the fused kernel must not modify qkv"
    assert torch.equal(ka.nan_to_num(7.0), kb.nan_to_num(7.0)) and torch.equal(va.nan_to_num(7.0), vb.nan_to_num(7.0)), "slab contents (rotated key / value at slot kv_len[b]) must be bit-identical"
    tol = 2e-5 if dtype == torch.float32 else 2 * ULP[dtype]
    assert float((out_a.float() - out_b.float()).abs().max()) < tol
    for b in range(B):  # and against an fp32 SDPA evaluation of the same rounded operands
        T = kv_len[b] + 1
        ref = _sdpa_ref(qa[b].cpu()[: nH * d].view(1, nH, d), ka[b, :, :T].cpu().transpose(0, 1), va[b, :, :T].cpu().transpose(0, 1), False)[0]
        err = float((out_b[b].cpu().float().view(nH, d) - ref).abs().max())
        assert err < (2e-5 if dtype == torch.float32 else 3 * ULP[dtype]), (b, err)
    # in-kernel combine (call_tag >= 0; production shape only: keys_in_flight 64, kernel-balanced splits): split 0's workgroup merges the
    # granule partials itself -- bit-identical to the two-launch form, also when the same workspace is reused call after call
    if kif == 64 and chunk == 0 and n_splits > 1:
        for rep, tag in enumerate([0, 1, 2, 0, 31]):
            kc, vc = k0.cuda().clone(), v0.cuda().clone()
            out_c = torch.full((B, nH * d), float("nan"), dtype=dtype, device="cuda")
            ops.attn_decode_rope(qb, cos.cuda(), sin.cuda(), posd, lens, kc, vc, out_c, ws, n_splits, nH, nKV, d, call_tag=tag)
            assert torch.equal(out_c, out_b), (rep, tag)
            assert torch.equal(kc.nan_to_num(7.0), kb.nan_to_num(7.0)) and torch.equal(vc.nan_to_num(7.0), vb.nan_to_num(7.0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nH,nKV,d", [(32, 32, 128), (8, 2, 128), (4, 4, 64)])
@pytest.mark.parametrize("n_parts,n_splits,tag", [(2, 1, -1), (2, 3, -1), (2, 3, 5), (1, 1, -1), (4, 2, 3)])
def test_attn_decode_rope_parts_equals_rounded_sum(ops, dtype, nH, nKV, d, n_parts, n_splits, tag):
    """dl_attn_decode_rope_parts (q|k|v handed over as the projection's fp32 k-range partial sums, round 6) == dl_attn_decode_rope on the sum of the ranges,
    added in range order and rounded once to the cache dtype: output and slab contents bit-identical, for one split, several splits with the separate merge
    launch and with the in-kernel merge.  The ranges live `part_stride` > B rows apart (the decode state's buffer is sized for two ranges of B rows)."""
    g = torch.Generator().manual_seed(23)
    kv_len = [0, 16, 170, 631, 65, 1023, 300]
    B, T_cap = len(kv_len), 1100
    pos = [5, 40, 631, 700, 65, 2000, 301]
    N = (nH + 2 * nKV) * d
    k0 = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    v0 = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    for b, T in enumerate(kv_len):
        k0[b, :, T:] = float("nan")
        v0[b, :, T:] = float("nan")
    buf = torch.full((n_parts, B + 3, N), float("nan"))  # rows past B: never read
    buf[:, :B] = torch.randn(n_parts, B, N, generator=g) / math.sqrt(n_parts)
    total = buf[0, :B].clone()
    for r in range(1, n_parts):
        total = total + buf[r, :B]  # fp32, range order
    qkv = total.to(dtype)
    cos, sin = orc.rope_table(d, 2048, 10000.0, dtype)
    lens = torch.tensor(kv_len, dtype=torch.int32).cuda()
    posd = torch.tensor(pos, dtype=torch.int32).cuda()
    ws = ops.attn_decode_workspace(B, nH, d, 8, "cuda")
    ka, va = k0.cuda().clone(), v0.cuda().clone()
    out_a = torch.full((B, nH * d), float("nan"), dtype=dtype, device="cuda")
    ops.attn_decode_rope(qkv.cuda(), cos.cuda(), sin.cuda(), posd, lens, ka, va, out_a, ws, n_splits, nH, nKV, d, call_tag=tag)
    kb, vb = k0.cuda().clone(), v0.cuda().clone()
    out_b = torch.full((B, nH * d), float("nan"), dtype=dtype, device="cuda")
    parts = buf.cuda()[:, :B]  # [n_parts, B, N] view: stride(0) = (B + 3) N
    if tag >= 0:
        ws.zero_()
    ops.attn_decode_rope_parts(parts, cos.cuda(), sin.cuda(), posd, lens, kb, vb, out_b, ws, n_splits, nH, nKV, d, call_tag=tag)
    assert torch.isfinite(out_b.float()).all()
    assert torch.equal(out_a, out_b)
    assert torch.equal(ka.nan_to_num(7.0), kb.nan_to_num(7.0)) and torch.equal(va.nan_to_num(7.0), vb.nan_to_num(7.0))


def test_attn_decode_rope_parts_bad_args(ops):
    d, nH = 128, 4
    cos, sin = orc.rope_table(d, 64, 10000.0, torch.bfloat16)
    lens = torch.zeros(2, dtype=torch.int32, device="cuda")
    k = torch.zeros(2, nH, 8, d, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(2, nH * d, dtype=torch.bfloat16, device="cuda")
    parts = torch.zeros(3, 2, 3 * nH * d, device="cuda")
    with pytest.raises(ops.HipOpsError, match="n_parts"):
        ops.attn_decode_rope_parts(parts, cos.cuda(), sin.cuda(), lens, lens, k, k.clone(), out, None, 1, nH, nH, d)
    c32, s32 = orc.rope_table(d, 64, 10000.0, torch.float32)
    with pytest.raises(ops.HipOpsError, match="bf16 / fp16"):
        ops.attn_decode_rope_parts(parts[:2], c32.cuda(), s32.cuda(), lens, lens, k.float(), k.float(), out.float(), None, 1, nH, nH, d)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(576, 512, 4096), (36, 128, 256), (100, 1536, 512), (70, 32, 64), (1, 2048, 512)])
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_linear_epilogues(ops, dtype, M, N, K, flags):
    g = torch.Generator().manual_seed(10)
    a = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    bias = (0.1 * torch.randn(N, generator=g)).to(dtype)
    r = torch.randn(M, N, generator=g).to(dtype)
    ref = F.linear(a.float(), w.float(), bias.float()).to(dtype)
    if flags & 1:
        ref = F.gelu(ref.float()).to(dtype)
    if flags & 2:
        ref = (r.float() + ref.float()).to(dtype)
    rd = r.cuda().clone()
    out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), flags, residual=rd if flags & 2 else None, out=rd if flags & 2 else None)
    # fp32 accumulate in a different order than the fp32 reference: <= 1 ulp of the output dtype after rounding
    # (the residual epilogue rounds twice, so a 1-ulp flip before the add can surface as 2 ulp after it)
    _close_ulp(out, ref, dtype, 2.0 if flags & 2 else 1.0, atol=1e-4 if dtype == torch.float32 else 1e-3, mag=r if flags & 2 else None)
    assert _frac_exact(out, ref, dtype) > 0.97


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B", [1, 3])
@pytest.mark.parametrize("N,K", [(4096, 4096), (22016, 4096), (4096, 11008), (1000, 256), (37, 512),
                                 (15360, 5120), (5120, 5120), (27648, 5120), (5120, 13824),  # these four: LLaVA-1.5-13B (configs[4])
                                 (33, 12296), (70, 16384), (12, 16392), (9, 8200), (21, 12288)])  # a row per wave pair (8192 < K <= 16384): ragged halves, odd N, largest K; beyond: generic kernel
def test_gemv_modes(ops, dtype, B, N, K):
    """dl_gemv against the eager op sequence it replaces (fp32 evaluation of the same rounded operands)."""
    g = torch.Generator().manual_seed(16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    wd = w.cuda()
    if B > ops.gemv_max_batch(K, dtype):  # x rows must fit LDS (fp32 at 13B width: 2 rows): the ABI refuses, callers use the GEMM path
        with pytest.raises(ops.HipOpsError):
            ops.gemv(wd, torch.empty(B, N, dtype=dtype, device="cuda"), x=torch.zeros(B, K, dtype=dtype, device="cuda"))
        return
    # PLAIN
    x = torch.randn(B, K, generator=g).to(dtype)
    y = torch.empty(B, N, dtype=dtype, device="cuda")
    ops.gemv(wd, y, x=x.cuda())
    ref = F.linear(x.float(), w.float()).to(dtype)
    _close_ulp(y, ref, dtype, 1.0, atol=1e-4 if dtype == torch.float32 else 2e-3)
    # ADDNORM (+ residual write-back to a distinct buffer)
    h = torch.randn(B, K, generator=g).to(dtype)
    dl = torch.randn(B, K, generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(dtype)
    h_out = torch.zeros(B, K, dtype=dtype, device="cuda")
    ops.gemv(wd, y, mode=ops.GEMV_ADDNORM, h_in=h.cuda(), h_out=h_out, delta=dl.cuda(), norm_w=nw.cuda(), eps=1e-5)
    hn = h + dl
    assert torch.equal(h_out.cpu(), hn)
    ref = F.linear(orc.rmsnorm(hn, nw, 1e-5).float(), w.float()).to(dtype)
    _close_ulp(y, ref, dtype, 2.0, atol=2e-4 if dtype == torch.float32 else 2e-2)  # a 1-ulp flip in x moves a K-term dot by ~ulp*|w||x|
    ops.gemv(wd, y, mode=ops.GEMV_ADDNORM, h_in=h.cuda(), h_out=None, delta=None, norm_w=nw.cuda(), eps=1e-5)
    ref = F.linear(orc.rmsnorm(h, nw, 1e-5).float(), w.float()).to(dtype)
    _close_ulp(y, ref, dtype, 2.0, atol=2e-4 if dtype == torch.float32 else 2e-2)
    # SILUMUL
    gu = torch.randn(B, 2 * K, generator=g).to(dtype)
    ops.gemv(wd, y, x=gu.cuda(), mode=ops.GEMV_SILUMUL)
    act = F.silu(gu[:, :K]) * gu[:, K:]
    ref = F.linear(act.float(), w.float()).to(dtype)
    _close_ulp(y, ref, dtype, 2.0, atol=2e-4 if dtype == torch.float32 else 2e-2)
    with pytest.raises(ops.HipOpsError):
        ops.gemv(wd, y, mode=ops.GEMV_ADDNORM, h_in=h_out, h_out=h_out, delta=dl.cuda(), norm_w=nw.cuda())  # in-place residual is a race
    # SILU_PAIR epilogue on a fused gate|up weight: act = silu(W[:I] x) * (W[I:] x), two workgroup caps
    if N % 2 == 0:
        I = N // 2
        gu_ref = F.linear(x.float(), w.float()).to(dtype)
        ref = F.silu(gu_ref[:, :I]) * gu_ref[:, I:]
        for cap in (512, 64):  # per-call workgroup cap (no process-global tuning state in the ABI)
            act = torch.empty(B, I, dtype=dtype, device="cuda")
            ops.gemv(wd, act, x=x.cuda(), mode=ops.GEMV_OUT_SILU_PAIR, grid_cap=cap)
            _close_ulp(act, ref, dtype, 4.0, atol=2e-4 if dtype == torch.float32 else 2e-2)
            ops.gemv(wd, y, x=x.cuda(), grid_cap=cap)
            _close_ulp(y, gu_ref, dtype, 1.0, atol=1e-4 if dtype == torch.float32 else 2e-3)


def _vp_sd(cfg, seed, gain):
    sd = fx.make_state_dict(cfg, seed=seed, predictor_gain=gain, with_projector=False)
    return {k: v for k, v in sd.items() if k.startswith("model.image_score_predictor.") or k.startswith("model.output_text_score_predictor.")}


def _tiny_pred_cfg(H, D, nhead, FF):
    return fx.make_config(hidden_size=H, intermediate_size=64, num_hidden_layers=0, num_attention_heads=H // 128, vocab_size=8, d_model=D, nhead=nhead, dim_feedforward=FF)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,D,nhead,FF,n_img,B", [(4096, 512, 8, 2048, 576, 2), (256, 128, 2, 256, 36, 3)])
def test_vision_predictor_vs_oracle(ops, dtype, H, D, nhead, FF, n_img, B):
    from dynamic_llava_amd.model import VisionPredictor

    cfg = _tiny_pred_cfg(H, D, nhead, FF)
    sd = _vp_sd(cfg, 11, 50.0)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, n_img, H, generator=g).to(dtype)
    sdt = {k: v.to(dtype) for k, v in sd.items()}
    ref_logit = orc.vision_predictor(sdt, "model.image_score_predictor.", x, torch.ones(B, n_img, 1, dtype=dtype), nhead, 2)
    ref32 = orc.vision_predictor({k: v.to(dtype).float() for k, v in sd.items()}, "model.image_score_predictor.", x.float(), torch.ones(B, n_img, 1), nhead, 2)
    vp = VisionPredictor(H, D, nhead, FF, 2)
    vp.load_state_dict({k[len("model.image_score_predictor.") :]: v for k, v in sd.items() if "image_score" in k})
    vp = vp.to(device="cuda", dtype=dtype)
    # packed form with leading/trailing text rows, as the model calls it
    pad_a, pad_b = 3, 5
    rows = torch.cat([torch.cat([torch.randn(pad_a, H, generator=g).to(dtype), x[b], torch.randn(pad_b, H, generator=g).to(dtype)]) for b in range(B)])
    L = pad_a + n_img + pad_b
    cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32).cuda()
    logits, score = vp.score_packed(rows.cuda(), cu, torch.full((B,), pad_a, dtype=torch.int32).cuda(), n_img)
    logits2 = vp(x.cuda())  # dense / hookable form must agree bit-for-bit
    assert torch.equal(logits, logits2)
    err = float((logits.float().cpu() - ref32).abs().max())
    noise = float((ref_logit.float() - ref32).abs().max())
    scale = float(ref32.abs().max())
    if dtype == torch.float32:
        assert err < 2e-4 * max(1.0, scale), (err, scale)
    else:
        # same noise class as the eager reference evaluated in the same dtype (vs. fp32 ground truth)
        assert err <= 2.0 * noise + 4 * ULP[dtype] * scale, (err, noise, scale)
    ref_score = F.log_softmax(logits.float().cpu(), dim=-1)[..., 0].to(dtype)
    _close_ulp(score, ref_score, dtype, 1.0, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,D,B", [(4096, 512, 1), (4096, 512, 32), (256, 128, 3), (5120, 512, 2)])
def test_text_predictor_vs_oracle(ops, dtype, H, D, B):
    from dynamic_llava_amd.model import TextPredictor

    cfg = _tiny_pred_cfg(H, D, 2, 256)
    sd = _vp_sd(cfg, 13, 50.0)
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, H, generator=g).to(dtype)
    ref = orc.text_predictor({k: v.to(dtype) for k, v in sd.items()}, "model.output_text_score_predictor.", x[:, None])[:, 0]
    ref32 = orc.text_predictor({k: v.to(dtype).float() for k, v in sd.items()}, "model.output_text_score_predictor.", x.float()[:, None])[:, 0]
    tp = TextPredictor(H, D)
    tp.load_state_dict({k[len("model.output_text_score_predictor.") :]: v for k, v in sd.items() if "output_text" in k})
    tp = tp.to(device="cuda", dtype=dtype)
    ws = ops.text_predictor_workspace(B, D, "cuda")
    lg = torch.empty(B, 2, dtype=torch.float32, device="cuda")
    dec = torch.empty(B, dtype=torch.int32, device="cuda")
    tp.decide(x.cuda(), ws, lg, dec)
    lg, dec = lg.cpu(), dec.cpu()
    err = float((lg - ref32).abs().max())
    noise = float((ref.float() - ref32).abs().max())
    scale = float(ref32.abs().max())
    if dtype == torch.float32:
        assert err < 2e-4 * max(1.0, scale)
    else:
        assert err <= 2.0 * noise + 4 * ULP[dtype] * scale, (err, noise, scale)
    assert torch.equal(dec.bool(), lg[:, 0] > lg[:, 1])
    gap = (ref32[:, 0] - ref32[:, 1]).abs()
    sure = gap > 4 * (err + noise)
    assert torch.equal(dec.bool()[sure], (ref[:, 0] > ref[:, 1])[sure]), "decision differs from the oracle away from the boundary"


@pytest.mark.parametrize("ldt", [torch.float32, torch.bfloat16])
def test_decode_advance(ops, ldt):
    g = torch.Generator().manual_seed(15)
    B, V = 5, 32000
    logits = torch.randn(B, V, generator=g).to(ldt)
    logits[1, 777] = logits[1, 12345] = 50.0  # tie -> lowest index
    logits[3, 2] = 60.0  # EOS
    nxt = torch.zeros(B, dtype=torch.int64, device="cuda")
    out = torch.zeros(B, 4, dtype=torch.int64, device="cuda")
    step = torch.tensor([0, 1, 2, 3, 0], dtype=torch.int32, device="cuda")
    fin = torch.tensor([0, 0, 0, 0, 1], dtype=torch.int32, device="cuda")
    lf = torch.tensor([10, 20, 30, 40, 50], dtype=torch.int32, device="cuda")
    ls = torch.tensor([5, 6, 7, 8, 9], dtype=torch.int32, device="cuda")
    dec = torch.tensor([1, 0, 1, 0, 1], dtype=torch.int32, device="cuda")
    ops.decode_advance(logits.cuda(), nxt, out, step, fin, 2, 0, lf, ls, dec)
    ref = logits.float().argmax(-1)
    assert nxt.cpu().tolist() == [int(ref[0]), 777, int(ref[2]), 2, 0]
    assert fin.cpu().tolist() == [0, 0, 0, 1, 1]
    assert step.cpu().tolist() == [1, 2, 3, 4, 1]
    assert lf.cpu().tolist() == [11, 21, 31, 41, 51] and ls.cpu().tolist() == [6, 6, 8, 8, 10]
    o = out.cpu()
    assert o[0, 0] == ref[0] and o[1, 1] == 777 and o[2, 2] == ref[2] and o[3, 3] == 2 and o[4, 0] == 0


def test_c_abi_rejects_bad_arguments(ops):
    lib = ops.lib()
    assert lib.dl_rmsnorm(None, None, None, 1, 4096, 1e-5, 2, None) == -1
    assert b"NULL" in lib.dl_last_error()
    x = torch.zeros(4, 100, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ops.HipOpsError):
        ops.rmsnorm(x, torch.ones(100, dtype=torch.bfloat16, device="cuda"), 1e-5)  # H % 8 != 0
    with pytest.raises(ops.HipOpsError):
        ops.rmsnorm(torch.zeros(4, 128), torch.ones(128), 1e-5)  # CPU tensor: no fallback


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nH,nKV,d", [(4, 4, 128), (4, 2, 64)])
@pytest.mark.parametrize("Lq", [[70, 5, 64], [300, 5, 270]])  # short chunks: plain kernel; > 256 queries: software-pipelined kernel
def test_attn_prefill_cached_chunk_on_slab(ops, dtype, nH, nKV, d, Lq):
    """Chunk of queries against the KV slab: query j of row b sees keys [0, kv_len[b] + j]."""
    g = torch.Generator().manual_seed(18)
    kv_len = [0, 37, 200]
    B, T_cap = 3, 600
    k_slab = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    v_slab = torch.randn(B, nKV, T_cap, d, generator=g).to(dtype)
    total = sum(Lq)
    q = torch.randn(total, nH * d, generator=g).to(dtype)
    cu = torch.tensor([0] + list(torch.tensor(Lq).cumsum(0)), dtype=torch.int32)
    out = torch.full((total, nH * d), float("nan"), dtype=dtype, device="cuda")
    ops.attn_prefill_cached(q.cuda(), k_slab.cuda(), v_slab.cuda(), torch.tensor(kv_len, dtype=torch.int32).cuda(), out, cu.cuda(), max(Lq),
                            max(a + b for a, b in zip(kv_len, Lq)), nH, nKV, d)
    out = out.cpu().float().view(total, nH, d)
    assert torch.isfinite(out).all()
    for b in range(B):
        a, e = int(cu[b]), int(cu[b + 1])
        Lk = kv_len[b] + Lq[b]
        qf = q[a:e].float().view(Lq[b], nH, d).transpose(0, 1)
        kf = k_slab[b, :, :Lk].float().repeat_interleave(nH // nKV, dim=0)
        vf = v_slab[b, :, :Lk].float().repeat_interleave(nH // nKV, dim=0)
        mask = torch.arange(Lk)[None, :] <= (torch.arange(Lq[b])[:, None] + kv_len[b])
        ref = F.scaled_dot_product_attention(qf[None], kf[None], vf[None], attn_mask=mask)[0].transpose(0, 1)
        tol = 2e-5 if dtype == torch.float32 else 6 * ULP[dtype]
        assert float((out[a:e] - ref).abs().max()) < tol, f"row {b}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize(
    "M,N,K,n_slices,wg_waves",
    [(32, 4096, 4096, 0, 0), (16, 12288, 4096, 1, 4), (5, 22016, 4096, 0, 8), (8, 4096, 11008, 0, 0), (32, 4096, 11008, 0, 4), (17, 200, 512, 2, 0),
     (1, 64, 256, 0, 0), (9, 132, 768, 3, 8), (32, 32000, 4096, 0, 0), (24, 1000, 1280, 1, 0),
     (8, 15360, 5120, 0, 0), (16, 27648, 5120, 0, 0), (6, 5120, 13824, 0, 0), (32, 5120, 5120, 0, 0), (32, 32000, 5120, 0, 0)],  # last five: LLaVA-1.5-13B projections (32 rows: o_proj, lm_head)
)
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_gemm_smallm(ops, dtype, M, N, K, n_slices, wg_waves, variant):
    """Small-batch decode GEMM on the matrix cores == F.linear with fp32 accumulation and one rounding: both batch-tile counts, ragged M / N
    (tiles beyond N clamp, rows beyond M are zero), uneven K slices, LDS-forced slicing (32 x 11008), strided X / Y, deterministic."""
    g = torch.Generator().manual_seed(43)
    x_full = torch.randn(M, K + 64, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    ref = F.linear(x_full[:, :K].float(), w.float()).to(dtype)
    y_full = torch.full((M, N + 8), float("nan"), dtype=dtype, device="cuda")
    out = ops.gemm_smallm(x_full.cuda()[:, :K], w.cuda(), out=y_full[:, :N], n_slices=n_slices, wg_waves=wg_waves, variant=variant)
    assert torch.isnan(y_full[:, N:]).all(), "must not write outside [M, N]"
    _close_ulp(out, ref, dtype, 1.0, atol=1e-3)
    assert _frac_exact(out, ref, dtype) > 0.97
    out2 = ops.gemm_smallm(x_full.cuda()[:, :K], w.cuda(), n_slices=n_slices, wg_waves=wg_waves, variant=variant)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,I", [(6, 4096, 11008), (16, 1024, 2816), (1, 512, 1024)])
def test_smallm_partials_consumers_bit_equal_reduce_then_op(ops, dtype, M, H, I):
    """dl_gemm_smallm(defer_reduce) + dl_add_rmsnorm_parts / dl_silu_mul_parts == dl_gemm_smallm (reduce launch) + dl_add_rmsnorm /
    dl_silu_mul, bit for bit: the consumers add the split-K partials in the same slice order and round at the same points."""
    g = torch.Generator().manual_seed(61)
    x = torch.randn(M, H, generator=g).to(dtype).cuda()
    act = torch.randn(M, I, generator=g).to(dtype).cuda()
    w_gu = (torch.randn(2 * I, H, generator=g) / math.sqrt(H)).to(dtype).cuda()
    w_dn = (torch.randn(H, I, generator=g) / math.sqrt(I)).to(dtype).cuda()
    h0 = torch.randn(M, H, generator=g).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype).cuda()
    ws = torch.empty(8 * M * 2 * I, dtype=torch.float32, device="cuda")
    # gate|up -> SiLU*up
    ref = ops.silu_mul(ops.gemm_smallm(x, w_gu, workspace=ws), out=torch.empty(M, I, dtype=dtype, device="cuda"))
    parts, s = ops.gemm_smallm_parts(x, w_gu, ws)
    assert parts.shape == (s, M, 2 * I)
    out = ops.silu_mul_parts(parts, torch.empty(M, I, dtype=dtype, device="cuda"))
    assert torch.equal(out, ref)
    # down -> residual add + RMSNorm
    dn = ops.gemm_smallm(act, w_dn, workspace=ws)
    h_a = h0.clone()
    x_a = ops.add_rmsnorm(h_a, dn, nw, 1e-5)
    parts, s = ops.gemm_smallm_parts(act, w_dn, ws)
    h_b = h0.clone()
    x_b = ops.add_rmsnorm_parts(h_b, parts, nw, 1e-5)
    assert torch.equal(h_a, h_b) and torch.equal(x_a, x_b)
    h_c = h0.clone()
    assert ops.add_rmsnorm_parts(h_c, parts) is None and torch.equal(h_c, h_a)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,nKV,d,T,n_layers", [(1, 2, 128, 9, 3), (3, 4, 128, 40, 2), (2, 2, 64, 17, 1)])
def test_kv_pack_rows(ops, dtype, B, nKV, d, T, n_layers):
    """In-place packing of the kept rows of an appended chunk for several layer slabs at once == slicing / concatenating per row (CU:165-241
    without the zero padding); rows outside [kv_len, kv_len + T) and other layers untouched."""
    g = torch.Generator().manual_seed(50)
    L_all, T_cap = n_layers + 2, 96
    slab = torch.randn((L_all, 2, B, nKV, T_cap, d), generator=g).to(dtype)
    kv_len = torch.randint(3, 40, (B,), generator=g).to(torch.int32)
    keep = (torch.rand((B, T), generator=g) > 0.4).to(torch.int32)
    keep[:, -1] = 1
    keep[0, :] = 1 if B > 1 else keep[0, :]
    ref = slab.clone()
    for l in range(1, 1 + n_layers):
        for kv in range(2):
            for b in range(B):
                n0 = int(kv_len[b])
                rows = ref[l, kv, b, :, n0 : n0 + T][:, keep[b].bool()]
                ref[l, kv, b, :, n0 : n0 + rows.shape[1]] = rows
    sd = slab.cuda()
    ops.kv_pack_rows(sd[1, 0], sd[1, 1], sd.stride(0), n_layers, keep.cuda().contiguous(), kv_len.cuda(), T_cap)
    got = sd.cpu()
    for l in range(L_all):
        for kv in range(2):
            for b in range(B):
                n0, nk = int(kv_len[b]), int(keep[b].sum())
                if 1 <= l <= n_layers:
                    assert torch.equal(got[l, kv, b, :, : n0 + nk], ref[l, kv, b, :, : n0 + nk]), (l, kv, b)
                    assert torch.equal(got[l, kv, b, :, n0 + T :], slab[l, kv, b, :, n0 + T :])
                else:
                    assert torch.equal(got[l, kv, b], slab[l, kv, b])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("total,span0,n_span,H", [(170, 150, 19, 4096), (60, 0, 60, 256), (33, 10, 0, 128), (700, 640, 59, 512), (41, 40, 1, 64)])
def test_compact_rows_by_mask(ops, dtype, total, span0, n_span, H):
    """dl_compact_rows_by_mask == nonzero + index_select of the eager instruct-predictor compaction (DML:2261-2375): rows, positions, the
    device-side counts; all-kept, all-dropped and empty spans."""
    g = torch.Generator().manual_seed(total)
    h = torch.randn(total, H, generator=g).to(dtype).cuda()
    pos = torch.randperm(2 * total, generator=g)[:total].to(torch.int32).cuda()
    for case in ("random", "all", "none"):
        dec = {"random": torch.randint(0, 2, (max(n_span, 1),), generator=g), "all": torch.ones(max(n_span, 1)), "none": torch.zeros(max(n_span, 1))}[case].to(torch.int32)[:n_span].cuda()
        idx = torch.cat([torch.arange(0, span0), torch.nonzero(dec.cpu()).flatten() + span0, torch.arange(span0 + n_span, total)]).cuda()
        for use_pos in (True, False):
            dec_arg = dec if n_span > 0 else torch.zeros(1, dtype=torch.int32, device="cuda")
            h_out, p_out, cu, counts = ops.compact_rows_by_mask(h, pos if use_pos else None, dec_arg, span0, n_span)
            n = int(idx.numel())
            assert counts.tolist() == [n, n - 1] and cu.tolist() == [0, n]
            assert torch.equal(h_out[:n], h.index_select(0, idx)) and not bool(h_out[n:].any())
            assert torch.equal(p_out[:n], (pos if use_pos else torch.arange(total, dtype=torch.int32, device="cuda")).index_select(0, idx))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,s", [(170, 4096, 11008, 8), (117, 4096, 4096, 4), (1, 256, 1024, 2), (192, 320, 1088, 8), (200, 512, 2048, 4), (631, 4096, 1024, 1), (33, 256, 128, 1), (40, 5120, 1024, 4), (9, 6144, 512, 3)])
def test_linear_splitk(ops, dtype, M, N, K, s):
    """Split-K projection: the slices summed in order == F.linear with fp32 accumulation (both tilings: all rows in one tile up to 192 rows,
    64x64 beyond), ragged M / N / K tails, strided A, deterministic; with dl_add_rmsnorm_parts == library GEMM + dl_add_rmsnorm up to rounding."""
    g = torch.Generator().manual_seed(77)
    a_full = torch.randn(M, K + 24, generator=g).to(dtype).cuda()
    a = a_full[:, :K]
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).cuda()
    ws = torch.full((s * M * N + 16,), float("nan"), dtype=torch.float32, device="cuda")
    parts = ops.linear_splitk(a, w, ws, s)
    assert parts.shape == (s, M, N) and torch.isnan(ws[s * M * N :]).all()
    ref = F.linear(a.float(), w.float())
    got = parts.sum(0)
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())) * math.sqrt(K / 1024 + 1)
    assert torch.equal(parts, ops.linear_splitk(a, w, torch.empty_like(ws), s))
    h0 = torch.randn(M, N, generator=g).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).cuda()
    h1, h2 = h0.clone(), h0.clone()
    x1 = ops.add_rmsnorm_parts(h1, parts.contiguous(), nw, 1e-5)
    x2 = ops.add_rmsnorm(h2, ref.to(dtype), nw, 1e-5)
    assert torch.equal(h1, h2) or float((h1.float() - h2.float()).abs().max()) <= 2 * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * float(h2.float().abs().max())
    _close_ulp(x1, x2, dtype, 2.0, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nH,nKV,d,H,T_old,with_delta", [(32, 32, 128, 4096, 199, True), (40, 40, 128, 5120, 0, True), (32, 32, 128, 4096, 255, False), (8, 4, 64, 1024, 37, True), (32, 32, 128, 4096, 700, True)])
def test_gemv_qkv_attn_bit_identical_to_the_two_launches(ops, dtype, nH, nKV, d, H, T_old, with_delta):
    """dl_gemv_qkv_attn (q|k|v projection with the add+RMSNorm prologue AND the single-split decode attention in one launch, the projection's
    outputs handed to the attention workgroups as granules) against dl_gemv(ADDNORM) + dl_attn_decode_rope(n_splits=1): projection row,
    residual stream and the appended K/V row bit for bit; repeated over steps / call tags on the same granule buffer (no stale granule may be
    consumed), empty cache, GQA, head_dim 64, rows longer than one trip.  The attention OUTPUT (round 4): the fused launch folds the new token in
    AFTER the merge of the slab keys' partials (attn_split_finish_newlast: the merge leaves the launch's tail), the stand-alone kernel before it
    -- the same sum in another order, so it is held to the rounding class (<= 2 ulp of the row's largest value, and against an fp32 reference of
    the attention itself) instead of to the bit; repeatability of the fused launch stays bit-exact."""
    from oracle.ref_cpu import rope_table

    g = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *shape, s=0.02: (torch.randn(*shape, device="cuda", generator=g) * s).to(dtype)
    N = (nH + 2 * nKV) * d
    W, nw = rnd(N, H), 1 + rnd(H, s=0.1)
    T_cap = T_old + 40
    cos, sin = (t.cuda() for t in rope_table(d, T_cap + 8, 10000.0, dtype))
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    gran = ops.gemv_qkv_attn_workspace(nH, nKV, d, "cuda")
    k0, v0 = rnd(1, nKV, T_cap, d, s=1.0), rnd(1, nKV, T_cap, d, s=1.0)
    eps = 1e-5
    for step in range(6):
        h0, delta = rnd(1, H, s=1.0), (rnd(1, H, s=1.0) if with_delta else None)
        lens = torch.tensor([T_old + step], dtype=torch.int32, device="cuda")
        pos = torch.tensor([T_old + step + 3], dtype=torch.int32, device="cuda")
        # reference: two launches
        k_r, v_r = k0.clone(), v0.clone()
        qkv_r, ho_r, out_r = torch.zeros(1, N, dtype=dtype, device="cuda"), torch.zeros(1, H, dtype=dtype, device="cuda"), torch.zeros(1, nH * d, dtype=dtype, device="cuda")
        ops.gemv(W, qkv_r, mode=ops.GEMV_ADDNORM, h_in=h0, h_out=ho_r, delta=delta, norm_w=nw, eps=eps)
        ops.attn_decode_rope(qkv_r, cos, sin, pos, lens, k_r, v_r, out_r, None, 1, nH, nKV, d, chunk_keys=256)
        for tag in (step & 0xff, 200 + step):  # the same step under another call tag as well
            k_f, v_f = k0.clone(), v0.clone()
            qkv_f, ho_f, out_f = torch.zeros_like(qkv_r), torch.zeros_like(ho_r), torch.zeros_like(out_r)
            ops.gemv_qkv_attn(W, qkv_f, h0, ho_f, delta, nw, eps, cos, sin, pos, lens, k_f, v_f, out_f, gran, tag, nH, nKV, d, err=err)
            assert torch.equal(qkv_f, qkv_r), (step, tag)
            ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
            diff = float((out_f.float() - out_r.float()).abs().max())
            assert diff <= 2 * ulp * float(out_r.float().abs().max()), (step, tag, diff)
            assert torch.equal(k_f, k_r) and torch.equal(v_f, v_r)
            if with_delta:
                assert torch.equal(ho_f, ho_r)
            if tag == (step & 0xff):
                first = out_f.clone()
                # fp32 reference of the attention on the rotated q / the slab after the append (what both kernels approximate)
                T = T_old + step + 1
                rep = nH // nKV
                half = d // 2
                q = qkv_r[0, : nH * d].view(nH, d).float()
                c_, s_ = cos[int(pos[0])].float(), sin[int(pos[0])].float()
                rot = lambda x: torch.cat([-x[..., half:], x[..., :half]], dim=-1)
                q_rot = ((q * c_).to(dtype).float() + (rot(q) * s_).to(dtype).float()).to(dtype).float()
                kk = k_r[0, :, :T].float().repeat_interleave(rep, dim=0)
                vv = v_r[0, :, :T].float().repeat_interleave(rep, dim=0)
                p_ = torch.softmax(torch.einsum("hd,htd->ht", q_rot, kk) / math.sqrt(d), dim=-1)
                ref32 = torch.einsum("ht,htd->hd", p_, vv).reshape(1, nH * d)
                e_f, e_r = float((out_f.float() - ref32).abs().max()), float((out_r.float() - ref32).abs().max())
                assert e_f <= max(2 * e_r, 2 * ulp * float(ref32.abs().max())), (step, e_f, e_r)
            else:
                assert torch.equal(out_f, first), "the fused launch must be repeatable bit for bit (other call tag, same inputs)"
        # several attention workgroups per head (128 slab keys each, partials merged by the head's first workgroup): everything bit-identical
        # except the attention output, which stays in the rounding class; rows shorter than a workgroup's share leave some workgroups empty
        for ns_ in (2, 3, 4):
            k_f, v_f = k0.clone(), v0.clone()
            qkv_f, ho_f, out_f = torch.zeros_like(qkv_r), torch.zeros_like(ho_r), torch.zeros_like(out_r)
            ops.gemv_qkv_attn(W, qkv_f, h0, ho_f, delta, nw, eps, cos, sin, pos, lens, k_f, v_f, out_f, gran, 50 + 8 * ns_ + step, nH, nKV, d, err=err, n_splits=ns_)
            assert torch.equal(qkv_f, qkv_r) and torch.equal(k_f, k_r) and torch.equal(v_f, v_r), (step, ns_)
            diff = float((out_f.float() - out_r.float()).abs().max())
            assert diff <= 2 * ulp * float(out_r.float().abs().max()), (step, ns_, diff)
            e_f = float((out_f.float() - ref32).abs().max())
            assert e_f <= max(2 * e_r, 2 * ulp * float(ref32.abs().max())), (step, ns_, e_f, e_r)
        k0, v0 = k_r, v_r  # the appended row stays for the next step
    assert int(err.item()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,I,D", [(4096, 11008, 1024), (5120, 13824, 1024), (1024, 2816, 256), (512, 1536, 64)])
def test_gemv_gu_tp_bit_identical_to_the_separate_launches(ops, dtype, H, I, D):
    """dl_gemv_gu_tp (gate|up projection with add+RMSNorm prologue and SiLU*up epilogue AND the text predictor's three stages as extra workgroups of
    the same launch, handing h1 / a1 on as granules) against dl_gemv + dl_text_predictor_decide: activation row, residual stream, predictor
    logits and decision bit for bit, repeated over steps / call tags on the same granule buffer."""
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *shape, s=0.02: (torch.randn(*shape, device="cuda", generator=g) * s).to(dtype)
    Wgu, nw = rnd(2 * I, H), 1 + rnd(H, s=0.1)
    w = ops.TpWeights()
    keep = [1 + rnd(H, s=0.1), rnd(H, s=0.1), rnd(D, H, s=0.05), rnd(D, s=0.1), rnd(D // 2, D, s=0.1), rnd(D // 2, s=0.1), rnd(D // 4, D // 2, s=0.2), rnd(D // 4, s=0.1),
            rnd(2, D // 4, s=0.5), rnd(2, s=0.1)]
    (w.ln_w, w.ln_b, w.l1_w, w.l1_b, w.l3_w, w.l3_b, w.l5_w, w.l5_b, w.l7_w, w.l7_b) = [t.data_ptr() for t in keep]
    eps = 1e-5
    gran = ops.gemv_gu_tp_workspace(D, "cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws_r, ws_f = ops.text_predictor_workspace(1, D, "cuda"), ops.text_predictor_workspace(1, D, "cuda")
    decs = []
    for step in range(8):
        h0, delta = rnd(1, H, s=1.0), rnd(1, H, s=1.0)
        pos = torch.tensor([50 + step // 2], dtype=torch.int32, device="cuda")
        y_r, ho_r = torch.zeros(1, I, dtype=dtype, device="cuda"), torch.zeros(1, H, dtype=dtype, device="cuda")
        lg_r, dec_r = torch.zeros(1, 2, dtype=torch.float32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.gemv(Wgu, y_r, mode=ops.GEMV_ADDNORM | ops.GEMV_OUT_SILU_PAIR, h_in=h0, h_out=ho_r, delta=delta, norm_w=nw, eps=eps)
        ops.text_predictor_decide(h0, w, D, ws_r, lg_r, dec_r)
        y_f, ho_f = torch.zeros_like(y_r), torch.zeros_like(ho_r)
        lg_f, dec_f = torch.full_like(lg_r, 7.0), torch.full_like(dec_r, 5)
        ops.gemv_gu_tp(Wgu, y_f, h0, ho_f, delta, nw, eps, w, D, ws_f, lg_f, dec_f, pos, gran, step & 1, err=err)
        assert torch.equal(y_f, y_r) and torch.equal(ho_f, ho_r), step
        assert torch.equal(lg_f, lg_r) and torch.equal(dec_f, dec_r) and torch.equal(ws_f, ws_r), (step, lg_f, lg_r)
        decs.append(int(dec_r))
    assert int(err.item()) == 0


# ---- round 5: dl_linear_packed (operand-order weights, LDS-DMA loaders + MFMA consumers, k ranges shared between workgroups) ----
def _lp_ref(x, w):
    return F.linear(x.float(), w.float())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize(
    "M,N,K,nu,ks",
    [(170, 12288, 4096, 6, 2), (170, 12288, 4096, 3, 1), (117, 4096, 4096, 2, 2), (192, 4096, 11008, 4, 4), (25, 2048, 1024, 0, 1), (32, 6144, 512, 2, 2), (1, 256, 128, 1, 2),
     (256, 512, 2048, 8, 8), (200, 1040, 320, 3, 1), (16, 48, 64, 1, 1), (170, 15360, 5120, 8, 2), (64, 320, 1088, 4, 1)],
)
def test_linear_packed_vs_fp32_every_layout(ops, dtype, M, N, K, nu, ks):
    """Y = X W^T on the operand-order weight copy against an fp32 product of the same rounded inputs: row-major and fragment-order X, strided rows,
    every units / k-range split, ragged last workgroup (N / 16 not a multiple of the units per workgroup), ragged row tiles; bit-identical between
    the two X layouts and from call to call (fixed summation order); dl_pack_x_tiles == the layout the header states."""
    g = torch.Generator().manual_seed(5)
    x_full = torch.randn(M, K + 40, generator=g).to(dtype).cuda()
    x = x_full[:, :K]
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).cuda()
    wp = ops.pack_weight_tiles(w)
    # the weight layout of include/dynllava.h, restated
    S = K // 32
    wp_ref = w.view(N // 16, 16, S, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    assert torch.equal(wp, wp_ref)
    xp = ops.pack_x_tiles(x)
    tiles = 4 * -(-(-(-M // 16)) // 4)
    xpad = torch.cat([x, x[-1:].expand(tiles * 16 - M, K)]) if tiles * 16 > M else x
    assert torch.equal(xp, xpad.reshape(tiles, 16, K // 64, 2, 4, 8).permute(2, 0, 3, 4, 1, 5).contiguous().view(-1))
    ws = ops.linear_packed_workspace(M, N, K, "cuda", ops.LP_STORE, nu, ks)
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    kw = dict(units_per_workgroup=nu, k_split=ks, workspace=ws, err=err)
    ref = _lp_ref(x, w)
    y_row = ops.linear_packed(x, wp, N, **kw)
    y_pk = ops.linear_packed(xp, wp, N, x_packed_mk=(M, K), **kw)
    assert y_row.shape == (M, N) and torch.equal(y_row, y_pk)
    assert torch.equal(y_pk, ops.linear_packed(xp, wp, N, x_packed_mk=(M, K), **kw)), "a k-split launch must leave its workspace ready for the next one"
    ulp = ULP[dtype]
    assert float((y_pk.float() - ref).abs().max()) <= (0.5 * ulp + 3e-5 * math.sqrt(K / 1024 + 1)) * max(1.0, float(ref.abs().max())) * 1.01
    assert int(err.item()) == 0
    if ws is not None:  # (the flag words lead the workspace, padded to 256 bytes)
        assert int(ws[:256].view(torch.int32).abs().sum()) == 0, "flag words must be zero after a launch"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,I,ks,scale", [(170, 4096, 11008, 1, 1.0), (117, 1024, 2816, 2, 1.0), (32, 512, 1536, 1, 1.0), (192, 5120, 13824, 1, 1.0), (170, 512, 2816, 1, 6.0), (64, 256, 1536, 1, 40.0),
                                            (48, 256, 1536, 1, 0.01)])
def test_linear_packed_epilogues_bit_equal_the_separate_launches(ops, dtype, M, H, I, ks, scale):
    """SiLU(gate) * up on the gate / up interleaved packing == dl_silu_mul applied to the ROUNDED plain output of the same kernel (DML:328: the
    projection is rounded, silu is rounded, the product is rounded); residual epilogue == rounded output added to the residual by torch
    (DML:1289 / 1295).  Round 6: the epilogue evaluates silu with v_exp_f32 / v_rcp_f32 and takes the exact expression only near a rounding boundary of the
    16-bit type (csrc/act_round.h) -- `scale` spreads the gate values over the saturating (x 6, x 40: |gate| up to ~150) and the tiny (x 0.01: fp16-subnormal
    results) ranges, every element still bit-equal to dl_silu_mul's exact expression."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(M, H, generator=g).to(dtype).cuda()
    w_gu = (scale * torch.randn(2 * I, H, generator=g) / math.sqrt(H)).to(dtype).cuda()
    cands = [ops.linear_packed_workspace(M, 2 * I, H, "cuda", e, 0, ks) for e in (ops.LP_SILU_PAIR, ops.LP_STORE)]  # (the chosen units per workgroup differ)
    ws = max((c for c in cands if c is not None), key=lambda c: c.numel(), default=None)
    plain = ops.linear_packed(x, ops.pack_weight_tiles(w_gu), 2 * I, k_split=ks, workspace=ws)  # same k order -> same fp32 sums whatever the unit order
    fused = ops.linear_packed(x, ops.pack_weight_tiles(w_gu, gate_up_pairs=True), 2 * I, epilogue=ops.LP_SILU_PAIR, k_split=ks, workspace=ws)
    assert fused.shape == (M, I)
    assert torch.equal(fused, ops.silu_mul(plain))
    w_o = (torch.randn(H, H, generator=g) / math.sqrt(H)).to(dtype).cuda()
    wp_o = ops.pack_weight_tiles(w_o)
    res = torch.randn(M, H, generator=g).to(dtype).cuda()
    ws2 = ops.linear_packed_workspace(M, H, H, "cuda", ops.LP_RESID, 0, 2)
    y = ops.linear_packed(x, wp_o, H, k_split=2, workspace=ws2)
    h = res.clone()
    ops.linear_packed(x, wp_o, H, out=h, epilogue=ops.LP_RESID, resid=h, k_split=2, workspace=ws2)  # in place on the residual stream
    assert torch.equal(h, (res.float() + y.float()).to(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,H", [(170, 4096), (1, 64), (117, 5120), (256, 1024), (33, 128)])
def test_packed_norm_outputs_are_the_row_major_ones_in_fragment_order(ops, dtype, rows, H):
    """dl_rmsnorm_packed / dl_add_rmsnorm_packed / dl_add_rmsnorm_parts_packed: same values as the row-major launches, stored where dl_pack_x_tiles
    would put them (rows past the last one are not written: compared on the rows that exist)."""
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(rows, H, generator=g) * 2).to(dtype).cuda()
    d = torch.randn(rows, H, generator=g).to(dtype).cuda()
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype).cuda()
    parts = torch.randn(3, rows, H, generator=g).cuda()
    tiles = 4 * -(-(-(-rows // 16)) // 4)

    def unpack(p):
        return p[: tiles * 16 * H].view(H // 64, tiles, 2, 4, 16, 8).permute(1, 4, 0, 2, 3, 5).reshape(tiles * 16, H)[:rows]

    assert torch.equal(unpack(ops.rmsnorm(x, w, 1e-5, packed=True)), ops.rmsnorm(x, w, 1e-5))
    h1, h2 = x.clone(), x.clone()
    assert torch.equal(unpack(ops.add_rmsnorm(h1, d, w, 1e-5, packed=True)), ops.add_rmsnorm(h2, d, w, 1e-5)) and torch.equal(h1, h2)
    h1, h2 = x.clone(), x.clone()
    assert torch.equal(unpack(ops.add_rmsnorm_parts(h1, parts, w, 1e-5, packed=True)), ops.add_rmsnorm_parts(h2, parts, w, 1e-5)) and torch.equal(h1, h2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,I,ks,nu", [(170, 4096, 11008, 4, 4), (117, 1024, 2816, 2, 0), (32, 512, 1536, 1, 0), (192, 5120, 13824, 4, 4)])
def test_linear_packed_fragment_order_output_feeds_the_next_call_and_partial_sums(ops, dtype, M, H, I, ks, nu):
    """The MLP chain of a prefill layer on dl_linear_packed: gate|up + SiLU * up written in FRAGMENT order (DL_LP_Y_PACKED) is exactly what
    dl_pack_x_tiles makes of the row-major result, and down_proj with DL_LP_PARTS on that input gives fp32 partial sums [k ranges, M, H] whose sum is
    the plain call's fp32 accumulation; fed to dl_add_rmsnorm_parts they reproduce library GEMM + dl_add_rmsnorm up to the rounding of the sum."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, H, generator=g).to(dtype).cuda()
    w_gu = (torch.randn(2 * I, H, generator=g) / math.sqrt(H)).to(dtype).cuda()
    w_dn = (torch.randn(H, I, generator=g) / math.sqrt(I)).to(dtype).cuda()
    wp_gu, wp_dn = ops.pack_weight_tiles(w_gu, gate_up_pairs=True), ops.pack_weight_tiles(w_dn)
    act_row = ops.linear_packed(x, wp_gu, 2 * I, epilogue=ops.LP_SILU_PAIR)
    act_pk = ops.linear_packed(x, wp_gu, 2 * I, epilogue=ops.LP_SILU_PAIR, y_packed=True)
    rows_pk = torch.equal(act_pk[: ops.pack_x_tiles(act_row).numel()], ops.pack_x_tiles(act_row))
    if not rows_pk:  # rows past M are not written by the epilogue (dl_pack_x_tiles repeats row M - 1 there): compare the rows that exist
        tiles = 4 * -(-(-(-M // 16)) // 4)
        un = lambda p_: p_[: tiles * 16 * I].view(I // 64, tiles, 2, 4, 16, 8).permute(1, 4, 0, 2, 3, 5).reshape(tiles * 16, I)[:M]
        assert torch.equal(un(act_pk), act_row)
    parts = ops.linear_packed(act_pk, wp_dn, H, epilogue=ops.LP_PARTS, units_per_workgroup=nu, k_split=ks, x_packed_mk=(M, I))
    assert parts.shape == (ks, M, H) and parts.dtype == torch.float32
    ref = F.linear(act_row.float(), w_dn.float())
    assert float((parts.sum(0) - ref).abs().max()) <= 3e-5 * math.sqrt(I / 1024 + 1) * max(1.0, float(ref.abs().max()))
    assert torch.equal(parts, ops.linear_packed(act_pk, wp_dn, H, epilogue=ops.LP_PARTS, units_per_workgroup=nu, k_split=ks, x_packed_mk=(M, I)))
    h0 = torch.randn(M, H, generator=g).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype).cuda()
    h1, h2 = h0.clone(), h0.clone()
    x1 = ops.add_rmsnorm_parts(h1, parts.contiguous(), nw, 1e-5)
    x2 = ops.add_rmsnorm(h2, ref.to(dtype), nw, 1e-5)
    assert float((h1.float() - h2.float()).abs().max()) <= 2 * ULP[dtype] * float(h2.float().abs().max())
    _close_ulp(x1, x2, dtype, 2.0, atol=2e-2)


def test_linear_packed_rejects_bad_arguments(ops):
    x = torch.zeros(8, 128, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(64, 128, dtype=torch.bfloat16, device="cuda")
    wp = ops.pack_weight_tiles(w)
    with pytest.raises(ops.HipOpsError):
        ops.pack_weight_tiles(torch.zeros(60, 128, dtype=torch.bfloat16, device="cuda"))  # N % 16
    with pytest.raises(ops.HipOpsError):
        ops.pack_weight_tiles(torch.zeros(64, 96, dtype=torch.bfloat16, device="cuda"))  # K % 64
    with pytest.raises(ops.HipOpsError):
        ops.pack_weight_tiles(torch.zeros(64, 128, device="cuda"))  # fp32
    with pytest.raises(ops.HipOpsError):
        ops.linear_packed(x, wp, 64, units_per_workgroup=5)
    with pytest.raises(ops.HipOpsError):
        ops.linear_packed(x, wp, 64, k_split=3, workspace=torch.zeros(1 << 20, dtype=torch.uint8, device="cuda"))  # K / 64 = 2 steps
    with pytest.raises((ops.HipOpsError, AssertionError)):
        ops.linear_packed(x, wp, 64, k_split=2)  # no workspace
    big = torch.zeros(300, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ops.HipOpsError):
        ops.linear_packed(big, wp, 64)  # more than one tile of rows
    assert ops.linear_packed(x[:0], wp, 64).shape == (0, 64)  # empty input: a no-op


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_parts,total,nH,nKV,d", [(2, 170, 32, 32, 128), (4, 37, 8, 8, 64), (1, 5, 4, 2, 128)])
def test_rope_kv_write_parts_equals_rope_on_the_rounded_sum(ops, dtype, n_parts, total, nH, nKV, d):
    """dl_rope_kv_write_parts (q|k|v as dl_linear_packed's fp32 partial sums) == dl_rope_kv_write on the sum of the parts (part order) rounded once to the
    model dtype: q / k rotated, v as is, slab rows -- bit for bit; rows past the last sequence untouched."""
    g = torch.Generator().manual_seed(9)
    W = (nH + 2 * nKV) * d
    pad = 3  # a launch sized for a width bucket: rows past cu[B] are padding
    parts = torch.randn(n_parts, total + pad, W, generator=g).cuda()
    acc = parts[0].clone()
    for s_ in range(1, n_parts):
        acc += parts[s_]
    qkv_ref = acc.to(dtype)
    B = 2
    cu = torch.tensor([0, total // 2, total], dtype=torch.int32, device="cuda")
    T_cap = total + 8
    cos = torch.randn(T_cap, d, generator=g).to(dtype).cuda()
    sin = torch.randn(T_cap, d, generator=g).to(dtype).cuda()
    pos = torch.randint(0, T_cap, (total + pad,), generator=g).to(torch.int32).cuda()
    kvb = torch.tensor([3, 0], dtype=torch.int32, device="cuda")
    outs = []
    for use_parts in (False, True):
        k_slab = torch.zeros(B, nKV, T_cap, d, dtype=dtype, device="cuda")
        v_slab = torch.zeros_like(k_slab)
        if use_parts:
            qkv = torch.full((total + pad, W), 7.0, dtype=dtype, device="cuda")
            ops.rope_kv_write(qkv, cos, sin, cu, pos, None, kvb, k_slab, v_slab, nH, nKV, d, parts=parts)
            assert bool((qkv[total:] == 7.0).all()), "padding rows must stay untouched"
        else:
            qkv = qkv_ref.clone()
            ops.rope_kv_write(qkv, cos, sin, cu, pos, None, kvb, k_slab, v_slab, nH, nKV, d)
        outs.append((qkv[:total].clone(), k_slab, v_slab))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
