"""dl_linear_tiles (csrc/linear_tiles.hip): the tiled MFMA GEMM of the vision side -- CLIP encoder projections (clip_encoder.py:53-71 ->
transformers CLIPEncoderLayer), the mlp2x_gelu projector (multimodal_projector/builder.py:172-179), the vision predictor's linears (DML:1348-1359).
Compared with an fp32 evaluation of the same Linear on the same 16-bit inputs (this tier's torch fp32 reference for a floating-point kernel):
tolerance = one rounding of the output dtype on top of the fp32 accumulation error, written out below."""
import pytest
import torch

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

SHAPES = [  # (M, N, K): CLIP q|k|v / out_proj / fc1 / fc2 at B = 1, projector, predictor, ragged edges, B = 2
    (577, 3072, 1024), (577, 1024, 1024), (577, 4096, 1024), (577, 1024, 4096), (576, 4096, 4096), (576, 512, 4096), (576, 2048, 512),
    (1, 64, 64), (17, 48, 128), (81, 272, 192), (1154, 1024, 1024),
]


def _ops():
    from dynamic_llava_amd import hip_ops as ops

    return ops


def _ref(x, w, b, act):
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    if act == "qgelu":
        a = y.to(x.dtype).float()
        t = (1.702 * a).to(x.dtype).float()
        y = a * torch.sigmoid(t).to(x.dtype).float()
    elif act == "gelu":
        y = torch.nn.functional.gelu(y.to(x.dtype).float())
    return y


def _tol(ref, dtype, act=False):
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # plain Linear: the output rounding (0.5 ulp) + fp32 summation order; activations: three more roundings of the eager op chain (input, 1.702 x, sigmoid)
    return (3.0 if act else 1.5) * ulp * ref.abs().clamp_min(1.0) + 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_tiles_bias_vs_fp32(M, N, K, dtype):
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
    b = torch.randn(N, device="cuda", generator=g).to(dtype)
    wp = ops.pack_weight_tiles(w)
    ref = _ref(x, w, b, None)
    y = ops.linear_tiles(x, wp, N, bias=b)
    assert (y.float() - ref).abs().le(_tol(ref, dtype)).all()
    # fragment-order input == row-major input, bit for bit (same accumulation order)
    xp = ops.pack_x_rows(x)
    y2 = ops.linear_tiles(xp, wp, N, bias=b, x_packed_mk=(M, K))
    assert torch.equal(y, y2)
    # no bias
    y3 = ops.linear_tiles(x, wp, N)
    ref3 = _ref(x, w, None, None)
    assert (y3.float() - ref3).abs().le(_tol(ref3, dtype)).all()


@pytest.mark.parametrize("shape", [542, 532, 522, 512, 521, 541, 20542, 20532, 20521, 20541, 1042, 1032, 1041])
def test_linear_tiles_every_built_tile_shape_same_bits(shape):
    """The result is a function of the k order only: every tile shape returns the same bits (one fp32 chain per output, k ascending)."""
    ops = _ops()
    M, N, K = 577, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / 32).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    wp = ops.pack_weight_tiles(w)
    y0 = ops.linear_tiles(x, wp, N, bias=b, tile_shape=542)
    y = ops.linear_tiles(x, wp, N, bias=b, tile_shape=shape)
    assert torch.equal(y, y0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_tiles_quick_gelu_epilogue_equals_separate_launch(dtype):
    """fc1 + QuickGELU in the epilogue == dl_linear_tiles followed by dl_quick_gelu, bit for bit; fragment-order output == pack_x_rows of it."""
    ops = _ops()
    M, N, K = 577, 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / 32).to(dtype)
    b = torch.randn(N, device="cuda", generator=g).to(dtype)
    wp = ops.pack_weight_tiles(w)
    y = ops.linear_tiles(x, wp, N, bias=b)
    want = ops.quick_gelu(y)
    got = ops.linear_tiles(x, wp, N, bias=b, epilogue=ops.LT_QGELU)
    assert torch.equal(got, want)
    ref = _ref(x, w, b, "qgelu")
    assert (got.float() - ref).abs().le(_tol(ref, dtype, act=True)).all()
    gp = ops.linear_tiles(x, wp, N, bias=b, epilogue=ops.LT_QGELU, y_packed=True)
    n_real = ops.tiles_x_numel(M, N)
    wantp = ops.pack_x_rows(want)
    # rows past M in the last tile are not written by the epilogue (pack_x_rows repeats row M - 1 there): compare through a second GEMM's eyes instead
    w2 = (torch.randn(1024, N, device="cuda", generator=g) / 64).to(dtype)
    wp2 = ops.pack_weight_tiles(w2)
    z1 = ops.linear_tiles(gp, wp2, 1024, x_packed_mk=(M, N))
    z2 = ops.linear_tiles(wantp, wp2, 1024, x_packed_mk=(M, N))
    assert torch.equal(z1, z2) and gp.numel() == n_real


def test_linear_tiles_gelu_epilogue_vs_fp32():
    ops = _ops()
    M, N, K = 576, 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / 32).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    wp = ops.pack_weight_tiles(w)
    got = ops.linear_tiles(x, wp, N, bias=b, epilogue=ops.LT_GELU)
    ref = _ref(x, w, b, "gelu")
    assert (got.float() - ref).abs().le(_tol(ref, torch.bfloat16, act=True)).all()


def test_linear_tiles_hardware_bf16_rounding_equals_torch_rne():
    """The epilogue rounds with v_cvt_pk_bf16_f32: the 16-bit output must be torch's round-to-nearest-even of the launch's own fp32 sums (partial-sum form,
    one k range: the same accumulators, unrounded) -- 2.4 M values, bit for bit; huge / tiny magnitudes through a scaled copy."""
    ops = _ops()
    M, N, K = 577, 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    for scale in (1 / 32, 2.0 ** 100, 2.0 ** -120):
        w = (torch.randn(N, K, device="cuda", generator=g) * scale).bfloat16()
        wp = ops.pack_weight_tiles(w)
        acc = ops.linear_tiles(x, wp, N, epilogue=ops.LT_PARTS, k_split=1)[0]
        y = ops.linear_tiles(x, wp, N)
        assert torch.equal(y, acc.to(torch.bfloat16))


@pytest.mark.parametrize("k_split", [1, 2, 4, 8])
def test_linear_tiles_partial_sums(k_split):
    """fc2 as k ranges: slice r holds the fp32 sum over range r; the slices added in order reproduce the one-range result to fp32 rounding."""
    ops = _ops()
    M, N, K = 577, 1024, 4096
    g = torch.Generator(device="cuda").manual_seed(k_split)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / 64).bfloat16()
    wp = ops.pack_weight_tiles(w)
    parts = ops.linear_tiles(x, wp, N, epilogue=ops.LT_PARTS, k_split=k_split)
    assert parts.shape == (k_split, M, N) and parts.dtype == torch.float32
    steps = K // 64
    for r in range(k_split):
        k0, k1 = 64 * (steps * r // k_split), 64 * (steps * (r + 1) // k_split)
        ref = x[:, k0:k1].float() @ w[:, k0:k1].float().t()
        assert (parts[r] - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    total = parts[0].clone()
    for r in range(1, k_split):
        total += parts[r]
    ref = x.float() @ w.float().t()
    assert (total - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()


def test_linear_tiles_row_position_invariance_and_batch():
    """A row's result does not depend on where it sits: the B = 2 packed batch returns the B = 1 rows twice (DESIGN section 5's invariant for this package's kernels)."""
    ops = _ops()
    M, N, K = 577, 3072, 1024
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / 32).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    wp = ops.pack_weight_tiles(w)
    y1 = ops.linear_tiles(x, wp, N, bias=b)
    y2 = ops.linear_tiles(torch.cat([x, x], 0).contiguous(), wp, N, bias=b)
    assert torch.equal(y2[:M], y1) and torch.equal(y2[M:], y1)


def test_linear_tiles_rejects_bad_arguments():
    ops = _ops()
    x = torch.zeros(32, 128, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(64, 128, device="cuda", dtype=torch.bfloat16)
    wp = ops.pack_weight_tiles(w)
    with pytest.raises(ops.HipOpsError):
        ops.linear_tiles(x, wp, 64, tile_shape=999)  # not built
    with pytest.raises(ops.HipOpsError):
        ops.linear_tiles(x, wp, 64, k_split=2)  # k ranges only as partial sums
    with pytest.raises(ops.HipOpsError):
        ops.linear_tiles(x.float(), wp.float(), 64)  # 16-bit types only
    with pytest.raises(ops.HipOpsError):
        ops.linear_tiles(x, wp, 64, epilogue=ops.LT_PARTS, k_split=4)  # K / 64 = 2 steps < 4 ranges


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,H", [(577, 1024), (1154, 1024), (5, 512), (576, 4096), (33, 2048)])
def test_layernorm_rows_family_vs_fp32(rows, H, dtype):
    """dl_layernorm_rows / dl_add_layernorm_rows / dl_add_layernorm_parts (a wave per row) against nn.LayerNorm evaluated in fp32 on the same inputs; the
    fragment-order output equals pack_x_rows of the row-major output bit for bit; the in-place residual stream is bit-exact (cast(h + cast(delta)))."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(rows + H)
    x = torch.randn(rows, H, device="cuda", generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(dtype)
    b = (0.1 * torch.randn(H, device="cuda", generator=g)).to(dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11

    def ln32(t):
        return torch.nn.functional.layer_norm(t.float(), (H,), w.float(), b.float(), 1e-5)

    y = ops.layernorm_rows(x, w, b, 1e-5)
    ref = ln32(x)
    assert (y.float() - ref).abs().le(ulp * ref.abs().clamp_min(1.0) + 1e-3).all()
    if H % 64 == 0:
        yp = ops.layernorm_rows(x, w, b, 1e-5, packed=True)
        n_full = (rows // 16) * 16  # rows of the last, partial tile past `rows` are not written: compare whole tiles, then the tail through unpacking
        idx = torch.arange(rows, device="cuda")
        want = ops.pack_x_rows(y)
        tiles = (rows + 15) // 16
        a = yp.view(H // 64, tiles, 2, 64, 8)
        c = want.view(H // 64, tiles, 2, 64, 8)
        lanes_ok = (torch.arange(64, device="cuda") % 16)[None, :] + 16 * torch.arange(tiles, device="cuda")[:, None] < rows  # [tiles, 64]
        assert torch.equal(a[:, lanes_ok.nonzero()[:, 0], :, lanes_ok.nonzero()[:, 1]], c[:, lanes_ok.nonzero()[:, 0], :, lanes_ok.nonzero()[:, 1]])
    # residual add (16-bit delta)
    d = torch.randn(rows, H, device="cuda", generator=g).to(dtype)
    h = x.clone()
    y2 = ops.add_layernorm_rows(h, d, w, b, 1e-5)
    h_ref = (x.float() + d.float()).to(dtype)
    assert torch.equal(h, h_ref)
    ref2 = ln32(h_ref)
    assert (y2.float() - ref2).abs().le(ulp * ref2.abs().clamp_min(1.0) + 1e-3).all()
    # residual add of fp32 k-range partial sums + bias
    parts = torch.randn(3, rows, H, device="cuda", generator=g)
    bias = torch.randn(H, device="cuda", generator=g).to(dtype)
    h = x.clone()
    y3 = ops.add_layernorm_parts(h, parts, bias, w, b, 1e-5)
    delta = (((parts[0] + parts[1]) + parts[2]) + bias.float()).to(dtype)
    h_ref = (x.float() + delta.float()).to(dtype)
    assert torch.equal(h, h_ref)
    ref3 = ln32(h_ref)
    assert (y3.float() - ref3).abs().le(ulp * ref3.abs().clamp_min(1.0) + 1e-3).all()
    # add only
    h = x.clone()
    assert ops.add_layernorm_parts(h, parts, None) is None
    assert torch.equal(h, (x.float() + ((parts[0] + parts[1]) + parts[2]).to(dtype).float()).to(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H,shape,ks", [(170, 4096, 642, 4), (117, 4096, 842, 8), (192, 4096, 642, 4), (32, 4096, 20242, 8), (16, 4096, 20142, 8), (24, 5120, 20242, 6), (179, 5120, 642, 3),
                                          (64, 4096, 442, 8), (48, 4096, 342, 8), (112, 4096, 742, 8), (128, 4096, 20842, 8)])
def test_linear_tiles_o_proj_partial_sum_shapes(M, H, shape, ks, dtype):
    """The decoder's o_proj (DML:1127) at <= 256 rows as TM row tiles x 8 units x k ranges of fp32 partial sums: the slices added in order equal the fp32 product of the
    16-bit inputs to fp32 rounding, for the tile shapes model._tiles_o_config picks (prefill M = 117..192 at 7B / 13B width, decode batches of 16..32 rows)."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(M + H + ks)
    x = torch.randn(M, H, device="cuda", generator=g).to(dtype)
    w = (torch.randn(H, H, device="cuda", generator=g) / 64).to(dtype)
    wp = ops.pack_weight_tiles(w)
    parts = ops.linear_tiles(x, wp, H, epilogue=ops.LT_PARTS, tile_shape=shape, k_split=ks)
    assert parts.shape == (ks, M, H)
    total = parts[0].clone()
    for r in range(1, ks):
        total += parts[r]
    ref = x.float() @ w.float().t()
    assert (total - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    with pytest.raises(ops.HipOpsError):
        ops.linear_tiles(x, wp, H, tile_shape=shape)  # these shapes are built for the partial-sum form only


def test_tiles_o_config_covers_every_row_count():
    """Every row count up to 256 at 7B / 13B width maps to a built tile shape and to one round of workgroups."""
    from dynamic_llava_amd.model import DynamicLlavaLlamaForCausalLM as M_

    built = {142, 242, 342, 442, 642, 742, 842}
    for H in (4096, 5120):
        for rows in range(1, 257):
            shp, ks = M_._tiles_o_config(rows, H)
            tm = (shp % 10000) // 100
            assert shp % 10000 in built and 1 <= ks <= 8
            n_mb = -(-((rows + 15) // 16) // tm)
            assert n_mb * -(-(H // 16) // 8) * ks <= 256 and n_mb <= 2
