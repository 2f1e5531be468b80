"""GPU: decode with dl_decode_block (csrc/decode_block.hip: o_proj -> gate|up -> down -> next q|k|v as ONE launch per layer on the LDS-DMA
engine, `model.use_block_decode`) against the launch path (dl_gemv x4 per layer).  Both run the same arithmetic in the same order
(shared gemv_dot.h), so the comparison is BIT-EXACT on logits, eviction decisions, KV contents and lengths -- any stale or torn
inter-workgroup hand-off shows up as a mismatch.  The launch path itself is pinned to the oracle / reference goldens elsewhere."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx  # noqa: E402


def _build(cfg_ns, sd, dtype):
    from dynamic_llava_amd.builder import build_from_state_dict
    from dynamic_llava_amd.config import DynamicLlavaConfig

    return build_from_state_dict(DynamicLlavaConfig.from_namespace(cfg_ns), sd, None, dtype=dtype, device="cuda")


def _drive(model, ids, feats, forced, block):
    """prefill + teacher-forced decode through forward(); returns per-step logits, decisions, final lengths and a KV checksum."""
    model.use_block_decode = block
    model.debug_records = {}
    out = model(ids.cuda(), image_features=feats.cuda())
    pkv = out.past_key_values
    logits, dec = [], []
    for j in range(forced.shape[0]):
        out = model(forced[j][:, None].cuda(), past_key_values=pkv)
        pkv = out.past_key_values
        logits.append(out.logits[0, -1].clone())
        d = model.debug_records.get("text_decision")
        dec.append(None if d is None else int(d[0]))
    st = model._dstate
    if block:
        assert st is not None and model._block_ok(st) and st.blk_sync is not None, "the block path must be the one that ran"
        model.check_block_decode()
    lens = [t.clone() for t in pkv[1]]
    n0, n1 = int(lens[0][0]), int(lens[-1][0])
    L, SL = model.config.num_hidden_layers, model.config.sparse_config["sparse_layer"]
    ksum = [pkv.k[i][0, :, : (n0 if i < SL else n1)].float().sum().item() for i in range(L)]
    vsum = [pkv.v[i][0, :, : (n0 if i < SL else n1)].float().sum().item() for i in range(L)]
    model.debug_records = None
    return logits, dec, (n0, n1), ksum, vsum


CASES = {
    # name: (config builder, vocab, n_sys, n_q, n_img tokens, steps)
    "7b_width": (lambda: fx.llava7b_config(num_hidden_layers=3), 4096, 35, 20, 576, 6),
    "13b_width": (lambda: fx.llava13b_config(num_hidden_layers=3), 4096, 35, 29, 576, 5),
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", sorted(CASES))
def test_block_decode_bit_identical_to_launch_path(name, dtype):
    mk, vocab, n_sys, n_q, n_img, steps = CASES[name]
    cfg = mk()
    if vocab:
        cfg.vocab_size = vocab
    sd = fx.make_state_dict(cfg, seed=21, predictor_gain=50.0)
    model = _build(cfg, sd, dtype)
    g = torch.Generator().manual_seed(7)
    ids = fx.make_prompt(cfg, n_sys, n_q, seed=4)[None]
    feats = torch.randn(1, n_img, cfg.hidden_size, generator=g).to(dtype)
    forced = fx.make_forced_tokens(cfg, steps, 1, seed=6)
    a = _drive(model, ids, feats, forced, block=False)
    b = _drive(model, ids, feats, forced, block=True)
    for j in range(steps):
        assert torch.equal(a[0][j], b[0][j]), f"step {j}: max |diff| {float((a[0][j] - b[0][j]).abs().max())}"
    assert a[1] == b[1] and a[2] == b[2]
    assert a[3] == b[3] and a[4] == b[4], "appended K/V rows differ"


def test_block_decode_generate_graph_replay_equals_launch_path():
    """generate(): the captured step (per layer: attention launch + dl_decode_block; predictor; advance) replays to the launch path's tokens."""
    dtype = torch.bfloat16
    cfg = fx.llava7b_config(num_hidden_layers=4)
    cfg.vocab_size = 4096
    sd = fx.make_state_dict(cfg, seed=23, predictor_gain=50.0)
    model = _build(cfg, sd, dtype)
    g = torch.Generator().manual_seed(9)
    ids = fx.make_prompt(cfg, 35, 20, seed=5)[None]
    feats = torch.randn(1, 576, cfg.hidden_size, generator=g).to(dtype)
    outs = {}
    for block in (False, True):
        for graph in (False, True):
            model.use_block_decode, model.use_hip_graph = block, graph
            outs[block, graph] = model.generate(ids.cuda(), image_features=feats.cuda(), max_new_tokens=40, eos_token_id=None).cpu()
            lens = model.last_cache[1]
            outs[block, graph, "lens"] = (int(lens[0][0]), int(lens[-1][0]))
            # a second request on the same state (same positions, same call tags): the granule workspace is cleared per request
            again = model.generate(ids.cuda(), image_features=feats.cuda(), max_new_tokens=40, eos_token_id=None).cpu()
            assert torch.equal(again, outs[block, graph]), (block, graph)
    ref = outs[False, False]
    for k in ((False, True), (True, False), (True, True)):
        assert torch.equal(outs[k], ref), k
        assert outs[k + ("lens",)] == outs[False, False, "lens"]
    kept = outs[True, True, "lens"][1] - (35 + 115 + 20)
    assert 0 <= kept <= 39


def test_fused_decode_launches_equal_the_separate_launches_in_generate():
    """generate() with dl_gemv_qkv_attn / dl_gemv_gu_tp (the defaults at batch 1) against the same step built from separate launches: tokens, KV
    lengths and the K/V rows of every layer bit for bit, graph replay and eager; the in-kernel hand-offs must not have given up."""
    dtype = torch.bfloat16
    cfg = fx.llava7b_config(num_hidden_layers=4)
    cfg.vocab_size = 4096
    sd = fx.make_state_dict(cfg, seed=29, predictor_gain=50.0)
    model = _build(cfg, sd, dtype)
    g = torch.Generator().manual_seed(3)
    ids = fx.make_prompt(cfg, 35, 20, seed=8)[None]
    feats = torch.randn(1, 576, cfg.hidden_size, generator=g).to(dtype)
    from dynamic_llava_amd.cache import KVSlabCache

    res = {}
    # the fused launch runs the attention body with four waves per workgroup; the stand-alone launch would pick eight for these rows (another
    # key -> lane-group dealing, last-bit differences): compare like with like
    KVSlabCache.eight_wave_single_split = False
    try:
        _run_fused_matrix(model, cfg, ids, feats, res)
    finally:
        KVSlabCache.eight_wave_single_split = True
    ref = res[False, False]
    for key in ((False, True), (True, False), (True, True)):
        out, lens, kv = res[key]
        assert torch.equal(out, ref[0]) and lens == ref[1], key
        for i in range(len(kv)):
            assert torch.equal(kv[i][0], ref[2][i][0]) and torch.equal(kv[i][1], ref[2][i][1]), (key, i)
    assert 0 <= ref[1][1] - (35 + 115 + 20) <= 47


def _run_fused_matrix(model, cfg, ids, feats, res):
    for fused in (False, True):
        for graph in (False, True):
            model.fuse_qkv_attn = model.fuse_gu_tp = fused
            model.use_hip_graph = graph
            out = model.generate(ids.cuda(), image_features=feats.cuda(), max_new_tokens=48, eos_token_id=None).cpu()
            model.check_device_errors()
            lens = model.last_cache[1]
            n0, n1 = int(lens[0][0]), int(lens[-1][0])
            c = model.last_cache
            L, SL = cfg.num_hidden_layers, cfg.sparse_config["sparse_layer"]
            kv = [(c.k[i][0, :, : (n0 if i < SL else n1)].clone(), c.v[i][0, :, : (n0 if i < SL else n1)].clone()) for i in range(L)]
            res[fused, graph] = (out, (n0, n1), kv)
