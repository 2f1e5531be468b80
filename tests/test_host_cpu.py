"""CPU: host logic, C-ABI surface, loud failure without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/dynllava.h is exported by the .so and bound in hip_ops.SIGNATURES."""
    from dynamic_llava_amd import hip_ops

    hdr = open(os.path.join(ROOT, "include", "dynllava.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dl_[a-z0-9_]+)\s*\(", hdr)) - {"dl_vp_block", "dl_vp_weights", "dl_tp_weights"}
    assert declared == set(hip_ops.SIGNATURES), declared ^ set(hip_ops.SIGNATURES)
    lib = hip_ops.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.dl_version() == hip_ops.ABI_VERSION == 4
    assert isinstance(lib.dl_last_error(), bytes)
    # workspace-size queries are pure host functions
    assert lib.dl_attn_decode_workspace_bytes(2, 32, 128, 8) == 2 * 32 * 8 * 132 * 8  # split partials (granules)
    assert lib.dl_attn_decode_workspace_bytes(2, 32, 128, 1) == 0
    assert lib.dl_vision_predictor_workspace_bytes(1, 576, 4096, 512, 2048, 2) > 576 * 4096 * 2


def test_struct_layouts_match_header():
    from dynamic_llava_amd import hip_ops

    assert ctypes.sizeof(hip_ops.VpBlock) == 11 * 8
    assert ctypes.sizeof(hip_ops.VpWeights) == 10 * 8 + 8 + 4 * 11 * 8
    assert ctypes.sizeof(hip_ops.TpWeights) == 10 * 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_hot_path_fails_loudly_without_gpu():
    from dynamic_llava_amd import hip_ops
    from dynamic_llava_amd.builder import build_random_model, load_pretrained_model
    from dynamic_llava_amd.config import DynamicLlavaConfig

    with pytest.raises(hip_ops.HipOpsError):
        hip_ops.require_gpu()
    with pytest.raises(hip_ops.HipOpsError):
        hip_ops.rmsnorm(torch.zeros(2, 128), torch.ones(128), 1e-5)
    with pytest.raises(hip_ops.HipOpsError):
        build_random_model(DynamicLlavaConfig(num_hidden_layers=1), device="cuda")
    with pytest.raises(hip_ops.HipOpsError):
        load_pretrained_model("/nonexistent", None, "x", device="cpu")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under dynamic_llava_amd/ may reference it."""
    pkg = os.path.join(ROOT, "dynamic_llava_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_config_roundtrip_and_state_dict_keys(tmp_path):
    from dynamic_llava_amd.config import DynamicLlavaConfig
    from dynamic_llava_amd.model import DynamicLlavaLlamaForCausalLM
    from oracle import fixtures as fx

    ns = fx.tiny_config()
    cfg = DynamicLlavaConfig.from_namespace(ns)
    cfg.save_pretrained(str(tmp_path))
    cfg2 = DynamicLlavaConfig.from_pretrained(str(tmp_path))
    assert cfg2.to_dict() == cfg.to_dict() and cfg2.sparse_config["sparse_layer"] == 2 and cfg2.n_image_tokens == 36
    import dynamic_llava_amd

    assert dynamic_llava_amd.LlavaLlamaForCausalLM is DynamicLlavaLlamaForCausalLM and dynamic_llava_amd.LlavaConfig is DynamicLlavaConfig
    m = DynamicLlavaLlamaForCausalLM(cfg)
    own = {k for k in m.state_dict() if "vision_tower" not in k}
    ref = set(fx.make_state_dict(ns, seed=0))  # the reference's key names (checked against the reference in oracle/make_golden.py)
    assert own == ref
    # fused QKV / gate|up must stay views of the original parameters (state_dict round-trips, no extra memory)
    l = m.model.layers[0]
    l.pack()
    assert l.self_attn.k_proj.weight.data_ptr() == l.w_qkv[256:].data_ptr()
    assert torch.equal(m.state_dict()["model.layers.0.mlp.up_proj.weight"], l.w_gu[512:])


def test_segment_indices_match_oracle():
    from dynamic_llava_amd.config import DynamicLlavaConfig
    from dynamic_llava_amd.model import DynamicLlavaLlamaForCausalLM
    from oracle import fixtures as fx
    from oracle.ref_cpu import Oracle

    ns = fx.tiny_config()
    ns.vocab_size = 30000  # "USER:" = ids 11889, 29901 (dynamic_llava_arch.py:36)
    m = DynamicLlavaLlamaForCausalLM(DynamicLlavaConfig.from_namespace(ns), with_vision_tower=False)
    o = Oracle(ns, fx.make_state_dict(ns, seed=0), torch.float32)
    ids = [1, 5, 6, -200, 9, 11889, 29901, 7, 8, 11889, 29901, 4, 3]
    feats = torch.zeros(1, 36, 256)
    (_, _, _, _, emb, _), (idx,) = o.prepare_inputs_labels_for_multimodal(torch.tensor([ids]), None, None, None, None, None, image_features=feats)
    seg = m._segments(ids, None, 36)
    assert seg == idx[0] and emb.shape[1] == seg["answer"][1]
    labels = [-100] * 9 + [5] * 4
    (_, _, _, _, _, _), (idx2,) = o.prepare_inputs_labels_for_multimodal(torch.tensor([ids]), None, None, None, torch.tensor([labels]), None, image_features=feats)
    assert m._segments(ids, labels, 36) == idx2[0]


def test_library_has_no_packed_f32_op_that_reads_src1_high_half_into_the_low_lane(tmp_path):
    """gfx950 / ROCm 7.2 (tools/pkfma_probe.hip): v_pk_{fma,mul,add}_f32 with op_sel[1] = 1 -- the low half of the
    packed operation takes src1's HIGH register, which is how hipcc's SLP vectoriser broadcasts an operand to two independent fp32
    chains -- returns wrong values whenever a wave of another kernel runs MFMA on the same SIMD (two processes on one GPU, or two
    streams of one process).  The library is built with -fno-slp-vectorize; this test disassembles what was built and fails if that
    instruction form is back."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    from dynamic_llava_amd import build_ext

    so = tmp_path / "lib.so"
    shutil.copy(build_ext.build(verbose=False), so)
    subprocess.run([objdump, "--offloading", so.name], cwd=tmp_path, check=True, capture_output=True)  # extracts the gfx950 code objects next to the copy
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert objs, "no gfx950 code object found in the library"
    bad, n_mfma = [], 0
    for o in objs:
        dis = subprocess.run([objdump, "-d", o], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        n_mfma += dis.count("v_mfma")
        bad += [ln.strip() for ln in dis.splitlines() if re.search(r"v_pk_(fma|mul|add)_f32", ln) and re.search(r"op_sel:\[[01],1", ln)]
    assert n_mfma > 0, "disassembly looks empty"
    assert not bad, f"{len(bad)} packed fp32 instructions with op_sel[1]=1, e.g. {bad[:3]}"


def test_decode_schedule_rule_is_the_same_chunked_and_step_by_step():
    """Round 4: the decode attention is scheduled from the lengths the predictor leaves, by a deterministic rule (KVSlabCache.sched_*): chunks of
    4, 4, then `sync_every` steps; chunk i uses the evicted group's longest row as observed after chunk i-2.  generate() evaluates the rule chunk by
    chunk (observations arrive one chunk late), a forward()-driven loop step by step with an immediate read at every chunk boundary.  Both must give
    every decode step the same bounds -- that is what keeps generate() and a forward() loop bit-identical -- and the bounds must always cover the
    true lengths (they size split-KV launches; a bound below the true length would only cost speed, but it would be a bug in the rule)."""
    import random

    import torch

    from dynamic_llava_amd.cache import KVSlabCache

    def lens_after(keep, n_sparse0, produced):  # evicted group's length when `produced` tokens exist: prefill + kept decode tokens so far
        return n_sparse0 + sum(keep[: produced - 1])

    for seed in range(20):
        rnd = random.Random(seed)
        S = rnd.choice([1, 3, 8, 16])
        n_full0, n_sparse0, max_new = rnd.randint(40, 700), rnd.randint(20, 200), rnd.randint(2, 90)
        keep = [rnd.random() < rnd.choice([0.0, 0.3, 0.9, 1.0]) for _ in range(max_new)]
        # --- generate(): chunk by chunk, observation of chunk k consumed before chunk k + 2 ---
        a = KVSlabCache(2, 1, 1, 1, 8, 8, torch.float32, "cpu")
        a.logical_cap = a.sparse_cap = 10 ** 6
        a.sched_begin(n_full0, n_sparse0, S)
        bounds_a, produced, pending = {}, 1, []
        while produced < max_new:
            while len(pending) > 1:
                at = pending.pop(0)
                a.sched_observe(at, lens_after(keep, n_sparse0, at))
            n = min(a.sched_chunk(), max_new - produced)
            for j in range(n):
                bounds_a[produced + j] = (a.key_bound(0), a.key_bound(1))
            produced += n
            a.sched_advance(n)
            pending.append(produced)
        # --- forward() loop: step by step, immediate read at chunk boundaries ---
        b = KVSlabCache(2, 1, 1, 1, 8, 8, torch.float32, "cpu")
        b.logical_cap = b.sparse_cap = 10 ** 6
        b.sched_begin(n_full0, n_sparse0, S)
        bounds_b = {}
        for produced in range(1, max_new):
            if b.sched_at_boundary():
                if b._sch["chunks"] > 0:
                    b.sched_observe(b._sch["produced"], lens_after(keep, n_sparse0, produced))
                b.sched_chunk()
            bounds_b[produced] = (b.key_bound(0), b.key_bound(1))
            b.sched_advance(1)
        assert bounds_a == bounds_b, f"seed {seed}: the two evaluations of the rule disagree"
        for produced, (fb, sb) in bounds_a.items():
            # the step that consumes token `produced` attends the rows cached so far + the new token
            assert fb >= n_full0 + produced and sb >= lens_after(keep, n_sparse0, produced) + 1, (seed, produced, fb, sb)
            assert sb <= lens_after(keep, n_sparse0, produced) + 1 + 2 * max(S, 4) + 4, "the bound stays within two chunks of the true length"


def test_split_factor_follows_the_bounds_not_the_capacity():
    import torch

    from dynamic_llava_amd.cache import KVSlabCache

    c = KVSlabCache(4, 2, 1, 32, 128, 2048, torch.float32, "meta")
    c.logical_cap, c.sparse_cap = 2048, 1588  # BASELINE configs[4]: 1588 slots reserved in layers >= 2 ...
    assert c.n_splits(3, 40) == 6 and c.n_splits(0, 40) == 6
    c.set_bounds(700, 260)  # ... of which 259 + the new token are in use
    assert c.n_splits(0, 40) == 6 and c.n_splits(3, 40) == 5
    c.single_split_max_keys = 576  # the fused q|k|v + attention launch of the 13B model (model._single_split_max_keys)
    assert c.n_splits(3, 40) == 1 and c.n_splits(0, 40) == 6
    c.set_bounds(None, None)
    assert c.n_splits(3, 40) == 6


def test_width_bucket_rule():
    """The prefill's width bucket: the smallest width >= W whose COMPACTED row count (W - 1 + kept image tokens) is a multiple of 16 -- never more
    than 15 extra rows, idempotent, monotone, and the identity when bucketing is off or nothing is compacted to a multiple."""
    from oracle import fixtures as fx
    from dynamic_llava_amd.config import DynamicLlavaConfig
    from dynamic_llava_amd.model import DynamicLlavaLlamaForCausalLM

    cfg = DynamicLlavaConfig.from_namespace(fx.tiny_config())
    m = DynamicLlavaLlamaForCausalLM(cfg, with_vision_tower=False)
    n_feat = 576
    kept = int(n_feat * cfg.sparse_config["vision_keep_rate"])
    prev = 0
    for W in range(2, 200):
        Wb = m._width_bucket(W, n_feat)
        assert W <= Wb < W + 16 and (Wb - 1 + kept) % 16 == 0 and m._width_bucket(Wb, n_feat) == Wb and Wb >= prev
        prev = Wb
    assert len({m._width_bucket(W, n_feat) for W in range(44, 101)}) == 5  # the eval stream's 57 widths (35 + 1 + U[8,64]) share five graphs
    m.prefill_width_bucket = 0
    assert all(m._width_bucket(W, n_feat) == W for W in (3, 57, 100))


def test_profile_tools_name_kernels_the_library_really_contains():
    """The evidence under profiles/ is produced by tools that pick dispatches BY KERNEL NAME (tools/pmc_report.py chunks the probe's main
    launches; bench.py names the dominant kernel): a kernel that was renamed or replaced (round 4: down_proj moved to
    gemv_b1_plain_halves_kernel and the PMC report silently lost a case) must fail here, not on the GPU box."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = open(os.path.join(root, "dynamic_llava_amd", "libdynllava_hip.so"), "rb").read()
    src = open(os.path.join(root, "tools", "pmc_report.py")).read()
    m = re.search(r"for k in \(([^)]*)\)", src)
    assert m, "pmc_report.py: kernel filter not found"
    names = re.findall(r'"([a-z0-9_]+)"', m.group(1))
    assert len(names) >= 6
    for n in names:
        assert n.encode() in blob, f"tools/pmc_report.py filters on `{n}`, which is not a kernel of the built library"
    # every weight-streaming kernel the batch-1 decode step can launch is covered by that filter
    for n in ("gemv_kernel", "gemv_b1_plain_kernel", "gemv_b1_plain_halves_kernel", "gemv_qkv_attn_kernel", "gemv_gu_tp_kernel"):
        assert n in names and n.encode() in blob
    probe = open(os.path.join(root, "tools", "pmc_probe.py")).read()
    for n in set(re.findall(r'"kernel": "([a-z0-9_]+)"', probe)):
        assert n.encode() in blob, f"tools/pmc_probe.py plans a case on `{n}`, which is not a kernel of the built library"


def test_linear_packed_launch_shapes_cover_the_chip_once():
    """Host-side choice of (units per workgroup, k ranges) for dl_linear_packed: at most one workgroup per CU (256), gate|up in gate / up PAIRS, two k ranges
    for q|k|v where 8 units per workgroup suffice, four for the narrow projections' partial sums (7B and 13B shapes)."""
    from dynamic_llava_amd.model import DynamicLlavaLlamaForCausalLM as M

    for n_units, pairs, want in ((768, False, (6, 2)), (1376, True, (6, 1)), (960, False, (8, 2)), (1728, True, (8, 1))):
        nu, ks = M._lp_config(n_units, pairs)
        assert (nu, ks) == want and -(-n_units // nu) * ks <= 256 and (not pairs or nu % 2 == 0)
    for n_units, want in ((256, (4, 4)), (320, (3, 2))):
        nu, ks = M._lp_config_parts(n_units)
        assert (nu, ks) == want and -(-n_units // nu) * ks <= 256
    # more than 128 rows (X-bound): 8 units x 8 ranges where that is exactly one workgroup per CU, and only while the partial sums fit the 8 x 192-row workspace
    assert M._lp_config_parts(256, 170) == (8, 8) and M._lp_config_parts(256, 117) == (4, 4) and M._lp_config_parts(256, 200) == (4, 4) and M._lp_config_parts(320, 170) == (3, 2)


def test_boundary_band_scales_with_the_logit_magnitude():
    """tests' keep / evict decision band (oracle/fixtures.py): a few ulps of the larger logit, never an absolute number."""
    import torch

    from oracle import fixtures as fx

    assert fx.boundary_band([0.3, -0.2], torch.bfloat16) == 8 * 2.0**-7  # magnitudes below 1 are held to 1
    assert fx.boundary_band([16.0, 15.0], torch.bfloat16) == 8 * 2.0**-7 * 16
    assert fx.boundary_band([16.0, 15.0], torch.float32) < 1e-2
    assert fx.decision_may_differ([4.0, 3.99], torch.bfloat16, [4.0, 1.0], torch.bfloat16)  # the first pair's gap (0.01) is inside its band (0.25)
    assert not fx.decision_may_differ([4.0, 3.0], torch.bfloat16, [4.0, 1.0], torch.bfloat16)
