"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol the
header declares, the nn.Module surface matches the reference's (names, shapes, errors), and the
product path refuses to run without a GPU instead of falling back."""
import math
import os
import re

import numpy as np
import pytest
import torch

import vqa_playground_pytorch_amd as pkg
from vqa_playground_pytorch_amd import _lib, layers, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    handle = _lib.lib()
    assert handle.vqa_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "vqa_mi355x.h")).read()
    declared = set(re.findall(r"\b(vqa_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(handle, name), name


def test_library_was_built_from_this_tree():
    """vqa_source_hash() (written into the binary by csrc/Makefile) equals the hash of csrc/ + include/ as they are: the .so
    that travels to the GPU box is not older than the sources beside it."""
    built, tree = _lib.built_from_this_tree()
    assert len(tree) == 64 and built == tree, "libvqa_mi355x.so is stale: rebuild with `python __graft_entry__.py build`"


def test_workspace_queries_need_no_gpu():
    handle = _lib.lib()
    n = handle.vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(512, 36, 310, 510, 2)
    assert n >= 512 * 510 * 4 + 2 * 510 * 310 * 4
    assert handle.vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(0, 36, 310, 510, 2) == 0
    # (weight-gradient slabs: 256 sample groups on the 4x4-MFMA kernel, 128 on the VALU kernels; the query covers both)
    assert handle.vqa_object_difference_attention_bwd_workspace_bytes(512, 36, 310, 4) == 256 * 4 * 36 * 310 * 4
    assert handle.vqa_object_difference_attention_bwd_workspace_bytes(100, 36, 310, 4) == 100 * 4 * 36 * 310 * 4


def test_bad_arguments_return_error_codes_without_gpu():
    handle = _lib.lib()
    rc = handle.vqa_pairwise_relation_reduce_fwd(None, None, None, None, 1, None, 1, 1, 4, 1, None)
    assert rc == -1 and b"null pointer" in handle.vqa_last_error()
    rc = handle.vqa_softmax_attention_pool_fwd(None, None, None, None, 1, 1, 4, 1, None)
    assert rc == -1


def _names(golden_dir, fname):
    gold = np.load(os.path.join(golden_dir, fname))
    return sorted(k[2:-5] for k in gold.files if k.startswith("g.") and k.endswith(".norm"))


@pytest.mark.parametrize("cls,fname,nans,count", [(pkg.CoR2Model, "cor2_b4.npz", 2000, 11940244),
                                                  (pkg.ODAModel, "oda_b4.npz", 3000, 7348434)])
def test_state_dict_abi_matches_reference(cls, fname, nans, count, golden_dir):
    model = cls(["PAD", "UNK"], nans)
    assert sorted(n for n, _ in model.named_parameters()) == _names(golden_dir, fname)
    assert sum(p.numel() for p in model.parameters()) == count
    sd = model.state_dict()
    assert sd["compress_v.conv.weight"].shape == (310, 2048, 1)
    if cls is pkg.ODAModel:
        assert sd["att.conv_att.conv.weight"].shape == (4, 11160, 1)
    else:
        assert sd["att1.conv_att.conv.weight"].shape == (4, 510, 1)
        assert sd["fusion_final.list_linear1.1.linear.weight"].shape == (510, 1240)


def test_reference_error_conventions():
    with pytest.raises(ValueError):
        layers.Linear(16, 7)(torch.zeros(2, 15))
    with pytest.raises(ValueError):
        layers.MyLinear(16, 7)(torch.zeros(2, 15))
    with pytest.raises(ValueError):
        layers.MyConv1d(16, 7, 1, 1)(torch.zeros(2, 16))
    with pytest.raises(AssertionError):
        layers.MyATT(16, 3, 12, 8)
    with pytest.raises(ValueError):
        layers.MutanFusion(8, 6, 16, 2)(torch.zeros(3, 5, 7), torch.zeros(3, 6))
    with pytest.raises(ValueError):
        layers.QuestionVectorInput(2400)(torch.zeros(2, 26, dtype=torch.long))


def test_bmul_bmatmul_match_reference_golden(golden_dir):
    from oracle import seeded
    blocks = np.load(os.path.join(golden_dir, "blocks.npz"))
    t = torch.from_numpy
    np.testing.assert_allclose(layers.bmul(t(seeded.seeded_array((3, 5, 8), 11)), t(seeded.seeded_array((3, 8), 12))).numpy(),
                               blocks["bmul.out"], rtol=1e-6)
    np.testing.assert_allclose(layers.bmul(t(seeded.seeded_array((3, 5, 5, 8), 13)), t(seeded.seeded_array((3, 8), 12))).numpy(),
                               blocks["bmul4.out"], rtol=1e-6)
    np.testing.assert_allclose(layers.bmatmul(t(seeded.seeded_array((3, 2, 5), 14)), t(seeded.seeded_array((3, 5, 12), 15))).numpy(),
                               blocks["bmatmul.out"], rtol=1e-5, atol=1e-6)


def test_no_cpu_fallback():
    """The HIP path must fail loudly on CPU tensors (no eager fallback)."""
    with pytest.raises(_lib.VqaLibraryError):
        ops.softmax_attention_pool(torch.zeros(2, 5, 4), torch.zeros(2, 5, 8))
    with pytest.raises(_lib.VqaLibraryError):
        ops.pairwise_relation_reduce(torch.zeros(2, 5, 8), torch.zeros(2, 8), torch.zeros(2, 8), torch.zeros(2, 5, 4))
    model = pkg.CoR2Model(["PAD"], 10)
    with pytest.raises(_lib.VqaLibraryError):
        model({"v": torch.zeros(2, 36, 2048), "q_idxes": torch.zeros(2, 2400)})


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vqa_playground_pytorch_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dirpath, f)
    for f in ["config/CoR2.py", "config/ODA.py"]:
        path = os.path.join(ROOT, f)
        if os.path.exists(path):
            assert not re.search(r"^\s*(from|import)\s+oracle", open(path).read(), re.M)


def test_my_linears_cpu_path_and_stack_groups():
    """Host logic of the batched [B,.] layers without a GPU: the per-module fallback of layers.my_linears (CPU tensors),
    ops.StackParams on separately allocated parameters (a real stack whose gradient reaches every member), and the stack
    groups a model hands to the trainer (same-shaped members, no parameter in two groups)."""
    import torch

    from vqa_playground_pytorch_amd import CoR2Model, ODAModel, layers, ops
    from vqa_playground_pytorch_amd.trainer import collect_stack_groups

    torch.manual_seed(0)
    mods = [layers.MyLinear(6, 5, p=0.5, af="relu").eval() for _ in range(3)]
    x = torch.randn(4, 6)
    got = layers.my_linears(mods, x)
    assert got.shape == (4, 3, 5)
    for g, m in enumerate(mods):
        assert torch.equal(got[:, g], m(x))
    got_gf = layers.my_linears(mods, torch.stack([x, 2 * x, 3 * x], 1), group_first=True)
    assert got_gf.shape == (3, 4, 5) and torch.equal(got_gf[2], mods[2](3 * x))

    ps = [torch.randn(2, 3, requires_grad=True) for _ in range(4)]
    st = ops.stack_params(ps)
    assert st.shape == (4, 2, 3) and torch.equal(st, torch.stack([p.detach() for p in ps]))
    (st * torch.arange(4.0).view(4, 1, 1)).sum().backward()
    assert all(torch.equal(p.grad, torch.full((2, 3), float(i))) for i, p in enumerate(ps))

    for model in (CoR2Model(["PAD", "UNK"], 50), ODAModel(["PAD", "UNK"], 50)):
        groups = collect_stack_groups(model)
        assert len(groups) >= 6
        seen = set()
        for group in groups:
            assert len(group) >= 2 and all(p.shape == group[0].shape for p in group)
            assert not (seen & {id(p) for p in group})
            seen |= {id(p) for p in group}


def test_tuned_gemm_table_is_well_formed():
    """The shipped TunableOp table (tuned_gemms.py): validator rows for gfx950 + one row per GEMM shape, the classifier and the
    question projections of the BASELINE batch among them."""
    from vqa_playground_pytorch_amd import tuned_gemms
    rows = [line.rstrip("\n").split(",") for line in open(tuned_gemms.TABLE) if line.strip()]
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert validators["GCN_ARCH_NAME"].startswith("gfx950") and "ROCBLAS_VERSION" in validators and "HIPBLASLT_VERSION" in validators
    shapes = [r for r in rows if r[0] != "Validator"]
    assert len(shapes) >= 50 and all(len(r) == 4 and float(r[3]) > 0 for r in shapes)
    assert len({(r[0], r[1]) for r in shapes}) == len(shapes)                      # one solution per (op, shape)
    keys = {r[1] for r in shapes}
    assert "tn_2000_512_510_ld_510_510_2000" in keys                              # CoR2 classifier forward at B = 512
    assert any(k.startswith("tn_310_512_2400_B_4") for k in keys)                  # the four question projections, batched


def test_side_outputs_resolve_on_every_way_out():
    """alpha_dict entries stored as callables (training forwards only) never leak as callables: dict(), ** unpacking, copy,
    deepcopy and pickling all see the resolved value (cor2.py alpha_dict['feature'])."""
    import copy
    import pickle
    from vqa_playground_pytorch_amd.layers import SideOutputs
    make = lambda: SideOutputs({"alpha1": (torch.ones(2),), "feature": lambda: torch.zeros(3)})  # noqa: E731
    assert torch.equal(make()["feature"], torch.zeros(3))
    assert torch.equal(dict(make())["feature"], torch.zeros(3))
    assert torch.equal({**make()}["feature"], torch.zeros(3))
    assert torch.equal(make().copy()["feature"], torch.zeros(3))
    assert torch.equal(copy.deepcopy(make())["feature"], torch.zeros(3))
    assert torch.equal(pickle.loads(pickle.dumps(make()))["feature"], torch.zeros(3))
    assert all(torch.is_tensor(v) or isinstance(v, tuple) for v in make().values())
    assert all(torch.is_tensor(v) or isinstance(v, tuple) for _, v in make().items())


def test_library_options_are_read_once(monkeypatch):
    """The library's VQA_* knobs come from the environment ONCE (first use) and change afterwards only through
    vqa_set_option (per-launch getenv let backward diverge from forward)."""
    import glob
    from vqa_playground_pytorch_amd import _lib
    csrc = os.path.join(ROOT, "vqa_playground_pytorch_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")):
        text = open(path).read()
        if os.path.basename(path) != "api.hip":
            assert "getenv(" not in text, "%s reads the environment at launch time: use vqa::option()" % path
    _lib.set_option("VQA_TEST_KNOB", 3)
    _lib.set_option("VQA_TEST_KNOB", None)
    with pytest.raises(_lib.VqaLibraryError):
        _lib.check(_lib.lib().vqa_set_option(b"", b"1"), "set_option")


def test_tune_mode_never_targets_the_shipped_table(monkeypatch):
    from vqa_playground_pytorch_amd import tuned_gemms
    monkeypatch.setattr(tuned_gemms, "_state", {"done": False, "mode": None, "loaded": None})
    monkeypatch.setenv("VQA_TUNED_GEMMS", "tune")
    monkeypatch.delenv("VQA_TUNED_GEMMS_FILE", raising=False)
    monkeypatch.delenv("PYTORCH_TUNABLEOP_ENABLED", raising=False)
    with pytest.raises(ValueError, match="VQA_TUNED_GEMMS_FILE"):
        tuned_gemms.enable()


def test_grad_slots_are_keyed_by_the_flat_buffer():
    """Two registrations live side by side; ending one backward window leaves the other's slots and hand-out list alone, and
    a parameter is handed its slot once per pass (module-global slot registry)."""
    from vqa_playground_pytorch_amd import ops
    p1, g1, p2, g2 = (torch.zeros(16) for _ in range(4))
    ops.set_grad_slots(p1, g1)
    ops.set_grad_slots(p2, g2)
    try:
        w1, w2 = p1[4:8], p2[8:12]
        a = ops._grad_like(w1)
        assert a.data_ptr() == g1[4:8].data_ptr()
        ops.set_grad_slots(p1, None)                       # trainer 1 leaves its window
        b = ops._grad_like(w2)
        assert b.data_ptr() == g2[8:12].data_ptr()         # trainer 2's registration is intact
        assert ops._grad_like(w2).data_ptr() != b.data_ptr()   # second use in one pass: a fresh tensor
        assert ops._grad_like(w1).data_ptr() != a.data_ptr()   # dropped registration: fresh tensor
        ops.begin_backward(p2)
        assert ops._grad_like(w2).data_ptr() == b.data_ptr()
    finally:
        ops.set_grad_slots(p1, None)
        ops.set_grad_slots(p2, None)


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`bench.py --gpus N` without a launcher starts its own ranks and must fail loudly when the node has fewer than N
    GPUs (here: none) instead of printing an n_gpus=1 line (missing #1)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VQA_ONE_GPU_REHEARSAL")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "only 0 GPU(s) visible" in r.stderr and "{" not in r.stdout
    # and a launcher whose rank count disagrees with --gpus is refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr


def test_head_mode_selects_the_grouped_phases_per_model(monkeypatch):
    """VQA_HEAD (head.MODE): auto groups CoR2's [B,.] layers except the glimpse projections and leaves ODA on the library
    GEMMs; grouped / legacy force one form for both; odd feature widths never take the grouped kernels (8-byte operand loads)."""
    from vqa_playground_pytorch_amd import head
    dims = (2400, 310, 2048, 510, 620, 2000)
    want = {"auto": (True, False, False), "grouped": (True, True, True), "legacy": (False, False, False)}
    for mode, (cor2, oda, glimpses) in want.items():
        monkeypatch.setattr(head, "MODE", mode)
        assert head.supported("cor2", *dims) is cor2
        assert head.supported("oda", *dims) is oda
        assert head.glimpses_grouped() is glimpses
    monkeypatch.setattr(head, "MODE", "grouped")
    assert head.supported("cor2", 2400, 311) is False


def test_grouped_phases_follow_the_products_engine_and_size_their_parts(monkeypatch):
    """head.Phase: with the split products (the default) the phases that measured faster on the split engine run there, the
    others on the fp32 MFMA; VQA_GROUPED_ENGINE pins one engine for all; the Python tile rule is the library's (vqa_grouped_gemm_split_tile_cols); the part planner cuts a phase so that its
    in-order schedule over 256 units is no longer than the uncut one's."""
    from vqa_playground_pytorch_amd import head
    L_ = _lib.lib()
    for n in list(range(1, 700)) + [1020, 1240, 2048, 2400, 3000]:
        assert head.Phase.split_tile_cols(n) == L_.vqa_grouped_gemm_split_tile_cols(n), n
    assert [head.Phase.split_tile_cols(n) for n in (310, 2048, 2400, 510)] == [160, 128, 160, 128]
    monkeypatch.setattr(head.Phase, "ENGINE", "auto")
    monkeypatch.delenv("VQA_F32_PRODUCTS", raising=False)
    assert head.Phase.engine("classifier_bwd", 512) == "split" and head.Phase.engine("q_proj_bwd", 512) == "mfma"
    assert head.Phase.engine("classifier_bwd", 128) == "mfma" and head.Phase.engine() == "mfma"      # small batches stay on the fp32 MFMA
    monkeypatch.setenv("VQA_F32_PRODUCTS", "mfma")
    assert head.Phase.engine("classifier_bwd", 512) == "mfma"
    monkeypatch.setattr(head.Phase, "ENGINE", "split")
    assert head.Phase.engine("q_proj_bwd", 5) == "split" and head.Phase.engine() == "split"
    monkeypatch.setattr(head.Phase, "ENGINE", "mixed")
    assert [head.Phase.engine(n, 512) for n in ("classifier_bwd", "classifier_fwd")] == ["split", "mfma"]
    monkeypatch.setattr(head.Phase, "ENGINE", "fp64")
    with pytest.raises(ValueError):
        head.Phase.engine()
    # the four question projections at B = 512: 4 x (4 x 2 tiles) x 75 steps -> 8 parts each = 256 items, one per CU
    part = head.Phase._plan_split([(8, 75)] * 4)
    assert math.ceil(75 / part) == 8
    # a phase that already holds a chip's worth of short items is left uncut
    assert head.Phase._plan_split([(64, 10)] * 2 + [(32, 10)] * 3) >= 10


def test_split_engine_is_the_default_and_refuses_what_it_cannot_run(monkeypatch):
    """ops.split_products: on by default (round 5) and with VQA_F32_PRODUCTS=split, off with VQA_F32_PRODUCTS=mfma, anything else
    is an error; on, it takes only the shapes the split engine runs (queried from the library, no GPU needed), the weight
    gradient only with K % 128 == 0, and never an operand at an offset that is not 16-byte aligned -- everything else stays on
    the fp32 MFMA engine.  The workspace queries need no GPU either."""
    L_ = _lib.lib()
    monkeypatch.delenv("VQA_F32_PRODUCTS", raising=False)
    assert ops.f32_products() == "split" and ops.split_products(18432, 2048, 310, 2048, 0.5)
    monkeypatch.setenv("VQA_F32_PRODUCTS", "mfma")
    assert not ops.split_products(18432, 2048, 310, 2048, 0.5)
    monkeypatch.setenv("VQA_F32_PRODUCTS", "bf16")
    with pytest.raises(ValueError):
        ops.split_products(18432, 2048, 310, 2048, 0.5)
    monkeypatch.setenv("VQA_F32_PRODUCTS", "split")
    assert ops.split_products(18432, 2048, 310, 2048, 0.5) and ops.split_products(18432, 2048, 310, 2048, 0.0)
    assert not ops.split_products(18432, 2048, 310, 2048, 0.3)            # one-bit masks only
    assert not ops.split_products(144, 2048, 310, 2048, 0.0)              # short matrices stay on the LDS-tile engine
    assert not ops.split_products(18432, 2000, 310, 2000, 0.0)            # K % 64
    assert ops.split_products(18432, 192, 310, 192, 0.0) and not ops.split_products(18432, 192, 310, 192, 0.0, weight_gradient=True)
    buf = torch.zeros(64)
    aligned = buf[(-buf.data_ptr() // 4) % 4:]
    assert aligned.data_ptr() % 16 == 0
    assert ops.split_products(18432, 2048, 310, 2048, 0.0, tensors=(aligned, None))
    assert not ops.split_products(18432, 2048, 310, 2048, 0.0, tensors=(aligned[1:],))
    assert L_.vqa_linear_act_fwd_split_workspace_bytes(2048, 310) >= 20 * 64 * 3072
    assert L_.vqa_linear_act_dw_split_workspace_bytes(18432, 2048, 310) >= 16 * 310 * 2048 * 4
    assert L_.vqa_relation_projection_dgrad_split_supported(512, 36, 2048, 310) == 1
    assert L_.vqa_relation_projection_dgrad_split_supported(512, 35, 2048, 310) == 0
    assert L_.vqa_relation_projection_dgrad_split_workspace_bytes(2048, 310) >= 128 * 10 * 3072


def test_hot_kernels_have_no_waterfall_loops():
    """Round 6 found three kernels whose buffer loads ran inside waterfall loops (v_readfirstlane x 4 + compare + branch per load)
    because hipcc could not see that a descriptor / scalar offset was wave-uniform: the K1 -> K5 fused forward (a descriptor made
    behind a loop with a divergent exit: +15 us), the encoder's batched GEMM (a problem index out of a vector-ALU division: 151 ->
    114 us per step) and K3a's logits kernels (the wave index threadIdx.x >> 6 without readfirstlane: -28 us per training step).
    tools/waterfall_check.py finds the pattern in the ISA; here the two files that hold the split engine's entry points are
    compiled (device code only) and checked: no waterfall loop between the MFMAs of any of their kernels."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("waterfall_check", os.path.join(ROOT, "tools", "waterfall_check.py"))
    wf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wf)
    if not os.path.exists(wf.HIPCC):
        pytest.skip("no hipcc here")
    for name in ("gru_gemm.hip", "linear_split.hip"):
        res = wf.check(os.path.join(wf.CSRC, name))
        assert res, name
        bad = {k: v for k, v in res.items() if v[1] > 0}
        assert not bad, (name, bad)


def test_repaired_loops_do_not_wait_for_loads_they_have_just_issued():
    """Round 6 (tools/sunk_loads_check.py): K2's weight gradient paid a memory latency per pass of its region loop (prefetch loads sunk
    to the end of the pass), the LDS-staged GEMM engines ran their global loads one stage ahead instead of two (a conditional second
    half in the loop body: vmcnt(0) in front of every stage's LDS writes).  The ISA of the repaired loops is checked here, device
    code only: the staged weight gradient's MFMA loops hold NO vector-memory load and read LDS in their first tenth; the bf16 GEMM's
    stage loops never wait for vmcnt(0)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sunk_loads_check", os.path.join(ROOT, "tools", "sunk_loads_check.py"))
    sl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sl)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "vqa_playground_pytorch_amd", "csrc")
    st = sl.loop_stats(sl.compile_to_asm(os.path.join(csrc, "object_difference.hip"), hipcc), ("oda_bwd_weight_mfma_staged_kernelILb1ELi1",))
    assert len(st) == 1, list(st)
    hot = [l for l in next(iter(st.values())) if len(l["mfma"]) >= 16]
    assert len(hot) >= 1, st
    for l in hot:
        assert not l["vmem"], l
        assert l["ds"] and max(l["ds"]) < max(4, l["n"] // 10), l
        assert not [w for i, w in l["waits"] if i < l["n"] // 2], l          # (the one wait is the explicit lgkmcnt(0) at the end)
    # the instances the configs[4] step launches (tools/by_grid.py table of the bf16 run); the 128 x 64 weight-gradient tile with
    # the dropout transform, not in that step, still shows one vmcnt(0) per pair of stages
    hot_bf16 = ("gemm_bf16_tn_kernelILi128ELi128ENS_13BfNoTransformELb1", "gemm_bf16_tn_kernelILi128ELi128ENS_10BfDropHalfELb1",
                "gemm_bf16_nt_kernelILi128ELi64ENS_13BfNoTransform", "gemm_bf16_nt_kernelILi128ELi64ENS_10BfDropHalf",
                "gemm_bf16_nt_kernelILi128ELi128ENS_13BfNoTransform")
    st = sl.loop_stats(sl.compile_to_asm(os.path.join(csrc, "bf16_path.hip"), hipcc), hot_bf16)
    assert len(st) == len(hot_bf16), list(st)
    for name, ls in st.items():
        stage_loops = [l for l in ls if len(l["mfma"]) >= 16 and l["vmem"]]
        assert stage_loops, name
        for l in stage_loops:
            assert not [w for i, w in l["waits"] if "vmcnt(0)" in w], (name, l["waits"])
